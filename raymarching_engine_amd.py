"""Import shim: the package directory is named ``raymarching-engine_amd`` (a
hyphen is not a valid Python identifier), so ``import raymarching_engine_amd``
lands here and this module replaces itself with the real package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raymarching-engine_amd")
_spec = importlib.util.spec_from_file_location(
    "raymarching_engine_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["raymarching_engine_amd"] = _mod
_spec.loader.exec_module(_mod)
