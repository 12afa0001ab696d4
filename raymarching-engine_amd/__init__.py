"""MI355X-native SDF raymarcher behind the render-job API of
radian628/raymarching-engine.  The compute path is
csrc/ -> libhip_raymarch.so (C ABI, include/hip_raymarch.h); this package is
the host side above it."""
from . import abi, scene, job, shard, params, capture  # noqa: F401

__all__ = ["abi", "scene", "job", "shard", "params", "capture", "native"]


def __getattr__(name):
    if name == "native":
        import importlib

        return importlib.import_module(".native", __name__)
    raise AttributeError(name)
