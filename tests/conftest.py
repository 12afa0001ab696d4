import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# a scene's culling grid is built once the scene has earned it (4 Mi pixel-samples: include/hip_raymarch.h
# rm_ctx_set_cull_min_pixels); the tests' renders are small and are about the grid's bits: with the first render
os.environ.setdefault("RM_CULL_MIN_PIXELS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The first `import torch` on a fresh box pages the image in and can take minutes -- once (round 4) longer than the per-test
    timeout of pytest.ini, inside whichever test happened to import it first.  Here it is nobody's test time."""
    try:
        import torch  # noqa: F401
    except Exception:  # a test that needs it says so itself
        pass
