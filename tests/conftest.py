import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The first `import torch` on a fresh box pages the image in and can take minutes -- once (round 4) longer than the per-test
    timeout of pytest.ini, inside whichever test happened to import it first.  Here it is nobody's test time."""
    try:
        import torch  # noqa: F401
    except Exception:  # a test that needs it says so itself
        pass
