import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# torch (used only for device-pointer plumbing in the multi-GPU tests) ships its own HIP runtime; load it before
# libugs.so pulls in the system one, whatever subset of the tests runs, or torch later reports "No HIP GPUs".
try:
    import torch  # noqa: F401,E402
except Exception:  # torch is optional for the CPU-only tests
    pass


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _heap_guard(request):
    """GPU tests only: host heap canaries around every test (tests/heapguard.py)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import heapguard
    g = request.session.__dict__.setdefault("_ugs_heap_guard", heapguard.HeapGuard())
    yield
    bad = g.check()
    g.plant()
    assert not bad, "host heap memory changed behind the process's back during this test (size, offset, byte): %r" % (bad[:8],)


def pytest_runtest_setup(item):
    """count the GPU tests of this session (tests/test_zz_gpu_coverage.py only judges a full run)"""
    if item.get_closest_marker("gpu") is not None:
        item.session.__dict__["_ugs_gpu_tests"] = item.session.__dict__.get("_ugs_gpu_tests", 0) + 1
