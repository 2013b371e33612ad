import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# NO torch in a test process that loads libugs.so: torch's wheel bundles a HIP runtime, an HSA runtime and an RCCL under the system
# libraries' SONAMEs (usearch12_amd/hostgroup.py).  Rounds 1-5 imported torch here first, so the whole GPU suite ran the product on
# torch's ROCm 7.0 runtime instead of the system one the product ships with; the tests that want torch.distributed
# (world-2 gloo) run it in child processes that never load libugs.so.


def d2h(ptr, nbytes):
    """`nbytes` of device memory at `ptr` as a numpy byte array (hipMemcpy through the runtime libugs.so is linked with)"""
    import ctypes
    import numpy as np
    from usearch12_amd import capi
    capi.lib()
    hip = ctypes.CDLL("libamdhip64.so.7")                         # (already mapped: the loader hands back the same library)
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    out = np.zeros(int(nbytes), np.uint8)
    if nbytes:
        rc = hip.hipMemcpy(out.ctypes.data, ctypes.c_void_p(int(ptr)), int(nbytes), 2)      # hipMemcpyDeviceToHost
        assert rc == 0, "hipMemcpy: %d" % rc
    return out


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
