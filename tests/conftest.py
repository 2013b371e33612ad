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
