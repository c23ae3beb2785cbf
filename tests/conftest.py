import os
import sys

# torch FIRST: its wheel bundles its own HIP runtime (ROCm 7.0), the engine library links the system's (ROCm 7.2).  Whichever is
# loaded second finds "No HIP GPUs are available"; bench.py imports torch before the engine for the same reason, and a test file
# that is collected alone (pytest tests/test_precision.py) must not depend on another file's import order.
try:
    import torch  # noqa: F401
except Exception:  # noqa: BLE001  (CPU-only checks still run without it)
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
