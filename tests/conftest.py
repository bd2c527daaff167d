import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A process that will use both torch's HIP runtime (RCCL tests) and liblinrad_hip.so must let torch find the device FIRST:
    once the library has initialised HIP through the system libamdhip64, torch's own copy reports no GPUs
    ("ProcessGroupNCCL is only supported with GPUs").  bench.py imports torch first for the same reason."""
    if any(it.get_closest_marker("gpu") for it in items):
        try:
            import torch
            if torch.cuda.device_count() > 0 and torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # noqa: BLE001  (no torch / no GPU: the gpu tests will say so themselves)
            pass
