import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# a fatal signal inside the native code prints the C stack of the thread it was raised on (liblinrad_hip's LRH_CRASH_TRACE handler, behind
# pytest's faulthandler): the library reads the switch when it is loaded
os.environ.setdefault("LRH_CRASH_TRACE", "1")
os.environ.setdefault("LRH_ALLOC_LOG", "2")     # allocation journal in memory (no output unless the crash trace prints it): a faulting address can be matched with a buffer


def pytest_sessionfinish(session, exitstatus):
    """which HIP / ROCr runtime files this process ended up with (torch bundles its own): gpurun_out/runtime_maps.txt"""
    try:
        with open("/proc/self/maps") as f:
            libs = sorted({ln.split()[-1] for ln in f if " r-xp " in ln and any(k in ln for k in ("libamdhip64", "libhsa-runtime64", "liblinrad", "librccl"))})
        if libs:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            ident = []
            try:                                            # which box / GPU this was: a fault that comes and goes is first of all correlated with that
                import glob
                import socket
                ident.append("host " + socket.gethostname())
                for fn in sorted(glob.glob("/sys/class/drm/card*/device/unique_id")):
                    ident.append(fn.split("/")[4] + " unique_id " + open(fn).read().strip())
                for fn in sorted(glob.glob("/sys/class/drm/card*/device/current_compute_partition")):
                    ident.append(fn.split("/")[4] + " compute_partition " + open(fn).read().strip())
            except Exception:  # noqa: BLE001
                pass
            with open(os.path.join(ROOT, "gpurun_out", "runtime_maps.txt"), "w") as f:
                f.write("\n".join(libs + ident) + "\n")
    except OSError:
        pass


import pytest  # noqa: E402


@pytest.hookimpl(trylast=True)
def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # pytest captures fd 2 while a test runs; its faulthandler plugin writes to a duplicate of the real stderr -- the library's crash trace goes
    # to the same descriptor (and to a file the GPU box hands back)
    try:
        from _pytest.faulthandler import fault_handler_stderr_fd_key
        os.environ["LRH_CRASH_TRACE_FD"] = str(config.stash[fault_handler_stderr_fd_key])
    except Exception:  # noqa: BLE001
        pass
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        os.environ.setdefault("LRH_CRASH_TRACE_FILE", os.path.join(ROOT, "gpurun_out", "crash_trace.txt"))
    except OSError:
        pass


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
