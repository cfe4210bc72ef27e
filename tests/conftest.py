import os
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is dense fp32 work on [rows, time_gap, 172] blocks.  A GPU box shows all of the host's cores (hundreds) but gives one
    # GPU's share of them (16): torch's default of one thread per visible core ran the oracle 10-20x slower than 16 threads do
    # (bench.py's cpu_baseline found the same).
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(autouse=True)
def _gpu_memory_log(request):
    """LSTEP_TEST_MEMLOG=<file>: after every GPU test, one line with the device memory still free and what torch's caching allocator holds
    (diagnostics for failures that only show behind the whole suite)."""
    yield
    path = os.environ.get("LSTEP_TEST_MEMLOG")
    if not path or request.node.get_closest_marker("gpu") is None:
        return
    import torch
    if not torch.cuda.is_available():
        return
    free, total = torch.cuda.mem_get_info()
    with open(path, "a") as f:
        f.write(f"{request.node.nodeid} free_GB={free / 2**30:.2f} total_GB={total / 2**30:.1f} reserved_GB={torch.cuda.memory_reserved() / 2**30:.2f} "
                f"allocated_GB={torch.cuda.memory_allocated() / 2**30:.2f}\n")


# ---- diagnostics (round 5): LSTEP_STREAM_LOG=<file> logs every stream PyTorch hands out from its round-robin pool (handle, running test,
# requesting code).  This is how round 4's fault was explained: the engine's update stream had been given the same queue as
# torch.cuda.graph's capture stream (profiles/r05_stream_alias_probe.txt).
_STREAM_LOG = os.environ.get("LSTEP_STREAM_LOG")
_CURRENT_TEST = ["<collection>"]


def _install_stream_probe():
    import traceback

    import torch
    orig_new = torch.cuda.Stream.__new__

    def new(cls, *a, **k):
        s = orig_new(cls, *a, **k)
        try:
            if s.cuda_stream != 0 and cls is torch.cuda.Stream:
                frames = [f for f in traceback.extract_stack()[:-1] if "dist-packages" not in f.filename and "/usr/lib" not in f.filename]
                who = "; ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in frames[-3:])
                with open(_STREAM_LOG, "a") as f:
                    f.write(f"STREAM {s.cuda_stream:#x} test={_CURRENT_TEST[0]} by={who}\n")
        except Exception:  # noqa: BLE001  (diagnostics must never break a test)
            pass
        return s
    torch.cuda.Stream.__new__ = staticmethod(new)


if _STREAM_LOG:
    _install_stream_probe()


@pytest.fixture(autouse=True)
def _stream_probe_test_name(request):
    _CURRENT_TEST[0] = request.node.nodeid
    yield


@pytest.fixture(autouse=True)
def _checked_build_record(request):
    """With a checked build loaded (LSTEP_LIB=.../liblstep_hip_checked.so, tools/build_checked.py) every GPU test ends by asking the library for
    its sticky out-of-range record: an id that would have faulted the GPU (or read a neighbour's memory) fails THIS test, by kernel."""
    checked = request.node.get_closest_marker("gpu") is not None and os.environ.get("LSTEP_LIB", "").find("checked") >= 0
    if checked:
        import torch
        checked = torch.cuda.is_available()
    if checked:
        from lstep_amd import _native as nat
        nat.set_debug_limits(0, 0, reset=True)      # (the table heights are process-wide: every test starts from "unknown"; models raise them)
    yield
    if checked:
        nat.check_device_errors()
