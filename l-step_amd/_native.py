"""ctypes binding of ``liblstep_hip.so`` (C ABI declared in ``include/lstep_hip.h``).

The library is built in-tree by :func:`build_library` (plain ``hipcc --offload-arch=gfx950``; no torch headers,
no JIT cache) so the ``.so`` travels with the repository snapshot to the GPU box.  There is NO fallback: if the
library cannot be loaded every op raises :class:`LstepNativeError`.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))
LIB_PATH = os.environ.get("LSTEP_LIB", os.path.join(CSRC, "liblstep_hip.so"))  # LSTEP_LIB: A/B builds for tuning
SOURCES = ["api.hip", "sampler.hip", "gather.hip", "history.hip", "segment.hip", "group.hip", "dense.hip", "tail.hip", "loss.hip", "head.hip", "fftcoef.hip", "update.hip", "adam.hip",
           "compose.hip", "shard.hip", "rng_sampler.hip", "hub.hip"]
HEADERS = [os.path.join(CSRC, "lstep_common.h"), os.path.join(CSRC, "lstep_mma.h"), os.path.join(INCLUDE, "lstep_hip.h")]

LSTEP_OK, LSTEP_EINVAL, LSTEP_EHIP = 0, -1, -2
ABI_VERSION = 40
BRANCH_EDGE_NODE, BRANCH_PE, WEIGHTED_SUM = 1, 2, 4


class LstepNativeError(RuntimeError):
    pass


class CsrStruct(C.Structure):
    """``lstep_csr_t``"""
    _fields_ = [("indptr", C.c_void_p), ("nbr", C.c_void_p), ("eid", C.c_void_p), ("ts", C.c_void_p),
                ("num_rows", C.c_int64), ("nnz", C.c_int64), ("max_degree", C.c_int64)]


class WgradDesc(C.Structure):
    """``lstep_wgrad_desc_t`` (include/lstep_hip.h): one product of ``lstep_linear_wgrad_batch``."""
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("m", C.c_int64), ("n", C.c_int32),
                ("k", C.c_int32), ("ldy", C.c_int32), ("ldx", C.c_int32), ("ld_dw", C.c_int32), ("reserved", C.c_int32)]


class RingRef(C.Structure):
    """``lstep_ring_ref_t``: a slot of the history ring whose index lives on the device, slot = (*start + add) % slots."""
    _fields_ = [("start", C.c_void_p), ("add", C.c_int32), ("slots", C.c_int32), ("slot_stride", C.c_int64)]


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES] + HEADERS


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps() if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False, defines=(), lib_path: str = None) -> str:
    """Compile every HIP source for gfx950 into ``csrc/liblstep_hip.so`` (cross-compiles without a GPU): one object per source under
    ``csrc/build/`` (compiled in parallel, re-compiled only when the source or a header is newer), then one link.  ``defines`` /
    ``lib_path``: an A/B build with extra -D flags into another file (e.g. ``LSTEP_EXACT_TANH=1``: libm ``tanhf`` instead of the
    hardware-exp form, tools/drift.py), loaded with the ``LSTEP_LIB`` environment variable."""
    out = lib_path or LIB_PATH
    if lib_path is None and not defines and not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise LstepNativeError("hipcc not found: cannot build liblstep_hip.so")
    tag = "".join(c if c.isalnum() else "_" for c in "_".join(defines))
    objdir = os.path.join(CSRC, "build" + ("_" + tag if tag else ""))
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}"] + [f"-D{d}" for d in defines]
    newest_header = max(os.path.getmtime(h) for h in HEADERS if os.path.exists(h))
    jobs = []
    for src in SOURCES:
        sp, op = os.path.join(CSRC, src), os.path.join(objdir, src.replace(".hip", ".o"))
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), newest_header):
            jobs.append((sp, op))

    def compile_one(job):
        sp, op = job
        cmd = base + ["-c", sp, "-o", op + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            os.replace(op + ".tmp", op)
        return sp, r

    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(len(jobs), int(os.environ.get("LSTEP_BUILD_JOBS", str(min(8, os.cpu_count() or 1))))))
        with ThreadPoolExecutor(workers) as pool:
            results = list(pool.map(compile_one, jobs))
        bad = [(sp, r) for sp, r in results if r.returncode != 0]
        if bad:
            raise LstepNativeError("hipcc failed:\n" + "\n".join(f"{sp}:\n{r.stdout}{r.stderr}" for sp, r in bad))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES] + ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(link))
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise LstepNativeError("hipcc (link) failed:\n" + r.stdout + r.stderr)
    os.replace(out + ".tmp", out)
    if out == LIB_PATH:
        global _LIB
        _LIB = None
    return out


_LIB = None

# name -> (restype, argtypes); must list every symbol of include/lstep_hip.h (tests/test_host_cpu.py::test_abi_exports_every_declared_symbol)
_P, _I32, _I64, _U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
SIGNATURES = {
    "lstep_abi_version": (C.c_int, []),
    "lstep_adam_step": (C.c_int, [_I32, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "lstep_last_error": (C.c_char_p, []),
    "lstep_stream_create": (C.c_int, [C.POINTER(C.c_void_p), _I32]),
    "lstep_stream_destroy": (C.c_int, [_P]),
    "lstep_debug_bounds_check_enabled": (C.c_int, []),
    "lstep_debug_set_limits": (C.c_int, [_I64, _I64]),
    "lstep_debug_device_error": (C.c_int, [C.POINTER(C.c_int64)]),
    "lstep_sample_recent": (C.c_int, [C.POINTER(CsrStruct), _P, _I64, _P, _I64, _I32, _P, _P, _P, _P, _P]),
    "lstep_time_encode": (C.c_int, [_P, _P, _I64, _P, _P, _I32, _P, _P]),
    "lstep_gather_aggregate_fwd": (C.c_int, [C.POINTER(CsrStruct), _P, _P, _P, _I32, _I32, _P, _P, _I32, _P, _P, _P, _I64, _I32,
                                             _I32, _U32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "lstep_gather_aggregate_fwd_skip": (C.c_int, [C.POINTER(CsrStruct), _P, _P, _P, _I32, _I32, _P, _P, _I32, _P, _P, _P, _I64, _I32,
                                                  _I32, _U32, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "lstep_hub_capacity": (_I64, [_I64, _I32]),
    "lstep_hub_worklist": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _I64, _P, _P, _I64, _P]),
    "lstep_hub_node_sums": (C.c_int, [C.POINTER(CsrStruct), _P, _I32, _P, _P, _I32, _P, _P, _P, _I64, _P, _I32, _P]),
    "lstep_gather_aggregate_bwd": (C.c_int, [C.POINTER(CsrStruct), _P, _I32, _I32, _P, _P, _I32, _P, _P, _P, _I64, _I32, _P, _P, _P,
                                             _I32, _I32, _I32, _P, _P, _P, _P, _P]),
    "lstep_gather_explicit_fwd": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P, _I32, _P, _P, _P, _I64, _I32, _I32, _U32, _P, _P, _P, _P, _P, _I64,
                                           _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "lstep_gather_explicit_bwd": (C.c_int, [_P, _I32, _I32, _P, _P, _I32, _P, _P, _I64, _I32, _P, _P, _P, _I64, _P, _P, _P, _I32, _I32, _I32,
                                           _P, _P, _P, _P, _P]),
    "lstep_history_filter_fwd": (C.c_int, [_P, _I64, _I64, _I32, _I32, _I32, _I32, _P, _I64, _P, _P, _P]),
    "lstep_history_filter_bwd_chunks": (_I64, [_I64]),
    "lstep_history_filter_bwd": (C.c_int, [_P, _I64, _I64, _I32, _I32, _I32, _I32, _P, _I64, _P, _P, _P]),
    "lstep_history_slot_bits": (C.c_int, [_P, _I32, _I64, _I32, _I32, C.POINTER(RingRef), _P]),
    "lstep_history_mark": (C.c_int, [_P, _I32, _I64, _I32, _P, _I64, _I32, _I32, C.POINTER(RingRef), _P]),
    "lstep_ring_tick": (C.c_int, [_P, _I32, _P]),
    "lstep_history_filter_runs_workspace": (_I64, [_I32, _I32]),
    "lstep_history_filter_runs_fwd": (C.c_int, [_P, _I64, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _P, _I64, _P, _P, _P, _P, _P, _P,
                                               C.POINTER(RingRef), _P]),
    "lstep_history_filter_runs_bwd_chunks": (_I64, [_I64, _I32]),
    "lstep_history_filter_runs_bwd": (C.c_int, [_P, _I64, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _P, _I64, _P, _P, C.POINTER(RingRef), _P]),
    "lstep_copy_rows": (C.c_int, [_P, _P, _I32, _I64, _P, _I64, _I64, C.POINTER(RingRef), _P]),
    "lstep_history_advance_oldest": (C.c_int, [_P, _P, _I32, _I64, _P, _I32, _I32, _I64, C.POINTER(RingRef), _P]),
    "lstep_history_filter_runs_finish": (C.c_int, [_P, _I32, _I32, _P, _P]),
    "lstep_segment_rows_sum_workspace": (C.c_int64, [_I64, _I32, _I32]),
    "lstep_segment_rows_sum": (C.c_int, [_P, _I32, _I32, _P, _P, _I32, _P, _P, _P, _I64, _P, _I32, _I32, _P, _P, _I64, _P]),
    "lstep_sort_live_bounded_workspace": (_I64, [_I64, _I64, _I32]),
    "lstep_sort_live_bounded": (C.c_int, [_P, _I64, _I32, _I32, _I64, _P, _I64, _P, _P, _P, _P, _P]),
    "lstep_segment_rows_sum_live": (C.c_int, [_P, _I32, _I32, _P, _P, _I32, _I64, _P, _P, _I32, _I32, _P, _I64, _P]),
    "lstep_pull_keys": (C.c_int, [_P, _I64, _P, _I64, _I32, _I32, _I64, _P, _P]),
    "lstep_pull_blocks": (C.c_int, [_P, _P, _I32, _I64, _I64, _P, _P, _P]),
    "lstep_owner_partition_workspace": (_I64, [_I64, _I32]),
    "lstep_owner_partition": (C.c_int, [_P, _I64, _P, _I32, _I64, _P, _I64, _P, _P, _P, _P, _P]),
    "lstep_scatter_owner_rows": (C.c_int, [_P, _I32, _P, _P, _I32, _I64, _P, _I32, _P, _P]),
    "lstep_rows_by_id": (C.c_int, [_P, _I64, _P, _I32, _P, _I32, _P]),
    "lstep_count_before_host": (C.c_int, [_P, _P, _I64, _P, _P, _I64, _P]),
    "lstep_sample_random_host": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "lstep_sample_random_sorted_host": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I32]),
    "lstep_batch_prepare": (C.c_int, [_P, _P, _P, _P, _I64, _P, _P, _P, _P, _P]),
    "lstep_padding_rows_finish": (C.c_int, [_P, _I64, _I32, _P, _I32, _P]),
    "lstep_scatter_add_overflow": (C.c_int, [_P, _I32, _I32, _P, _P, _P, _I64, _I32, _P, _I32, _P]),
    "lstep_spliced_grad_small": (C.c_int, [_P, _I64, _I32, _P, _I32, _P, _P, _I64, _P, _I32, _I32, _P, _I32, _I64, _P]),
    "lstep_scatter_rows": (C.c_int, [_P, _I32, _P, _I64, _P, _P]),
    "lstep_residual_tanh_rows": (C.c_int, [_P, _I32, _P, _I64, _P, _I32, _P]),
    "lstep_group_by_key_workspace": (_I64, [_I64, _I32]),
    "lstep_group_by_key": (C.c_int, [_P, _I64, _I32, _I32, _P, _I64, _P, _P, _P, _P, _P, _P]),
    "lstep_fft_coef_fwd": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P, _P]),
    "lstep_fft_coef_bwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P]),
    "lstep_padding_rows_sum_blocks": (_I64, [_I64]),
    "lstep_padding_rows_sum": (C.c_int, [_P, _I32, _P, _I64, _P, _I32, _I32, _P, _P]),
    "lstep_scatter_add_rows": (C.c_int, [_P, _I32, _I32, _P, _I64, _P, _I32, _P]),
    "lstep_update_entries_p1": (C.c_int, [_P, _I64, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "lstep_update_keys_p2": (C.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P]),
    "lstep_update_entries_p2": (C.c_int, [_P, _P, _I64, _P, _P, _P, _I32, _I32, _P, _I64, _P, _P, _P, _P, _P]),
    "lstep_update_rows": (C.c_int, [_P, _I32, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, C.POINTER(RingRef), _I32, _I32, _P]),
    "lstep_update_rows_pre": (C.c_int, [_P, _I32, _P, _I64, _P, _P, _P, _P, _P, _P, _I32, _I32, _P, C.POINTER(RingRef), _I32, _I32, _P]),
    "lstep_widen_ids": (C.c_int, [_P, _I64, _P, _P, _P]),
    "lstep_update_entries_p2_dev": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "lstep_head_fwd": (C.c_int, [_P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "lstep_head_bwd": (C.c_int, [_P, _P, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "lstep_head_pack": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P]),
    "lstep_link_loss_workspace": (_I64, [_I64]),
    "lstep_link_loss": (C.c_int, [_P, _P, _I64, _P, _P, _P, _I32, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "lstep_sort_live_workspace": (_I64, [_I64, _I32]),
    "lstep_sort_live": (C.c_int, [_P, _I64, _I32, _P, _I64, _P, _P, C.POINTER(C.c_int64), _P]),
    "lstep_linear_wgrad_workspace": (_I64, [_I64, _I32, _I32]),
    "lstep_linear_wgrad": (C.c_int, [_P, _I32, _P, _I32, _I64, _I32, _I32, _P, _I32, _P, _P, _I64, _P]),
    "lstep_linear_wgrad_batch_workspace": (_I64, [_I32, C.POINTER(WgradDesc)]),
    "lstep_linear_wgrad_batch": (C.c_int, [_I32, C.POINTER(WgradDesc), _P, _I64, _P]),
    "lstep_small_gemm": (C.c_int, [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I32, _I32, _I32, C.c_float, C.c_float, _P]),
    "lstep_tail_weights_pack": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "lstep_tail_weights_unpack": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "lstep_tail_fwd": (C.c_int, [_P, _I32, _P, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "lstep_tail_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _I64, _P]),
}


def load_library():
    """Load (never build) the shared library and attach the prototypes.  Raises if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise LstepNativeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the L-STEP HIP ops)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (torch/lib), our library links against
    # the SONAME libamdhip64.so.7.  Import torch first so that SONAME resolves to the copy torch already mapped;
    # loading /opt/rocm's copy beside it gives a second runtime that sees no device.
    import torch  # noqa: F401
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.lstep_abi_version() != ABI_VERSION:
        raise LstepNativeError("liblstep_hip.so ABI version mismatch; rebuild")
    _LIB = lib
    return lib


CHECK_TAGS = {1: "gather forward: row's own node id", 2: "gather forward: neighbour id (edge / PE channel slots)", 3: "gather forward: edge id",
              4: "gather forward: neighbour id (node channel slots)", 5: "gather backward: row's own node id", 6: "gather backward: neighbour id",
              7: "gather backward: edge id", 8: "sampler: node id", 9: "history filter: node id", 10: "update_rows: row id",
              11: "link loss: endpoint id", 12: "row scatter / residual update: row id", 13: "segment sum: source row", 14: "segment sum: segment id",
              15: "rows_by_id: row id", 16: "scatter_owner_rows: row id", 17: "spliced_grad_small: own-node id"}


def bounds_check_enabled() -> bool:
    """Is the loaded library a checked build (``-DLSTEP_BOUNDS_CHECK=1``: ``build_checked_library``, loaded with ``LSTEP_LIB=...``)?"""
    return bool(load_library().lstep_debug_bounds_check_enabled())


_DEBUG_LIMITS = [0, 0]


def set_debug_limits(node_rows: int, edge_rows: int, reset: bool = False):
    """Row counts of the node-shaped tables and of the edge table for the guards of a checked build whose entry points carry no row count
    of their own (no-op in a product build).  The limits are process-wide, so they only ever GROW (the largest tables any model of the
    process holds: an id valid for one of them never trips a guard); ``reset=True`` starts over (tests/conftest.py: per test)."""
    lib = load_library()
    if lib.lstep_debug_bounds_check_enabled():
        if reset:
            _DEBUG_LIMITS[:] = [0, 0]
        _DEBUG_LIMITS[0], _DEBUG_LIMITS[1] = max(_DEBUG_LIMITS[0], int(node_rows)), max(_DEBUG_LIMITS[1], int(edge_rows))
        check(lib.lstep_debug_set_limits(_DEBUG_LIMITS[0], _DEBUG_LIMITS[1]))


def device_error():
    """None, or (tag, description, index, limit, count) of the first out-of-range id a checked build's kernels met since the last call."""
    lib = load_library()
    if not lib.lstep_debug_bounds_check_enabled():
        return None
    out = (C.c_int64 * 4)()
    check(lib.lstep_debug_device_error(out))
    if out[0] == 0:
        return None
    return int(out[0]), CHECK_TAGS.get(int(out[0]), "?"), int(out[1]), int(out[2]), int(out[3])


def check_device_errors():
    """Raise if a checked build's kernels met an out-of-range id (they read the padding row instead of faulting the GPU)."""
    err = device_error()
    if err is not None:
        tag, what, idx, limit, count = err
        raise LstepNativeError(f"checked build: out-of-range id in {what} (tag {tag}): index {idx}, table rows {limit}; {count} offending load(s)")


CHECKED_LIB_PATH = os.path.join(CSRC, "liblstep_hip_checked.so")


def build_checked_library(force: bool = False) -> str:
    """The same sources with ``-DLSTEP_BOUNDS_CHECK=1`` into ``csrc/liblstep_hip_checked.so`` (use: ``LSTEP_LIB=<that path> python -m pytest ...``)."""
    if not force and os.path.exists(CHECKED_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(CHECKED_LIB_PATH) for d in _deps() if os.path.exists(d)):
        return CHECKED_LIB_PATH
    return build_library(defines=("LSTEP_BOUNDS_CHECK=1",), lib_path=CHECKED_LIB_PATH)


def check(rc: int):
    """Map the C error convention onto the reference's Python exceptions (AssertionError for K <= 0)."""
    if rc == LSTEP_OK:
        return
    msg = load_library().lstep_last_error().decode("utf-8", "replace")
    if rc == LSTEP_EINVAL and "greater than 0" in msg:
        raise AssertionError(msg)  # reference: utils/utils.py:156
    if rc == LSTEP_EINVAL:
        raise ValueError(msg)
    raise LstepNativeError(msg)


def ptr(t):
    """Device (or host) address of a torch tensor, or NULL."""
    return None if t is None else C.c_void_p(t.data_ptr())


def _raw_stream(device_index=None) -> int:
    """Handle of the current HIP stream of ``device_index`` (default: the current device).  ``torch.cuda.current_stream()`` builds a
    Python Stream object on every call (~8 us: a third of a millisecond per training iteration at ~35 native launches); the raw getter
    is a plain C call."""
    import torch

    get = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if get is None:      # older / newer torch without the private getter
        return torch.cuda.current_stream(device_index).cuda_stream
    return get(torch._C._cuda_getDevice() if device_index is None else device_index)


def current_stream():
    return C.c_void_p(_raw_stream())


def quiesce_collectives(device=None):
    """Call right before a graph capture that a stream which has carried launch-by-launch collectives will start or join.

    torch's NCCL watchdog thread polls the END EVENT of every collective issued launch by launch until it has seen it complete (one sweep
    of its list every ~100 ms).  A synchronous collective runs on the current stream, so its end event is recorded there -- on this
    package's update / pull streams in the launch-by-launch iterations.  When such a stream starts or joins a capture before the watchdog's
    next sweep, ``hipEventQuery`` on that event fails with hipErrorCapturedEvent ("operation not permitted on an event last recorded in a
    capturing stream") although the record itself was not captured, and the watchdog takes the process down: the abort round 4 saw once in
    ``bench.py`` and round 5 once in the RCCL golden-trace test, reproduced at will by ``tools/nccl_capture_after_eager_probe.py``
    (scenario ``same``: dies; capture on a stream that never carried a collective, or a drained watchdog list: fine).  So: finish the
    device's work and give the watchdog one sweep to drop the completed works -- a quarter of a second, once per capture."""
    import time

    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    try:
        if dist.get_backend() != "nccl":
            return
    except Exception:  # noqa: BLE001  (a group without a default backend name: nothing to wait for)
        return
    if torch.cuda.is_available():
        torch.cuda.synchronize(device)
    time.sleep(float(os.environ.get("LSTEP_CAPTURE_QUIESCE_S", "0.25")))


_ROLE_STREAMS = {}


def role_stream(dev, role: str, priority: int = 0):
    """THE stream of ``role`` on ``dev``: a HIP stream of our own (``lstep_stream_create``), wrapped as ``torch.cuda.ExternalStream``, one per
    (device, role) for the life of the process.  Roles: "update" (update_pe beside the backward pass), "ring-copy", "aux" (weight-gradient
    products, weight composition), "side" (edge re-gather of the backward pass), "capture" (every graph capture of this package), "pull",
    "dist-copy" (lstep_amd.parallel).

    Why not ``torch.cuda.Stream()``: PyTorch hands those out round-robin from a pool of 32 per device, and ``torch.cuda.graph`` takes its
    default capture stream from the same pool.  Behind ~130 tests of one process the engine's update stream WAS that capture stream
    (profiles/r05_stream_alias_probe.txt), while update_pe was issued by a second host thread: whatever that thread enqueued while the
    autograd thread captured the weight-composition backward was recorded into the graph instead of executed and replayed with stale
    arguments on every later step -- round 4's "memory access fault / identical wrong table, only behind the whole suite".  Dedicated
    streams cannot collide with the pool or with each other: two roles never share a queue, whatever else the process has created."""
    import torch

    dev = torch.device(dev)
    if dev.type != "cuda":
        return None
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (index, role)
    st = _ROLE_STREAMS.get(key)
    if st is None:
        lib = load_library()
        handle = C.c_void_p()
        with torch.cuda.device(index):
            check(lib.lstep_stream_create(C.byref(handle), int(priority)))
        st = _ROLE_STREAMS[key] = torch.cuda.ExternalStream(handle.value, device=torch.device("cuda", index))
    return st


_WORKSPACES = {}


def _workspace(dev, need: int):
    """Device scratch, one buffer per (device, stream, HOST THREAD), grown on demand.

    A scratch buffer lives across the launches of ONE native call sequence (partial sums and their reduction, a sort's passes, chunk flags
    and their join).  The stream is part of the key because sequences on different streams run concurrently; the host thread because the
    main thread (forward pass, update_pe) and the autograd engine's device thread (backward pass) may both issue sequences onto one
    stream -- harmless for stream order, fatal for a buffer shared by both (a clobbered chunk-flag word is an out-of-range read in the join
    kernel).  Python threads are keyed by name, the autograd engine's device thread by its ident."""
    import threading

    import torch

    if torch.cuda.is_current_stream_capturing():
        # inside a graph capture the scratch belongs to the graph's private pool: never cached, never shared with eager launches
        return torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
    th = threading.current_thread()
    who = th.ident if isinstance(th, threading._DummyThread) else th.name
    wkey = (dev, _raw_stream(dev.index if hasattr(dev, "index") and dev.index is not None else None), who)
    ws = _WORKSPACES.get(wkey)
    if ws is None or ws.numel() < need:
        ws = _WORKSPACES[wkey] = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=dev)
    return ws


def segment_workspace(dev, num_entries: int, width: int, time_dim: int = 0):
    """(pointer, bytes) of the scratch that makes ``lstep_segment_rows_sum`` independent of the order in which its waves finish (hub
    segments are joined in chunk order); (None, 0) with LSTEP_SEGMENT_ATOMICS=1, the A/B switch back to float atomics."""
    if os.environ.get("LSTEP_SEGMENT_ATOMICS") == "1":
        return None, 0
    need = int(load_library().lstep_segment_rows_sum_workspace(int(num_entries), int(width), int(time_dim)))
    if need == 0:
        return None, 0
    ws = _workspace(dev, need)
    return ws, ws.numel()


def sort_live(keys, key_bits: int):
    """``lstep_sort_live``: (sorted_keys, order, num_live) of the non-negative int32 keys; one host sync inside the call."""
    import torch

    lib = load_library()
    n = keys.numel()
    dev = keys.device
    sorted_keys = torch.empty(n, dtype=torch.int32, device=dev)
    order = torch.empty(n, dtype=torch.int32, device=dev)
    live = C.c_int64(0)
    if n:
        ws = _workspace(dev, int(lib.lstep_sort_live_workspace(n, key_bits)))
        with torch.cuda.device(dev):
            check(lib.lstep_sort_live(ptr(keys), n, int(key_bits), ptr(ws), ws.numel(), ptr(sorted_keys), ptr(order), C.byref(live),
                                      current_stream()))
    return sorted_keys, order, int(live.value)


def sort_live_bounded(keys, key_bits: int, sentinel: int, capacity: int):
    """``lstep_sort_live_bounded``: (sorted_keys [capacity], order [capacity], live_index [n], count (device int32 [1])); no host sync."""
    import torch

    lib = load_library()
    n = keys.numel()
    dev = keys.device
    sorted_keys = torch.empty(capacity, dtype=torch.int32, device=dev)
    order = torch.empty(capacity, dtype=torch.int32, device=dev)
    live_index = torch.empty(n, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws = _workspace(dev, int(lib.lstep_sort_live_bounded_workspace(n, capacity, key_bits)))
    with torch.cuda.device(dev):
        check(lib.lstep_sort_live_bounded(ptr(keys), n, int(key_bits), int(sentinel), int(capacity), ptr(ws), ws.numel(), ptr(sorted_keys),
                                          ptr(order), ptr(live_index), ptr(count), current_stream()))
    return sorted_keys, order, live_index, count


def small_mm(a, b, out=None, beta: float = 0.0):
    """``a @ b`` (2-D fp32 device tensors, any strides) through ``lstep_small_gemm``; ``out`` (any strides) receives
    ``a @ b + beta * out``.  For weight-sized operands: one wave per 16 x 16 output tile."""
    import torch

    lib = load_library()
    m, k = a.shape
    n = b.shape[1]
    if b.shape[0] != k or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise ValueError("small_mm: fp32 [m, k] @ [k, n] expected")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        check(lib.lstep_small_gemm(ptr(a), a.stride(0), a.stride(1), ptr(b), b.stride(0), b.stride(1), ptr(out), out.stride(0), out.stride(1),
                                   m, n, k, 1.0, float(beta), current_stream()))
    return out


def linear_wgrad(dy, x, want_bias: bool = True, out=None):
    """``(dy^T x, dy.sum(0))`` through ``lstep_linear_wgrad``; dy [m, n], x [m, k] fp32 device tensors with unit column stride.
    ``out=(dw, db)``: contiguous destination tensors (either may be None)."""
    import torch

    lib = load_library()
    m, n = dy.shape
    k = x.shape[1]
    if dy.stride(1) != 1 or x.stride(1) != 1 or x.shape[0] != m or dy.dtype != torch.float32 or x.dtype != torch.float32:
        raise ValueError("linear_wgrad: fp32 [m, n] / [m, k] operands with unit column stride expected")
    dev = dy.device
    dw = out[0] if out is not None and out[0] is not None else torch.empty((n, k), dtype=torch.float32, device=dev)
    db = (out[1] if out is not None and out[1] is not None else torch.empty(n, dtype=torch.float32, device=dev)) if want_bias else None
    if tuple(dw.shape) != (n, k) or not dw.is_contiguous() or (db is not None and (db.numel() != n or not db.is_contiguous())):
        raise ValueError("linear_wgrad: destination tensors must be contiguous [n, k] / [n]")
    if m == 0:
        dw.zero_()
        if db is not None:
            db.zero_()
        return dw, db
    ws = _workspace(dev, int(lib.lstep_linear_wgrad_workspace(m, n, k)))
    with torch.cuda.device(dev):
        check(lib.lstep_linear_wgrad(ptr(dy), dy.stride(0), ptr(x), x.stride(0), m, n, k, ptr(dw), k, ptr(db), ptr(ws), ws.numel(),
                                     current_stream()))
    return dw, db


def linear_wgrad_batch(items):
    """``lstep_linear_wgrad_batch``: ``items`` = [(dy, x, want_bias, out)] with the conventions of :func:`linear_wgrad` (``out`` = (dw, db)
    destination tensors or None); returns [(dw, db)].  One partial launch and one reduction launch for all products."""
    import torch

    lib = load_library()
    if not items:
        return []
    if len(items) > 8:
        return linear_wgrad_batch(items[:8]) + linear_wgrad_batch(items[8:])
    dev = items[0][0].device
    descs = (WgradDesc * len(items))()
    outs, keep = [], []
    for i, (dy, x, want_bias, out) in enumerate(items):
        m, n = dy.shape
        k = x.shape[1]
        if dy.stride(1) != 1 or x.stride(1) != 1 or x.shape[0] != m or dy.dtype != torch.float32 or x.dtype != torch.float32:
            raise ValueError("linear_wgrad_batch: fp32 [m, n] / [m, k] operands with unit column stride expected")
        dw = out[0] if out is not None and out[0] is not None else torch.empty((n, k), dtype=torch.float32, device=dev)
        db = (out[1] if out is not None and out[1] is not None else torch.empty(n, dtype=torch.float32, device=dev)) if want_bias else None
        if tuple(dw.shape) != (n, k) or not dw.is_contiguous() or (db is not None and (db.numel() != n or not db.is_contiguous())):
            raise ValueError("linear_wgrad_batch: destination tensors must be contiguous [n, k] / [n]")
        if m == 0:
            dw.zero_()
            if db is not None:
                db.zero_()
        descs[i] = WgradDesc(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, m, n, k,
                             dy.stride(0) if m else n, x.stride(0) if m else k, k, 0)
        outs.append((dw, db))
        keep.append((dy, x))
    need = int(lib.lstep_linear_wgrad_batch_workspace(len(items), descs))
    ws = _workspace(dev, need)
    with torch.cuda.device(dev):
        check(lib.lstep_linear_wgrad_batch(len(items), descs, ptr(ws), ws.numel(), current_stream()))
    return outs


class PendingCounts:
    """The three counts of a ``group_by_key(..., wait=False)`` call on their way to the host: a pinned buffer, an asynchronous copy and
    an event recorded right behind it.  ``get()`` waits for THAT point of the stream only, not for whatever was enqueued later."""

    def __init__(self, summary):
        import torch

        self.host = torch.empty(3, dtype=torch.int32, pin_memory=True)
        self.host.copy_(summary, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self._keep = summary

    def get(self):
        self.event.synchronize()
        return self.host.tolist()


def group_by_key(keys, key_bits: int, limit: int, wait: bool = True):
    """``lstep_group_by_key`` on an int32 device tensor.  Returns (sorted_keys, order, seg, uniq, (n_unique, n_below, n_unique_below));
    the three counts cost one host sync (``wait=False``: a ``PendingCounts`` instead, read later with ``.get()``; ``wait=None``: the
    device tensor ``summary`` int32 [3] itself, for consumers that read counts on the device).
    The scratch buffer is cached per device and grown on demand."""
    import torch

    lib = load_library()
    n = keys.numel()
    dev = keys.device
    need = int(lib.lstep_group_by_key_workspace(n, key_bits))
    ws = _workspace(dev, need)
    sorted_keys = torch.empty(n, dtype=torch.int32, device=dev)
    order = torch.empty(n, dtype=torch.int32, device=dev)
    seg = torch.empty(n, dtype=torch.int32, device=dev)
    uniq = torch.empty(n, dtype=torch.int32, device=dev)
    summary = torch.empty(3, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.lstep_group_by_key(ptr(keys), n, int(key_bits), int(limit), ptr(ws), ws.numel(), ptr(sorted_keys), ptr(order), ptr(seg),
                                     ptr(uniq), ptr(summary), current_stream()))
    if wait is None:       # counts stay on the device: the int32 [3] tensor itself
        return sorted_keys, order, seg, uniq, summary
    return sorted_keys, order, seg, uniq, (summary.tolist() if wait else PendingCounts(summary))
