"""Diagnostic: run a pytest selection with every ``torch.empty`` / ``torch.empty_like`` result pre-filled, so that a kernel that READS a
buffer it was supposed to write first shows up deterministically (fresh device memory is zero-filled by the driver, which hides such reads
in a short process; after big tests the caching allocator hands out dirty blocks).

    python tools/poison_empty.py nan  tests/test_hip_parity.py -k rng_sampler_vs_oracle     # float buffers = NaN, integer buffers = 0
    python tools/poison_empty.py one  tests/test_hip_parity.py -k rng_sampler_vs_oracle     # float buffers = NaN, integer buffers = 1

Integer buffers are never filled with an out-of-range value: a stray index must not become a GPU fault.  Not part of the test suite."""
import sys

import torch

MODE = sys.argv[1]
assert MODE in ("nan", "one")
_empty, _empty_like = torch.empty, torch.empty_like


def _fill(t):
    if t.numel() == 0 or t.is_meta:
        return t
    try:
        if t.is_floating_point() or t.is_complex():
            t.fill_(float("nan"))
        elif t.dtype == torch.bool:
            t.fill_(False)
        elif t.dtype == torch.uint8:
            pass            # scratch buffers of the native library: typed inside the kernels
        else:
            t.fill_(1 if MODE == "one" else 0)
    except Exception:       # (e.g. a capture in progress refusing the fill: leave the buffer alone)
        pass
    return t


def empty(*a, **k):
    return _fill(_empty(*a, **k))


def empty_like(*a, **k):
    return _fill(_empty_like(*a, **k))


torch.empty, torch.empty_like = empty, empty_like

import pytest  # noqa: E402

sys.exit(pytest.main(["-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + sys.argv[2:]))
