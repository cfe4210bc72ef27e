#!/usr/bin/env python3
"""Build the checked twin of the library (``-DLSTEP_BOUNDS_CHECK=1`` -> l-step_amd/csrc/liblstep_hip_checked.so) and say how to use it.

    python tools/build_checked.py
    LSTEP_LIB=$PWD/l-step_amd/csrc/liblstep_hip_checked.so python -m pytest tests -m gpu -x -q

In a checked build every id-indexed load of the kernels (neighbour / edge / node ids of the gather stage, row ids of update_pe, of the
history filter, of the loss and of the row scatters) is compared with its table's row count; an id out of range reads the padding row
instead of faulting the GPU and is recorded (which load, the id, the limit).  tests/conftest.py asks for that record behind every GPU
test when the loaded library is a checked build: ONE ordinary run of the suite names the kernel."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lstep_amd import _native as nat  # noqa: E402

if __name__ == "__main__":
    print(nat.build_checked_library(force="--force" in sys.argv))
