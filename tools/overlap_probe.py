"""Do a matrix-core kernel and an HBM-bound kernel overlap when launched on two streams?  lstep_linear_wgrad beside a device-to-device
copy: the default tiling (6 x 4 tiles per wave, <= 256 registers: a second wave fits on every SIMD) against LSTEP_WGRAD_BIG=1 (one
512-register wave per SIMD, which owns the chip while it runs).  Measured on MI355X: big 497 + 438 us alone, 1073 us together (no
overlap at all); small 514 + 436 us alone, 737 us together.  usage: python tools/overlap_probe.py ; LSTEP_WGRAD_BIG=1 python tools/overlap_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat

dev = "cuda"
m, n, k = 49152, 176, 272
dy, x = torch.randn(m, n, device=dev), torch.randn(m, k, device=dev)
a, b = torch.empty(256 << 20, dtype=torch.float32, device=dev), torch.empty(256 << 20, dtype=torch.float32, device=dev)   # 1 GiB each
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
reps = 6


def wg():
    for _ in range(reps):
        nat.linear_wgrad(dy, x)


def cp():
    b.copy_(a)


def timed(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    if fa:
        with torch.cuda.stream(s1):
            fa()
    if fb:
        with torch.cuda.stream(s2):
            fb()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for _ in range(2):
    timed(wg, cp)
ta, tb, tab = min(timed(wg, None) for _ in range(3)), min(timed(None, cp) for _ in range(3)), min(timed(wg, cp) for _ in range(3))
print(f"LSTEP_WGRAD_BIG={os.environ.get('LSTEP_WGRAD_BIG')}: {reps} x wgrad alone {ta:.0f} us, 2 GiB copy alone {tb:.0f} us, together {tab:.0f} us "
      f"(sum {ta + tb:.0f}, max {max(ta, tb):.0f})")
