"""lstep_update_rows in isolation: time and TFLOP/s at the row counts of update_pe's two phases.  usage: python tools/update_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat

dev = "cuda"
lib = nat.load_library()
torch.manual_seed(0)
N = 1_000_001
table = torch.randn(N, 172, device=dev) * 0.1
w1, b1 = 0.05 * torch.randn(176, 272, device=dev), torch.zeros(176, device=dev)
w2, b2 = 0.05 * torch.randn(176, 176, device=dev), torch.zeros(176, device=dev)
ws, bs = 0.05 * torch.randn(176, 176, device=dev), torch.zeros(176, device=dev)
for n, with_self in ((45000, True), (213000, False), (262144, False), (65536, False)):
    agg = torch.randn(n, 272, device=dev)
    ids = torch.randperm(N - 1, device=dev)[:n] + 1

    def run():
        nat.check(lib.lstep_update_rows(nat.ptr(agg), 272, nat.ptr(ids), n, nat.ptr(w1), nat.ptr(b1), nat.ptr(w2), nat.ptr(b2),
                                        nat.ptr(ws) if with_self else None, nat.ptr(bs) if with_self else None, nat.ptr(table), None, 172,
                                        None, None, 1, 0, nat.current_stream()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * n * (272 * 176 + 176 * 176 + (176 * 176 if with_self else 0))
    print(f"n={n:7d} self={with_self}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
