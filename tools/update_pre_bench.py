"""lstep_update_rows_pre in isolation (update_pe phase 2 at the c4 shape: ~290 k touched rows, mirror slot written too): the slab-chain kernel
(LSTEP_UPDATE_LDS=0) against the persistent LDS-resident kernel with 8 / 12 waves per workgroup.  usage: python tools/update_pre_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat

dev = "cuda"
lib = nat.load_library()
torch.manual_seed(0)
N, P, TD = 1_000_001, 172, 100
table = torch.randn(N, P, device=dev) * 0.1
mirror = torch.zeros(N, P, device=dev)
w1b, b1 = 0.05 * torch.randn(176, 112, device=dev), torch.zeros(176, device=dev)
w2, b2 = 0.05 * torch.randn(176, 176, device=dev), torch.zeros(176, device=dev)
for n in (290000, 131072, 65536):
    agg = torch.randn(n, 176 + TD, device=dev)
    ids = torch.randperm(N - 1, device=dev)[:n] + 1
    live = torch.tensor([n - 1000], dtype=torch.int32, device=dev)
    for lds, waves in (("0", "12"), ("1", "8"), ("1", "12")):
        os.environ["LSTEP_UPDATE_LDS"], os.environ["LSTEP_UPDATE_LDS_WAVES"] = lds, waves

        def run():
            nat.check(lib.lstep_update_rows_pre(nat.ptr(agg), 176 + TD, nat.ptr(ids), n, nat.ptr(w1b), nat.ptr(b1), nat.ptr(w2), nat.ptr(b2),
                                                nat.ptr(table), nat.ptr(mirror), P, TD, nat.ptr(live), None, 1, 0, nat.current_stream()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        fl = 2.0 * n * (112 * 176 + 176 * 176)
        print(f"n={n:7d} {'LDS-resident, ' + waves + ' waves' if lds == '1' else 'slab chain (S = 3)    '}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
