"""lstep_update_rows with a capacity-sized launch and a device-resident live count vs an exact-sized launch.  usage: python tools/update_cap_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat

dev = "cuda"
lib = nat.load_library()
torch.manual_seed(0)
N = 1_000_001
table = torch.randn(N, 172, device=dev) * 0.1
w1, b1 = torch.randn(176, 272, device=dev) * 0.05, torch.zeros(176, device=dev)
w2, b2 = torch.randn(176, 176, device=dev) * 0.05, torch.zeros(176, device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for live_n, cap in ((290_000, 290_000), (290_000, 655_361), (32_500, 32_768), (1, 1)):
    agg = torch.randn(cap, 272, device=dev)
    ids = torch.randperm(N - 1, device=dev)[:cap] + 1
    live = torch.tensor([live_n], dtype=torch.int32, device=dev)

    def run(use_live):
        nat.check(lib.lstep_update_rows(nat.ptr(agg), 272, nat.ptr(ids), cap if use_live else live_n, nat.ptr(w1), nat.ptr(b1), nat.ptr(w2), nat.ptr(b2),
                                        None, None, nat.ptr(table), None, 172, nat.ptr(live) if use_live else None, None, 1, 0, nat.current_stream()))
    a, b = timeit(lambda: run(False)), timeit(lambda: run(True))
    fl = live_n * 158e3
    print(f"live {live_n:7d} capacity {cap:7d}: exact launch {a:8.1f} us ({fl / a / 1e6:5.1f} TF/s) | capacity launch + device count {b:8.1f} us")
