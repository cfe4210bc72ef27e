"""lstep_small_gemm vs torch.mm on the weight-composition shapes.  usage: python tools/small_mm_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat
dev = "cuda"


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (m, n, k) in [(172, 272, 172), (172, 172, 172), (172, 272, 272), (272, 272, 172)]:
    a, b = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev)
    at = torch.randn(k, m, device=dev).t()
    print(f"[{m}x{k}]@[{k}x{n}]: native {timeit(lambda: nat.small_mm(a, b)):6.1f} us (A transposed view {timeit(lambda: nat.small_mm(at, b)):6.1f}) | torch.mm {timeit(lambda: a @ b):6.1f} us (transposed {timeit(lambda: at @ b):6.1f})")
