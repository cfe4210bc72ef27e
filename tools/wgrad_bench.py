"""lstep_linear_wgrad vs torch (hipBLASLt): correctness against an fp64 product and time per call for the dense-tail shapes.
usage: python tools/wgrad_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat

dev = "cuda"
torch.manual_seed(0)


def timeit(fn, reps=20):
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


for (m, n, k) in ((49152, 176, 272), (49152, 272, 272), (49152, 176, 352), (49152, 176, 624), (32768, 172, 344), (49153, 176, 272), (4100, 176, 272), (1000, 64, 48),
                  (7, 16, 16), (16384, 176, 640)):
    dy = torch.randn(m, n, device=dev)
    x = torch.randn(m, k, device=dev)
    dw, db = nat.linear_wgrad(dy, x)
    ref = (dy.double().t() @ x.double())
    refb = dy.double().sum(0)
    t_dw = dy.t() @ x
    err = (dw.double() - ref).abs().max().item()
    err_t = (t_dw.double() - ref).abs().max().item()
    errb = (db.double() - refb).abs().max().item()
    us = timeit(lambda: nat.linear_wgrad(dy, x))
    c = 16 if m % 16 == 0 else 1
    us_t = timeit(lambda: (torch.bmm(dy.view(c, m // c, -1).transpose(1, 2), x.view(c, m // c, -1)).sum(0), dy.sum(0)))
    fl = 2.0 * m * n * k
    print(f"m={m:6d} n={n:4d} k={k:4d}  max|err| {err:.2e} (torch {err_t:.2e}) bias {errb:.2e} | lstep {us:7.1f} us {fl / us / 1e6:6.1f} TF/s | torch bmm+sum {us_t:7.1f} us {fl / us_t / 1e6:6.1f} TF/s")

# ---- the six products of a training step: one (partial, reduce) launch pair each against ONE batched pair (lstep_linear_wgrad_batch)
print("six products of a step, one by one vs batched (us):")
for m in (600, 1800, 12288, 49152):
    shapes = [(272, 272), (176, 272), (176, 352), (176, 624), (176, 176), (176, 352)]
    items = [(torch.randn(m if i != 4 else m // 3, n, device=dev), torch.randn(m if i != 4 else m // 3, k, device=dev), True, None) for i, (n, k) in enumerate(shapes)]
    outs = [(torch.empty(n, k, device=dev), torch.empty(n, device=dev)) for (n, k) in shapes]
    items = [(a, b, c, o) for (a, b, c, _), o in zip(items, outs)]
    t_seq = timeit(lambda: [nat.linear_wgrad(a, b, out=o) for a, b, _, o in items])
    t_bat = timeit(lambda: nat.linear_wgrad_batch(items))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        [nat.linear_wgrad(a, b, out=o) for a, b, _, o in items]
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        nat.linear_wgrad_batch(items)
    print(f"  m = {m:6d}: launch by launch {t_seq:7.1f} / {t_bat:7.1f}   graph replay {timeit(g.replay):7.1f} / {timeit(g2.replay):7.1f}")
