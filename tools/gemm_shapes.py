"""Micro-benchmark: fp32 torch (hipBLASLt) GEMM time for the dense-tail shapes, with and without padding of the
awkward 172/272 widths.  usage: python tools/gemm_shapes.py"""
import torch, time
dev = "cuda"
def bench(m, k, n, reps=20, bwd=False):
    x = torch.randn(m, k, device=dev, requires_grad=bwd)
    w = torch.randn(n, k, device=dev, requires_grad=bwd)
    b = torch.randn(n, device=dev)
    for _ in range(3):
        y = torch.nn.functional.linear(x, w, b)
        if bwd: y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        y = torch.nn.functional.linear(x, w, b)
        if bwd: y.backward(torch.ones_like(y))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * m * k * n * (3 if bwd else 1)
    return ms, fl / ms / 1e9
for m in (49152, 262144):
    for (k, n) in ((272, 272), (288, 288), (272, 172), (272, 176), (272, 192), (288, 192), (172, 172), (176, 176), (192, 192), (344, 172), (352, 176), (384, 192),
                   (616, 172), (640, 192), (640, 176), (444, 172)):
        f = bench(m, k, n)
        b = bench(m, k, n, bwd=True) if m == 49152 else (0, 0)
        print(f"M={m:7d} K={k:4d} N={n:4d}  fwd {f[0]*1e3:8.1f} us {f[1]:6.1f} TF/s   fwd+bwd {b[0]*1e3:8.1f} us {b[1]:6.1f} TF/s")
