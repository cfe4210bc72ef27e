"""Micro-benchmark: input-gradient GEMMs dX[M,K] = dY[M,N] @ W[N,K] of the dense tail (fp32, hipBLASLt via torch)."""
import torch
dev = "cuda"
def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
m = 49152
for (n, k) in ((288, 288), (176, 288), (176, 352), (176, 640), (176, 464), (176, 344)):
    dy = torch.randn(m, n, device=dev); w = torch.randn(n, k, device=dev); wt = w.t().contiguous()
    t1 = timeit(lambda: dy @ w)
    t2 = timeit(lambda: torch.nn.functional.linear(dy, wt))          # same product with W stored transposed
    npad = (n + 31) // 32 * 32
    dyp = torch.zeros(m, npad, device=dev); dyp[:, :n] = dy; wp = torch.zeros(npad, k, device=dev); wp[:n] = w
    t3 = timeit(lambda: dyp @ wp)
    fl = 2.0 * m * n * k
    print(f"dY[{m},{n}] @ W[{n},{k}]: NN {t1:6.1f} us ({fl/t1/1e6:5.1f} TF)  via W^T stored {t2:6.1f} us ({fl/t2/1e6:5.1f} TF)  N padded to {npad}: {t3:6.1f} us ({fl/t3/1e6:5.1f} TF)")
