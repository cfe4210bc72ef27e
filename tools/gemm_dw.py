"""Micro-benchmark: weight-gradient GEMM dW[N,K] = dY[M,N]^T X[M,K] formulations in torch fp32 (hipBLASLt)."""
import torch
dev = "cuda"
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (m, k, n) in ((49152, 288, 288), (49152, 288, 176), (49152, 352, 176), (49152, 640, 176), (49152, 272, 272), (49152, 616, 172), (262144, 288, 176)):
    x = torch.randn(m, k, device=dev); dy = torch.randn(m, n, device=dev)
    xt = x.t().contiguous(); dyt = dy.t().contiguous()
    fl = 2.0 * m * k * n
    res = {
        "dy.t()@x": timeit(lambda: torch.mm(dy.t(), x)),
        "(x.t()@dy).t()": timeit(lambda: torch.mm(x.t(), dy)),
        "dyT_contig@x": timeit(lambda: torch.mm(dyt, x)),
        "dyT_contig@xT_contig.t()": timeit(lambda: torch.mm(dyt, xt.t())),
        "chunked8 bmm": timeit(lambda: torch.bmm(dy.view(8, m // 8, n).transpose(1, 2), x.view(8, m // 8, k)).sum(0)),
        "chunked32 bmm": timeit(lambda: torch.bmm(dy.view(32, m // 32, n).transpose(1, 2), x.view(32, m // 32, k)).sum(0)),
    }
    print(f"M={m} K={k} N={n}: " + "  ".join(f"{a}: {t:7.1f}us ({fl / t / 1e6:5.1f}TF)" for a, t in res.items()))
