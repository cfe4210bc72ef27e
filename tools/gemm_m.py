"""Micro-benchmark: how the row count M changes hipBLASLt's fp32 throughput for the update_pe MLP shapes,
and whether fixed-size row blocks (looped or batched) give a steadier rate."""
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
F = torch.nn.functional
for m in (16384, 32768, 49152, 65536, 131072):
    x1 = torch.randn(m, 288, device=dev); w1 = torch.randn(176, 288, device=dev); b1 = torch.randn(176, device=dev)
    x2 = torch.randn(m, 176, device=dev); w2 = torch.randn(176, 176, device=dev)
    t1 = timeit(lambda: F.linear(x1, w1, b1)); t2 = timeit(lambda: F.linear(x2, w2, b1))
    print(f"M={m:7d}: 288->176 {t1:7.1f} us ({2*m*288*176/t1/1e6:6.1f} TF)   176->176 {t2:7.1f} us ({2*m*176*176/t2/1e6:6.1f} TF)")
m = 278528
x1 = torch.randn(m, 288, device=dev); w1 = torch.randn(176, 288, device=dev); b1 = torch.randn(176, device=dev)
x2 = torch.randn(m, 176, device=dev); w2 = torch.randn(176, 176, device=dev)
for blk in (16384, 32768, 65536, 139264):
    def loop1():
        return [F.linear(x1[i:i + blk], w1, b1) for i in range(0, m, blk)]
    def loop2():
        return [F.linear(x2[i:i + blk], w2, b1) for i in range(0, m, blk)]
    t1, t2 = timeit(loop1), timeit(loop2)
    print(f"M={m} in blocks of {blk:6d}: 288->176 {t1:7.1f} us ({2*m*288*176/t1/1e6:6.1f} TF)   176->176 {t2:7.1f} us ({2*m*176*176/t2/1e6:6.1f} TF)")
    if m % blk == 0:
        c = m // blk
        t1 = timeit(lambda: torch.baddbmm(b1, x1.view(c, blk, 288), w1.t().expand(c, 288, 176)))
        t2 = timeit(lambda: torch.baddbmm(b1, x2.view(c, blk, 176), w2.t().expand(c, 176, 176)))
        print(f"   batched ({c} x {blk}): 288->176 {t1:7.1f} us ({2*m*288*176/t1/1e6:6.1f} TF)   176->176 {t2:7.1f} us ({2*m*176*176/t2/1e6:6.1f} TF)")
