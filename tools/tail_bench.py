"""lstep_tail_fwd (one fused launch) vs the library-GEMM tail: max difference and time.  usage: python tools/tail_bench.py [rows]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat

dev = "cuda"
torch.manual_seed(0)
lib = nat.load_library()


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


for m in [int(a) for a in sys.argv[1:]] or [49152, 65536, 600, 37, 16384]:
    xe = torch.randn(m, 272, device=dev); xp = torch.randn(m, 272, device=dev)
    c1 = torch.zeros(m, 624, device=dev); c2 = torch.zeros(m, 352, device=dev)
    c1[:, :176] = torch.randn(m, 176, device=dev); c2[:, :176] = 0.1 * torch.randn(m, 176, device=dev)
    sc = 0.06
    w1, b1 = sc * torch.randn(272, 272, device=dev), sc * torch.randn(272, device=dev)
    wn1, bn1 = sc * torch.randn(176, 272, device=dev), sc * torch.randn(176, device=dev)
    wq, bq = sc * torch.randn(176, 352, device=dev), sc * torch.randn(176, device=dev)
    wall, ball = sc * torch.randn(176, 624, device=dev), sc * torch.randn(176, device=dev)
    out = torch.empty(m, 176, device=dev)

    def fused():
        nat.check(lib.lstep_tail_fwd(nat.ptr(xe), 272, nat.ptr(xp), 272, nat.ptr(c1), nat.ptr(c2), nat.ptr(out), nat.ptr(w1), nat.ptr(b1),
                                     nat.ptr(wn1), nat.ptr(bn1), nat.ptr(wq), nat.ptr(bq), nat.ptr(wall), nat.ptr(ball), m, nat.current_stream()))

    def ref(dt=torch.float32):
        f = lambda t: t.to(dt)
        h1 = torch.relu(f(xe) @ f(w1).t() + f(b1))
        p1 = torch.relu(f(xp) @ f(wn1).t() + f(bn1))
        own = f(c2[:, :176])
        q = own + torch.tanh(torch.cat([own, p1], 1) @ f(wq).t() + f(bq))
        o = torch.cat([f(c1[:, :176]), h1, q], 1) @ f(wall).t() + f(ball)
        return h1, p1, q, o

    fused()
    torch.cuda.synchronize()
    h1, p1, q, o = ref(torch.float64)
    errs = [(c1[:, 176:448].double() - h1).abs().max().item(), (c2[:, 176:].double() - p1).abs().max().item(),
            (c1[:, 448:].double() - q).abs().max().item(), (out.double() - o).abs().max().item()]
    r32 = ref()
    errs32 = [(r32[0].double() - h1).abs().max().item(), (r32[3].double() - o).abs().max().item()]
    us = timeit(fused)
    us_t = timeit(ref)
    fl = 2.0 * m * (272 * 272 + 176 * 272 + 176 * 352 + 176 * 624)
    print(f"m={m:6d} max|err| h1 {errs[0]:.1e} p1 {errs[1]:.1e} q {errs[2]:.1e} out {errs[3]:.1e} (torch fp32: h1 {errs32[0]:.1e} out {errs32[1]:.1e}) | "
          f"fused {us:7.1f} us {fl / us / 1e6:6.1f} TF/s | torch {us_t:7.1f} us")

print("---- backward")
for m in [int(a) for a in sys.argv[1:]] or [49152, 600, 37]:
    xe = torch.randn(m, 272, device=dev, requires_grad=True); xp = torch.randn(m, 272, device=dev, requires_grad=True)
    xn = torch.randn(m, 176, device=dev); own = (0.1 * torch.randn(m, 176, device=dev)).requires_grad_(True)
    sc = 0.06
    ws = [sc * torch.randn(*sh, device=dev) for sh in ((272, 272), (272,), (176, 272), (176,), (176, 352), (176,), (176, 624), (176,))]
    w1, b1, wn1, bn1, wq, bq, wall, ball = ws
    c1 = torch.zeros(m, 624, device=dev); c2 = torch.zeros(m, 352, device=dev)
    c1[:, :176] = xn; c2[:, :176] = own.detach()
    out = torch.empty(m, 176, device=dev)
    nat.check(lib.lstep_tail_fwd(nat.ptr(xe), 272, nat.ptr(xp), 272, nat.ptr(c1), nat.ptr(c2), nat.ptr(out), nat.ptr(w1), nat.ptr(b1),
                                 nat.ptr(wn1), nat.ptr(bn1), nat.ptr(wq), nat.ptr(bq), nat.ptr(wall), nat.ptr(ball), m, nat.current_stream()))
    go = torch.randn(m, 176, device=dev)
    # fp64 autograd reference
    d = lambda t: t.detach().double().requires_grad_(t.requires_grad)
    xe_, xp_, own_ = d(xe), d(xp), d(own)
    h1 = torch.relu(xe_ @ w1.double().t() + b1.double())
    p1 = torch.relu(xp_ @ wn1.double().t() + bn1.double())
    z = torch.cat([own_, p1], 1) @ wq.double().t() + bq.double()
    q = own_ + torch.tanh(z)
    o = torch.cat([xn.double(), h1, q], 1) @ wall.double().t() + ball.double()
    h1.retain_grad(); p1.retain_grad(); z.retain_grad()
    o.backward(go.double())
    w1t, wn1t, wqt, wallt = (w.t().contiguous() for w in (w1, wn1, wq, wall))
    dxe, dxp = torch.empty(m, 272, device=dev), torch.empty(m, 272, device=dev)
    down = torch.empty(m, 352, device=dev)
    dh1, dp1, dz = torch.empty(m, 272, device=dev), torch.empty(m, 176, device=dev), torch.empty(m, 176, device=dev)

    def bwd():
        nat.check(lib.lstep_tail_bwd(nat.ptr(go), nat.ptr(c1), nat.ptr(c2), nat.ptr(w1t), nat.ptr(wn1t), nat.ptr(wqt), nat.ptr(wallt),
                                     nat.ptr(dxe), nat.ptr(dxp), nat.ptr(down), 352, nat.ptr(dh1), nat.ptr(dp1), nat.ptr(dz), m, nat.current_stream()))
    bwd()
    torch.cuda.synchronize()
    # masks from the fp32 activations the kernel itself stored (an fp64 re-computation flips the relu of values within rounding of 0)
    h1m, p1m = c1[:, 176:448] > 0, c2[:, 176:] > 0
    errs = {"dxe": (dxe.double() - dh1.double() @ w1.double()).abs().max().item(), "dxp": (dxp.double() - dp1.double() @ wn1.double()).abs().max().item(),
            "down": (down[:, :176].double() - own_.grad).abs().max().item(), "dh1": ((dh1.double() - h1.grad) * h1m).abs().max().item() + (dh1 * ~h1m).abs().max().item(),
            "dp1": ((dp1.double() - p1.grad) * p1m).abs().max().item() + (dp1 * ~p1m).abs().max().item(), "dz": (dz.double() - z.grad).abs().max().item()}
    us = timeit(bwd)
    fl = 2.0 * m * (176 * 176 * 3 + 272 * 176 * 2 + 272 * 272)
    print(f"m={m:6d} max|err| " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()) + f" | fused bwd {us:7.1f} us {fl / us / 1e6:6.1f} TF/s")
