"""History filter kernels in isolation on a c4-sized ring (1 M rows x 102 slots x 172 floats = 70 GB): every snapshot is a clone of the one
before with a fraction `p` of its rows rewritten and marked in the change mask (p = 0.21 is what the c4 workload produces).  Prints the time
of the dense and of the run kernels (forward, backward) and checks that they agree.
usage: python tools/history_bench.py [p] [rows]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd.engine import HistoryRing
from lstep_amd.model import _HistoryFilter

p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.21
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_001
dev = torch.device("cuda", 0)
T, P, U = 100, 172, 45_000
ring = HistoryRing(rows, P, T, dev)
g = torch.Generator(device=dev); g.manual_seed(0)
for i in range(T):
    ring.start, ring.len = 0, i
    cur = ring.spare()
    if i == 0:
        cur.normal_(0.0, 0.1, generator=g)
        ring.begin_slot(all_changed=True)
    else:
        cur.copy_(ring.buf[i - 1])
        ring.begin_slot()
        ids = torch.nonzero(torch.rand(rows, device=dev, generator=g) < p).reshape(-1)
        cur[ids] = torch.randn(ids.numel(), P, device=dev, generator=g) * 0.1
        ring.mark(ids)
ring.start, ring.len = 0, T
ids = torch.sort(torch.randperm(rows - 1, device=dev, generator=g)[:U] + 1).values
coef = (torch.randn(T, P, device=dev, generator=g) / T).requires_grad_(True)
gout = torch.randn(U, P, device=dev, generator=g)


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


res = {}
for name, mask in (("dense", None), ("runs", ring.mask)):
    out = _HistoryFilter.apply(coef, ring.buf, ring.geom(), ids, mask)
    coef.grad = None
    out.backward(gout)
    res[name] = (out.detach().clone(), coef.grad.clone())
    t_f = timeit(lambda: _HistoryFilter.apply(coef.detach(), ring.buf, ring.geom(), ids, mask))

    def fb():
        o = _HistoryFilter.apply(coef, ring.buf, ring.geom(), ids, mask)
        o.backward(gout)
    t_fb = timeit(fb)
    print(f"{name:5s}: forward {t_f:7.1f} us   forward + backward {t_fb:7.1f} us   (backward ~ {t_fb - t_f:7.1f} us)")
do, dg = (res["dense"][0] - res["runs"][0]).abs().max().item(), (res["dense"][1] - res["runs"][1]).abs().max().item()
print(f"max |dense - runs|: out {do:.2e} (|out| max {res['dense'][0].abs().max().item():.2e}), dcoef {dg:.2e} (|dcoef| max {res['dense'][1].abs().max().item():.2e})")
bits = ring.mask.view(torch.int32)
print(f"rows {rows}, U {U}, p {p}: mean runs per node {1 + (T - 1) * p:.1f}")
