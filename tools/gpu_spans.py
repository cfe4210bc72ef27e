"""GPU-side spans of a training iteration on the critical stream (HIP events): forward, backward, optimiser + turnaround.
usage: python tools/gpu_spans.py [batch] [iters]"""
import os, sys, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd.workload import build_workload
from lstep_amd.optim import FusedAdam

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
wl = build_workload("synth-1M-20M", dev, time_gap=2000, batch=B, seed=0)
eng, model = wl.engine, wl.model
model.train()
opt = FusedAdam(model.parameters(), lr=1e-4)
gen = torch.Generator(device=dev); gen.manual_seed(1)
start = wl.num_edges // 2
from lstep_amd.workload import evolve_history
evolve_history(eng, wl.stream, start, B, wl.num_nodes)   # the history the algorithm itself produces (as bench.py does)
marks = []
orig_backward = torch.Tensor.backward
orig_step = FusedAdam.step


def ev(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


def backward(self, *a, **k):
    ev("fwd_end")
    r = orig_backward(self, *a, **k)
    ev("bwd_launched")
    return r


from lstep_amd import model as M
orig_join = M.LSTEP.join_aux_stream
tails = []


def join_aux(self):
    M._flush_deferred()
    # where the other two streams are when the critical stream has finished its backward kernels
    e_main = torch.cuda.Event(enable_timing=True); e_main.record()
    e_aux = torch.cuda.Event(enable_timing=True); e_aux.record(M._aux_stream(dev))
    tails.append((e_main, e_aux, None))
    return orig_join(self)


M.LSTEP.join_aux_stream = join_aux
orig_tjoin = threading.Thread.join


def tjoin(self, *a, **k):
    r = orig_tjoin(self, *a, **k)
    if self.name == "lstep-update-pe" and tails:
        e_upd = torch.cuda.Event(enable_timing=True); e_upd.record(eng._update_stream)
        tails[-1] = (tails[-1][0], tails[-1][1], e_upd)
    return r


threading.Thread.join = tjoin


def step(self):
    ev("before_adam")
    r = orig_step(self)
    ev("after_adam")
    return r


torch.Tensor.backward = backward
FusedAdam.step = step


def run(i):
    lo = start + i * B
    src, dst, ts, eid = wl.stream.batch(lo, lo + B)
    neg = torch.randint(1, wl.num_nodes + 1, (B,), generator=gen, device=dev)
    ev("begin")
    eng.train_iteration(opt, 1000 + i, src, dst, ts, eid, neg, lookahead=wl.stream.batch(lo + B, lo + 2 * B)[:2])


for i in range(5):
    run(i)
torch.cuda.synchronize()
marks.clear(); tails.clear()
import gc; gc.collect(); gc.freeze()
for i in range(iters):
    run(5 + i)
torch.cuda.synchronize()
acc, prev = {}, None
for name, e in marks:
    if prev is not None:
        key = f"{prev[0]}->{name}"
        acc[key] = acc.get(key, 0.0) + prev[1].elapsed_time(e)
    prev = (name, e)
tot = 0.0
for k, v in acc.items():
    print(f"  {k:28s} {v / iters:7.3f} ms (GPU time between the two points of the critical stream)")
    tot += v / iters
print(f"  sum {tot:.3f} ms")
aux = sum(a.elapsed_time(b) for a, b, c in tails) / len(tails)
upd = sum(a.elapsed_time(c) for a, b, c in tails if c is not None) / max(1, sum(1 for t in tails if t[2] is not None))
print(f"  after the critical stream's last backward kernel: auxiliary stream finishes {aux:+.3f} ms later, update stream {upd:+.3f} ms later")
