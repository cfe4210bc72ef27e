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
marks.clear()
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
