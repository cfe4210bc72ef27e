"""Host-side phase times of the engine's training iteration (overlap mode): forward enqueue, backward (host), join of the update
thread, optimiser; plus the GPU-complete time per iteration.  usage: python tools/phase_times.py [batch] [iters]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd.workload import build_workload
from lstep_amd.optim import FusedAdam
from lstep_amd import engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
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
orig_join = threading.Thread.join
orig_step = FusedAdam.step


def backward(self, *a, **k):
    marks.append(("fwd_done", time.perf_counter()))
    r = orig_backward(self, *a, **k)
    marks.append(("bwd_done", time.perf_counter()))
    return r


def join(self, *a, **k):
    r = orig_join(self, *a, **k)
    if self.name == "lstep-update-pe":
        marks.append(("joined", time.perf_counter()))
    return r


def step(self):
    r = orig_step(self)
    marks.append(("stepped", time.perf_counter()))
    return r


torch.Tensor.backward = backward
threading.Thread.join = join
FusedAdam.step = step


def run(i):
    lo = start + i * B
    src, dst, ts, eid = wl.stream.batch(lo, lo + B)
    neg = torch.randint(1, wl.num_nodes + 1, (B,), generator=gen, device=dev)
    marks.append(("begin", time.perf_counter()))
    eng.train_iteration(opt, 1000 + i, src, dst, ts, eid, neg)
    marks.append(("end", time.perf_counter()))


for i in range(5):
    run(i)
torch.cuda.synchronize()
marks.clear()
t0 = time.perf_counter()
for i in range(iters):
    run(5 + i)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / iters * 1e3
acc, prev = {}, None
for name, t in marks:
    if prev is not None and name != "begin":
        acc[f"{prev[0]}->{name}"] = acc.get(f"{prev[0]}->{name}", 0.0) + (t - prev[1])
    prev = (name, t)
print(f"wall {wall:.2f} ms/iter")
for k, v in acc.items():
    print(f"  {k:24s} {v / iters * 1e3:7.3f} ms")
