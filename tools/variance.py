"""Step-time stability of the engine: six consecutive blocks of 20 training iterations.  NOGC=1: collector off; NOGC=2: gc.freeze() after
warm-up (what bench.py does).  Without either, one block in a run contains a full cyclic collection (35-50 ms).
usage: [NOGC=0|1|2] python tools/variance.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd.workload import build_workload
from lstep_amd.optim import FusedAdam
B=16384
dev = torch.device("cuda", 0)
wl = build_workload("synth-1M-20M", dev, time_gap=2000, batch=B, seed=0)
eng, model = wl.engine, wl.model
model.train()
opt = FusedAdam(model.parameters(), lr=1e-4)
gen = torch.Generator(device=dev); gen.manual_seed(1)
start = wl.num_edges // 2
from lstep_amd.workload import evolve_history
evolve_history(eng, wl.stream, start, B, wl.num_nodes)   # the history the algorithm itself produces (as bench.py does)
def run(i):
    lo = start + i * B
    src, dst, ts, eid = wl.stream.batch(lo, lo + B)
    neg = torch.randint(1, wl.num_nodes + 1, (B,), generator=gen, device=dev)
    eng.train_iteration(opt, 1000 + i, src, dst, ts, eid, neg)
for i in range(3): run(i)
import gc
if os.environ.get("NOGC") == "1": gc.disable()
if os.environ.get("NOGC") == "2": gc.collect(); gc.freeze()
torch.cuda.synchronize()
ts=[]
for rep in range(6):
    t0=time.perf_counter()
    for i in range(20): run(3+rep*20+i)
    torch.cuda.synchronize()
    ts.append((time.perf_counter()-t0)/20*1e3)
print("per-20-step ms:", [round(t,2) for t in ts])
