"""Host-side cost of one training iteration: cProfile of the main thread over N iterations at a small batch (GPU work is
short, so the wall time is the host's), update_pe inline (LSTEP_NO_OVERLAP=1) so its Python shows up too.
usage: LSTEP_NO_OVERLAP=1 python tools/host_profile.py [batch] [iters]      (DIST=1: the distributed engine on one rank, with look-ahead)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd.workload import build_workload
from lstep_amd.optim import FusedAdam

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
DIST = os.environ.get("DIST") == "1"
wl = build_workload("synth-1M-20M", dev, time_gap=2000, batch=B, seed=0, sharded=DIST)
eng, model = wl.engine, wl.model
model.train()
opt = FusedAdam(model.parameters(), lr=1e-4)
if DIST:
    import torch.distributed as dist
    from lstep_amd.parallel import DistributedLstep
    from lstep_amd.workload import prefill_distributed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=dev)
    eng = DistributedLstep(eng, opt)
    prefill_distributed(eng, seed=0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
start = wl.num_edges // 2
from lstep_amd.workload import evolve_history
evolve_history(eng, wl.stream, start, B, wl.num_nodes)   # the history the algorithm itself produces (as bench.py does)


def step(i):
    lo = start + i * B
    src, dst, ts, eid = wl.stream.batch(lo, lo + B)
    neg = torch.randint(1, wl.num_nodes + 1, (B,), generator=gen, device=dev)
    nxt = wl.stream.batch(lo + B, lo + 2 * B)[:2] if DIST else None
    return eng.train_iteration(opt, 1000 + i, src, dst, ts, eid, neg, lookahead=nxt)


for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(iters):
    step(5 + i)
pr.disable()
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0) / iters:.2f} ms/iter (with cProfile overhead)")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
