"""Where do the framework fill / copy / element-wise launches of one training iteration come from?  Runs a few launch-by-launch iterations of a
small workload under torch.profiler with Python stacks and prints, per aten op that launches a kernel, the innermost frames inside this repo.
usage: python tools/find_fills.py [workload]"""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from lstep_amd.workload import build_workload, evolve_history
from lstep_amd.optim import FusedAdam

name = sys.argv[1] if len(sys.argv) > 1 else "wikipedia"
dev = torch.device("cuda", 0)
wl = build_workload(name, dev, seed=0)
eng, model = wl.engine, wl.model
eng.use_step_graph = False
opt = FusedAdam(model.parameters(), lr=1e-4)
start = wl.num_edges // 2
model.eval()
evolve_history(eng, wl.stream, start, wl.batch, wl.num_nodes)
model.train()
B = wl.batch


def step(i):
    lo = start + (wl.T + i) * B
    src, dst, ts, eid = wl.stream.batch(lo, lo + B)
    neg = wl.negatives(i) if hasattr(wl, "negatives") else torch.randint(1, wl.num_nodes + 1, (B,), device=dev)
    return eng.train_iteration(opt, 1000 + i, src, dst, ts, eid, neg)


for i in range(4):
    step(i)
torch.cuda.synchronize()

# (1) Python-level callers of the allocation / fill / copy entry points
import traceback
calls = collections.Counter()


def wrap(owner, attr):
    orig = getattr(owner, attr)

    def f(*a, **k):
        fr = [x for x in traceback.extract_stack()[:-1] if "l-step_amd" in x.filename or "lstep_amd" in x.filename]
        calls[(attr, " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-2:]))] += 1
        return orig(*a, **k)
    setattr(owner, attr, f)
    return orig


saved = [(o, a, wrap(o, a)) for o, a in ((torch, "zeros"), (torch, "zeros_like"), (torch, "cat"), (torch, "full"), (torch.Tensor, "zero_"),
                                          (torch.Tensor, "fill_"), (torch.Tensor, "copy_"), (torch.Tensor, "contiguous"), (torch.Tensor, "sum"),
                                          (torch.Tensor, "index_fill_"), (torch.Tensor, "to"), (torch.Tensor, "clone"), (torch, "mm"))]
step(4)
torch.cuda.synchronize()
for o, a, orig in saved:
    setattr(o, a, orig)
print("---- Python-level calls in one iteration")
for (op, where), n in sorted(calls.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:3d} x {op:12s} {where}")

# (2) every aten op of one iteration (names and counts), autograd-internal ones included; for the fill / zero / copy family the innermost
#     Python frames (C++-internal calls show the frame of the Python call that led into them)
try:
    cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
except Exception:  # noqa: BLE001
    cfg = None
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=cfg) as prof:
    step(5)
torch.cuda.synchronize()
ops = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::"))
print("---- aten ops in one iteration")
print(", ".join(f"{n} x {k[6:]}" for k, n in ops.most_common(60)))
print("---- stacks of the fill / zero / copy family")
where = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::zeros_like", "aten::cat", "aten::mul", "aten::add_", "aten::add"):
        st = [f for f in (ev.stack or []) if "l-step_amd" in f or "lstep_amd" in f or "torch/autograd" in f or "optim" in f]
        where[(ev.name, " <- ".join(f.strip().split("/")[-1][:60] for f in st[:3]) or "(no Python frame: autograd engine / C++)")] += 1
for (op, w), n in sorted(where.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:3d} x {op:14s} {w}")
