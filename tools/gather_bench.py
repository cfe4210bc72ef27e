"""Gather stage in isolation on a c4-shaped graph (1 M nodes / 20 M edges, 49 152 rows, K = 20, time_gap = 2000): forward kernel and the
backward kernel in the engine's mode (slot dots + spliced-row hits, no atomics), HIP events around 10 launches each.
usage: python tools/gather_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat
from lstep_amd.sampler import NeighborSampler

dev = torch.device("cuda", 0)
nat.LIB_PATH = os.environ.get("LSTEP_LIB", nat.LIB_PATH)       # A/B against another build of the library
lib = nat.load_library()
n, e, B, K, G, D, F = 1_000_000, 20_000_000, 49152, 20, 2000, 100, int(os.environ.get("GATHER_F", "172"))      # GATHER_F=192: 768-byte rows, whole 128-byte lines
LE, LN = (D + F + 15) // 16 * 16, (F + 15) // 16 * 16
gen = torch.Generator(device=dev); gen.manual_seed(0)
src = torch.randint(1, n + 1, (e,), generator=gen, device=dev)
dst = torch.randint(1, n + 1, (e,), generator=gen, device=dev)
ts = torch.sort(torch.rand(e, dtype=torch.float64, generator=gen, device=dev) * 2e8).values
eid = torch.arange(1, e + 1, device=dev)
sampler = NeighborSampler.from_device_edges(src, dst, eid, ts, n)
ids = torch.cat([src[e // 2:e // 2 + B // 3], dst[e // 2:e // 2 + B // 3], torch.randint(1, n + 1, (B // 3,), generator=gen, device=dev)])
times = ts[e // 2:e // 2 + B // 3].repeat(3)
tw = (1.0 / 10 ** torch.linspace(0, 9, D, device=dev)).float(); tb = torch.zeros(D, device=dev); aw = torch.rand(K, device=dev)
node_raw = torch.randn(n + 1, F, device=dev); edge_raw = torch.randn(e + 1, F, device=dev); pe = torch.randn(n + 1, F, device=dev)
oe = torch.empty(B, LE, device=dev); on = torch.empty(B, LN, device=dev); op = torch.empty(B, LE, device=dev); os_ = torch.empty(B, LN, device=dev)
cnt = torch.empty(B, dtype=torch.int32, device=dev)
g_edge = torch.randn(B, LE, device=dev); g_pe = torch.randn(B, LE, device=dev); g_self = torch.randn(B, LN, device=dev)
slot_of = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
bn = torch.unique(ids[:2 * (B // 3)])
slot_of[bn] = torch.arange(bn.numel(), dtype=torch.int32, device=dev)
slot_dot = torch.empty(B, K, device=dev); hits = torch.empty(B, K, dtype=torch.int32, device=dev)
grad_rows = torch.zeros(bn.numel(), F, device=dev)


def fwd():
    nat.check(lib.lstep_gather_aggregate_fwd(sampler.csr, nat.ptr(node_raw), nat.ptr(edge_raw), nat.ptr(pe), F, F, nat.ptr(tw), nat.ptr(tb), D,
                                             nat.ptr(aw), nat.ptr(ids), nat.ptr(times), B, K, G, 3, nat.ptr(oe), nat.ptr(on), nat.ptr(op), nat.ptr(os_),
                                             LE, LN, LE, LN, nat.ptr(cnt), nat.current_stream()))


def bwd():
    nat.check(lib.lstep_gather_aggregate_bwd(sampler.csr, nat.ptr(edge_raw), F, F, nat.ptr(tw), nat.ptr(tb), D, nat.ptr(ids), nat.ptr(times),
                                             nat.ptr(cnt), B, K, nat.ptr(g_edge), nat.ptr(g_pe), nat.ptr(g_self), LE, LE, LN, nat.ptr(slot_of),
                                             nat.ptr(slot_dot), nat.ptr(grad_rows), nat.ptr(hits), nat.current_stream()))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms_f = timed(fwd)
c = cnt.long(); k, v = c.clamp(max=K), c.clamp(max=G)
fb = (4 * F * (2 * k + v + 3) + 24 * k + 8 * v).sum().item()
print(f"gather fwd: {ms_f * 1e3:7.1f} us  {fb / 1e9:.3f} GB algorithmic -> {fb / ms_f / 1e9:.2f} TB/s")
ms_b = timed(bwd)
bb = (688 * k + 16 * k + 4 * 2 * K).sum().item() + B * (272 + 272 + 176) * 4
print(f"gather bwd: {ms_b * 1e3:7.1f} us  {bb / 1e9:.3f} GB algorithmic -> {bb / ms_b / 1e9:.2f} TB/s   (mean k = {k.float().mean().item():.1f})")
