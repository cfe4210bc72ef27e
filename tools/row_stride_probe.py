"""Does a 704-byte (64-B aligned) row stride beat the reference-shaped 688-byte rows in the gather kernel?  The kernel is run on
the same graph with feature / PE width 172 (688-B rows straddling 128-B lines at arbitrary offsets) and 176 (704-B rows: every row
starts on a 64-B boundary and covers exactly six 128-B lines).  Measured on MI355X: 0.518 vs 0.511 ms per 49 152-row launch, 5.65 vs
5.72 G rows/s -- the kernel is bound by the rate of scattered ~700-byte row reads, not by the 19 % of line-straddle bytes, so padding
the tables is not worth breaking zero-copy acceptance of reference-shaped [N+1, 172] tensors.  usage: python tools/row_stride_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lstep_amd import _native as nat
from lstep_amd.sampler import NeighborSampler

dev = torch.device("cuda", 0)
lib = nat.load_library()
n, e, B, K, G, D = 1_000_000, 20_000_000, 49152, 20, 2000, 100
gen = torch.Generator(device=dev); gen.manual_seed(0)
src = torch.randint(1, n + 1, (e,), generator=gen, device=dev)
dst = torch.randint(1, n + 1, (e,), generator=gen, device=dev)
ts = torch.sort(torch.rand(e, dtype=torch.float64, generator=gen, device=dev) * 2e8).values
eid = torch.arange(1, e + 1, device=dev)
sampler = NeighborSampler.from_device_edges(src, dst, eid, ts, n)
ids = torch.cat([src[e // 2:e // 2 + B // 3], dst[e // 2:e // 2 + B // 3], torch.randint(1, n + 1, (B // 3,), generator=gen, device=dev)])
times = ts[e // 2:e // 2 + B // 3].repeat(3)
tw = torch.rand(D, device=dev); tb = torch.zeros(D, device=dev); aw = torch.rand(K, device=dev)
for F in (172, 176):
    node_raw = torch.randn(n + 1, F, device=dev); edge_raw = torch.randn(e + 1, F, device=dev); pe = torch.randn(n + 1, F, device=dev)
    oe = torch.empty(B, F + D, device=dev); on = torch.empty(B, 176, device=dev); op = torch.empty(B, F + D, device=dev); os_ = torch.empty(B, 176, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)

    def run():
        nat.check(lib.lstep_gather_aggregate_fwd(sampler.csr, nat.ptr(node_raw), nat.ptr(edge_raw), nat.ptr(pe), F, F, nat.ptr(tw), nat.ptr(tb), D,
                                                 nat.ptr(aw), nat.ptr(ids), nat.ptr(times), B, K, G, 3, nat.ptr(oe), nat.ptr(on), nat.ptr(op), nat.ptr(os_),
                                                 F + D, 176, F + D, 176, nat.ptr(cnt), nat.current_stream()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    c = cnt.long()
    k, v = c.clamp(max=K), c.clamp(max=G)
    rows = (2 * k + v + 3).sum().item()
    print(f"row width {F} ({4 * F} B): {ms:.3f} ms per launch, {rows} rows moved -> {rows * 4 * F / ms / 1e6:.0f} GB/s of row bytes, {rows / ms / 1e6:.2f} G rows/s")
    del node_raw, edge_raw, pe
