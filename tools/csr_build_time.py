"""Time the GPU CSR build (NeighborSampler.from_device_edges) at the benchmark scale."""
import sys, os, time
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch
from lstep_amd.sampler import NeighborSampler
dev = "cuda"
for n, e in ((1_000_000, 20_000_000), (4_000_000, 100_000_000)):
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    src = torch.randint(1, n + 1, (e,), generator=gen, device=dev); dst = torch.randint(1, n + 1, (e,), generator=gen, device=dev)
    ts = torch.sort(torch.rand(e, dtype=torch.float64, generator=gen, device=dev) * 1e7).values
    eid = torch.arange(1, e + 1, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = NeighborSampler.from_device_edges(src, dst, eid, ts, n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n} nodes / {e} edges: CSR ({s.nnz} entries) built on the GPU in {dt*1e3:.0f} ms, peak memory {torch.cuda.max_memory_allocated()/1e9:.1f} GB")
    del s, src, dst, ts, eid
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
