"""Synthetic benchmark workloads of BASELINE.json, generated directly in HBM.

Uniform endpoints over 1..N, sorted float64 timestamps over ``1e6 * E / 1e5``, N(0,1) features with a zero padding
row, ``0.1 * N(0,1)`` PE history pre-filled to T snapshots (steady state: the history window is full).
The big tables never exist on the host: 20 M edges x 172 floats is 13.8 GB, the T = 100 history of 1 M nodes 69 GB.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import synth
from .engine import EdgeStream, LstepEngine
from .model import LSTEP, MergeLayer
from .sampler import NeighborSampler

WORKLOADS = {
    # name: (nodes, edges, batch, num_neighbors) -- BASELINE.json configs[0..3]; time_gap 2000 and T 100 are the reference defaults
    "enron": (184, 125_235, 200, 20),
    "wikipedia": (9_227, 157_474, 600, 20),
    "reddit": (10_984, 672_447, 4096, 32),
    "synth-1M-20M": (1_000_000, 20_000_000, 16384, 20),
    "synth-4M-100M": (4_000_000, 100_000_000, 16384, 20),
    "tiny": (2_000, 40_000, 256, 20),
}


@dataclass
class Workload:
    name: str
    num_nodes: int
    num_edges: int
    batch: int
    K: int
    G: int
    T: int
    stream: EdgeStream
    model: torch.nn.Sequential
    engine: LstepEngine
    sampler: NeighborSampler

    def describe(self) -> str:
        return (f"synthetic temporal graph {self.num_nodes} nodes / {self.num_edges} edges / feat_dim={synth.FEAT_DIM}, "
                f"batch_size={self.batch}, num_neighbors={self.K}, time_gap={self.G}, num_fft_batches={self.T}, "
                f"{getattr(self.sampler, 'sample_neighbor_strategy', 'recent')} sampling")


def build_hip_model(node_raw, edge_raw, sampler, K, T, state_dict=None, device="cuda"):
    """``nn.Sequential(LSTEP, MergeLayer)`` as the reference wraps it (train_LSTEP_link_prediction.py:140-142), optionally loaded from a
    reference-keyed ``state_dict`` (``strict=True``: same names, shapes and dtypes)."""
    bb = LSTEP(node_raw, edge_raw, sampler, sampler, pe_dim=synth.PE_DIM, num_neighbors=K, time_feat_dim=synth.TIME_DIM,
               num_fft_batches=T, device=device)
    pred = MergeLayer(synth.FEAT_DIM, synth.FEAT_DIM, synth.FEAT_DIM, 1).to(device)
    model = torch.nn.Sequential(bb, pred)
    if state_dict is not None:
        model.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)
    return model


def build_workload(name: str, device, time_gap: int = 2000, num_fft_batches: int = 100, seed: int = 0, batch: int = None,
                   sharded: bool = False, zipf: float = None, sampler: str = "recent") -> Workload:
    """``sharded=True`` (multi-GPU): no full history ring is allocated; ``prefill_distributed`` fills the owner shards."""
    n, e, b, k = WORKLOADS[name]
    if batch is not None:
        b = batch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    if zipf is None:
        src = torch.randint(1, n + 1, (e,), generator=gen, device=dev)
        dst = torch.randint(1, n + 1, (e,), generator=gen, device=dev)
    else:  # power-law node popularity (hub skew): P(node of rank r) ~ r^-zipf, ranks randomly assigned to ids
        prob = torch.arange(1, n + 1, device=dev, dtype=torch.float64) ** (-zipf)
        cdf = torch.cumsum(prob / prob.sum(), 0)
        perm = torch.randperm(n, generator=gen, device=dev) + 1
        draw = lambda: perm[torch.searchsorted(cdf, torch.rand(e, dtype=torch.float64, generator=gen, device=dev)).clamp(max=n - 1)]  # noqa: E731
        src, dst = draw(), draw()
    span = 1e6 * e / 1e5
    ts = torch.sort(torch.rand(e, dtype=torch.float64, generator=gen, device=dev) * span).values
    eid = torch.arange(1, e + 1, device=dev)
    strategy = sampler
    sampler = NeighborSampler.from_device_edges(src, dst, eid, ts, n, seed=0 if strategy != "recent" else None, sample_neighbor_strategy=strategy,
                                                time_scaling_factor=1e-6 if strategy == "time_interval_aware" else 0.0)
    node_raw = torch.randn((n + 1, synth.FEAT_DIM), generator=gen, device=dev)
    node_raw[0] = 0
    edge_raw = torch.randn((e + 1, synth.FEAT_DIM), generator=gen, device=dev)
    edge_raw[0] = 0
    torch.manual_seed(seed)
    bb = LSTEP(node_raw, edge_raw, sampler, sampler, pe_dim=synth.PE_DIM, num_neighbors=k, time_feat_dim=synth.TIME_DIM,
               num_fft_batches=num_fft_batches, device=dev)
    pred = MergeLayer(synth.FEAT_DIM, synth.FEAT_DIM, synth.FEAT_DIM, 1).to(dev)
    model = torch.nn.Sequential(bb, pred)
    eng = LstepEngine(bb, pred, k, time_gap, make_ring=not sharded)
    if not sharded:
        # steady state: the window already holds T snapshots
        ring = eng.ring
        for s in range(ring.T):
            ring.buf[s].normal_(0.0, 0.1, generator=gen)
        ring.start, ring.len = 0, ring.T
        ring.adopt_full_slots()     # random snapshots: every row differs from the one before (the dense worst case; see evolve_history)
    return Workload(name, n, e, b, k, time_gap, num_fft_batches, EdgeStream(src, dst, ts, eid), model, eng, sampler)


def prefill_distributed(dl, seed: int = 0):
    """Steady-state history for ``DistributedLstep``: identical current table on every rank, T snapshots per owner shard."""
    gen = torch.Generator(device=dl.device)
    gen.manual_seed(seed + 17)
    dl.table.normal_(0.0, 0.1, generator=gen)           # same seed on every rank -> identical replicas
    ring = dl.ring
    gen.manual_seed(seed + 1000 + dl.rank)
    for s in range(ring.T - 1):
        ring.buf[s].normal_(0.0, 0.1, generator=gen)
    ring.buf[ring.T - 1].copy_(dl.table[dl.rank::dl.W])  # newest snapshot = the current table's owned rows
    ring.start, ring.len = 0, ring.T
    ring.adopt_full_slots()


def evolve_history(runner, stream: EdgeStream, first_edge: int, batch: int, num_nodes: int, seed: int = 4321) -> int:
    """Replace the random pre-fill by what the algorithm itself produces: run the engine's own evaluation iteration (splice, update_pe,
    snapshot append; ``runner`` = LstepEngine or DistributedLstep) over the up-to-T batches of ``batch`` edges that END at
    ``first_edge``.  As in the reference, each snapshot is then a clone of the previous one plus the rows its batch wrote.
    Returns the number of batches run."""
    ring = runner.ring
    n = min(ring.T, first_edge // batch)
    dev = stream.src.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    with torch.no_grad():
        for j in range(n):
            lo = first_edge - (n - j) * batch
            src, dst, ts, eid = stream.batch(lo, lo + batch)
            neg_src = torch.randint(1, num_nodes + 1, (batch,), generator=gen, device=dev)
            neg_dst = torch.randint(1, num_nodes + 1, (batch,), generator=gen, device=dev)
            runner.eval_iteration(1000 + j, src, dst, ts, eid, neg_src, neg_dst)
    return n
