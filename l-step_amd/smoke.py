"""One tiny end-to-end invocation of the HIP path, checked against the CPU oracle (used by ``__graft_entry__.smoke``)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import protocol, synth
from .engine import EdgeStream, LstepEngine
from .model import LSTEP, MergeLayer
from .sampler import NeighborSampler


def build_hip_model(node_raw, edge_raw, sampler, K, T, state_dict=None, device="cuda"):
    bb = LSTEP(node_raw, edge_raw, sampler, sampler, pe_dim=synth.PE_DIM, num_neighbors=K, time_feat_dim=synth.TIME_DIM,
               num_fft_batches=T, device=device)
    pred = MergeLayer(synth.FEAT_DIM, synth.FEAT_DIM, synth.FEAT_DIM, 1).to(device)
    model = torch.nn.Sequential(bb, pred)
    if state_dict is not None:
        model.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)
    return model


def run_smoke(device="cuda:0", batches=3, B=32, K=5, T=4, G=2000) -> float:
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle.lstep_oracle import OracleNeighborSampler, build_oracle_model  # checker only

    g = synth.make_temporal_graph(num_nodes=96, num_edges=3000, seed=5)
    node_raw, edge_raw = synth.make_features(96, 3000, seed=6)
    pe0 = synth.make_initial_pe(96, seed=7)
    sd = synth.make_state_dict(K, T, seed=8)
    o_model = build_oracle_model(node_raw, edge_raw, OracleNeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=96), K, T, sd)
    h_sampler = NeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=96, device=device)
    h_model = build_hip_model(node_raw, edge_raw, h_sampler, K, T, sd, device)
    o_opt = torch.optim.Adam(o_model.parameters(), lr=1e-4)
    h_opt = torch.optim.Adam(h_model.parameters(), lr=1e-4)
    o_state = protocol.ProtocolState(history=torch.zeros(97, 0, synth.PE_DIM), initial_pe=torch.from_numpy(pe0.copy()))
    eng = LstepEngine(h_model[0], h_model[1], K, G)
    stream = EdgeStream.from_numpy(g["src"], g["dst"], g["ts"], g["eid"], device)
    init = torch.from_numpy(pe0.copy()).to(device)
    worst = 0.0
    for b in range(batches):
        lo = 1500 + b * B
        sl = slice(lo, lo + B)
        neg = synth.make_negatives(96, B, seed=40 + b)
        ro = protocol.train_iteration(o_model[0], o_model[1], o_opt, o_state, b, g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], neg, K, G, T)
        rh = eng.train_iteration(h_opt, b, *stream.batch(lo, lo + B), torch.from_numpy(neg).to(device), initial_pe=init)
        snap = eng.ring.last().cpu()
        worst = max(worst, float((snap - o_state.history[:, -1, :]).abs().max()))
        if ro is not None:
            worst = max(worst, float(np.abs(rh["predicts"].cpu().numpy() - ro["predicts"]).max()))
    return worst
