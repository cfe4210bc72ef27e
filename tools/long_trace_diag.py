"""Diagnostic: per-step gradient differences of the engine against tests/golden/traces_long.npz (GPU)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import *  # noqa
from lstep_amd import synth
from lstep_amd.engine import EdgeStream, LstepEngine
from lstep_amd.sampler import NeighborSampler
from lstep_amd.workload import build_hip_model
from lstep_amd.optim import FusedAdam
DEV = "cuda:0"
z = np.load(os.path.join(ROOT, "tests/golden/traces_long.npz"))
sync = os.environ.get("SYNC_FROM_ORACLE") == "1"
g, node_raw, edge_raw, pe0 = trace_inputs()
s = NeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=g["num_nodes"], device=DEV)
model = build_hip_model(node_raw, edge_raw, s, TRACE_K, TRACE_T, synth.make_state_dict(TRACE_K, TRACE_T), DEV)
model.train()
eng = LstepEngine(model[0], model[1], TRACE_K, TRACE_G)
eng.use_step_graph = os.environ.get("GRAPH") == "1"
opt = FusedAdam(model.parameters(), lr=1e-4)
stream = EdgeStream.from_numpy(g["src"], g["dst"], g["ts"], g["eid"], DEV)
init = torch.from_numpy(pe0.copy()).to(DEV)
for b in range(LONG_BATCHES):
    lo = TRACE_START + b * TRACE_B
    neg = torch.from_numpy(synth.make_negatives(g["num_nodes"], TRACE_B, seed=500 + b)).to(DEV)
    res = eng.train_iteration(opt, b, *stream.batch(lo, lo + TRACE_B), neg, initial_pe=init)
    snap = float(np.abs(eng.ring.last().cpu().numpy() - z[f"b{b}/snapshot"]).max())
    if res is None:
        print(f"b{b}: snapshot {snap:.2e}"); continue
    worst = []
    for k, p in model.named_parameters():
        if f"b{b}/grads/{k}" not in z.files:
            continue
        a = p.grad.detach().cpu().numpy()
        a = np.stack([a.real, a.imag], -1) if np.iscomplexobj(a) else a
        got = a[::LONG_GRAD_STRIDE] if a.size > 20000 else a
        ref = z[f"b{b}/grads/{k}"]
        worst.append((float(np.abs(got - ref).max()), float(np.abs(ref).max()), float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)), k))
    worst.sort(reverse=True)
    pr = float(np.abs(res["predicts"].cpu().numpy() - z[f"b{b}/predicts"]).max())
    print(f"b{b}: snapshot {snap:.2e} predicts {pr:.2e} | worst grads: " + "; ".join(f"{k.split('.',1)[1]} d={d:.1e} max={m:.1e} rel2={r:.1e}" for d, m, r, k in worst[:3]))

# ---- second pass: the oracle in lockstep (weights copied from the HIP model before every step), full gradient matrices
if os.environ.get("LOCKSTEP", "1") == "1":
    from oracle.lstep_oracle import OracleNeighborSampler, build_oracle_model
    from lstep_amd import protocol
    print("---- lockstep with the oracle (same weights every step)")
    model = build_hip_model(node_raw, edge_raw, s, TRACE_K, TRACE_T, synth.make_state_dict(TRACE_K, TRACE_T), DEV)
    model.train()
    eng = LstepEngine(model[0], model[1], TRACE_K, TRACE_G)
    opt = FusedAdam(model.parameters(), lr=1e-4)
    om = build_oracle_model(node_raw, edge_raw, OracleNeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=g["num_nodes"]), TRACE_K, TRACE_T,
                            synth.make_state_dict(TRACE_K, TRACE_T))
    om.train()
    oopt = torch.optim.SGD(om.parameters(), lr=0.0)
    st = protocol.ProtocolState(history=torch.zeros(g["num_nodes"] + 1, 0, synth.PE_DIM), initial_pe=torch.from_numpy(pe0.copy()))
    init = torch.from_numpy(pe0.copy()).to(DEV)
    hooks = {}
    om[0].edge_agg.register_forward_hook(lambda m, i, o: hooks.setdefault("h", []).append(o.detach().squeeze().clone()))
    for b in range(LONG_BATCHES):
        om.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
        lo = TRACE_START + b * TRACE_B
        sl = slice(lo, lo + TRACE_B)
        negn = synth.make_negatives(g["num_nodes"], TRACE_B, seed=500 + b)
        hooks.clear()
        # same state: the oracle continues from the ENGINE's history
        if b > 0:
            st.history = eng.ring.as_reference_tensor().cpu()
        ro = protocol.train_iteration(om[0], om[1], oopt, st, b, g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], negn, TRACE_K, TRACE_G, TRACE_T)
        res = eng.train_iteration(opt, b, *stream.batch(lo, lo + TRACE_B), torch.from_numpy(negn).to(DEV), initial_pe=init)
        if res is None:
            continue
        h = torch.cat(hooks["h"]).abs()
        ga, gb = model[0].edge_mlp_1.weight.grad.cpu(), om[0].edge_mlp_1.weight.grad
        rows = (ga - gb).abs().max(dim=1).values
        bad = (rows > 1e-6).nonzero().reshape(-1).tolist()
        print(f"b{b}: min |pre-relu h| = {float(h.min()):.2e} (second {float(h.flatten().kthvalue(2).values):.2e}); edge_mlp_1.weight rows differing > 1e-6: {bad[:10]} (max {float(rows.max()):.2e})")
