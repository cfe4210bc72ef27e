"""Long-run drift of the HIP engine against the CPU oracle (same seeds, same protocol) over many training batches."""
import sys, os
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np, torch
from lstep_amd import protocol, synth
from lstep_amd.engine import EdgeStream, LstepEngine
from lstep_amd.sampler import NeighborSampler
from lstep_amd.workload import build_hip_model
from oracle.lstep_oracle import OracleNeighborSampler, build_oracle_model

N, E, B, K, T, G, batches = 300, 20000, 64, 10, 8, 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda:0"
g = synth.make_temporal_graph(num_nodes=N, num_edges=E, seed=5)
node_raw, edge_raw = synth.make_features(N, E, seed=6)
pe0 = synth.make_initial_pe(N, seed=7)
sd = synth.make_state_dict(K, T, seed=8)
om = build_oracle_model(node_raw, edge_raw, OracleNeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=N), K, T, sd)
hm = build_hip_model(node_raw, edge_raw, NeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=N, device=dev), K, T, sd, dev)
oo, ho = torch.optim.Adam(om.parameters(), lr=1e-4), torch.optim.Adam(hm.parameters(), lr=1e-4)
st = protocol.ProtocolState(history=torch.zeros(N + 1, 0, 172), initial_pe=torch.from_numpy(pe0.copy()))
eng = LstepEngine(hm[0], hm[1], K, G)
stream = EdgeStream.from_numpy(g["src"], g["dst"], g["ts"], g["eid"], dev)
init = torch.from_numpy(pe0.copy()).to(dev)
for b in range(batches):
    lo = 8000 + b * B
    sl = slice(lo, lo + B)
    neg = synth.make_negatives(N, B, seed=40 + b)
    ro = protocol.train_iteration(om[0], om[1], oo, st, b, g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], neg, K, G, T)
    rh = eng.train_iteration(ho, b, *stream.batch(lo, lo + B), torch.from_numpy(neg).to(dev), initial_pe=init)
    ds = float((eng.ring.last().cpu() - st.history[:, -1, :]).abs().max())
    dp = float(np.abs(rh["predicts"].cpu().numpy() - ro["predicts"]).max()) if ro else 0.0
    dw = max(float((p.detach().cpu() - q.detach()).abs().max()) for p, q in zip(hm.parameters(), om.parameters()))
    if b % 5 == 0 or b == batches - 1:
        print(f"batch {b:3d}: max|snapshot diff| {ds:.2e}  max|prob diff| {dp:.2e}  max|weight diff| {dw:.2e}  loss {ro['loss'] if ro else float('nan'):.5f}")
