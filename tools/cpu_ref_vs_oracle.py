"""Build-container only: time the REAL reference (imported from /root/reference with the two import shims of
tests/golden/make_golden.py) against the oracle restatement on the bench's CPU-baseline sample, same threads, same inputs.
Shows that the oracle costs what the reference costs (within +-10 %)."""
import os, sys, time
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import make_golden as mg            # imports the reference (LSTEP, MergeLayer, Data, get_neighbor_sampler)
from lstep_amd import protocol, synth
from oracle.lstep_oracle import OracleNeighborSampler, build_oracle_model

N, E, B, K, G, T = 50_000, 1_000_000, 512, 20, 2000, 100
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(threads)
g = synth.make_temporal_graph(N, E, seed=0)
node_raw, edge_raw = synth.make_features(N, E, seed=1)
sd = synth.make_state_dict(K, T)

def make_runner(model):
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    hist = 0.1 * torch.randn(N + 1, T, synth.PE_DIM, generator=torch.Generator().manual_seed(0))
    st = protocol.ProtocolState(history=hist)

    def step(it):
        sl = slice(E // 2 + it * B, E // 2 + (it + 1) * B)
        neg = synth.make_negatives(N, B, seed=it)
        t0 = time.perf_counter()
        protocol.train_iteration(model[0], model[1], opt, st, 1000 + it, g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], neg, K, G, T)
        return time.perf_counter() - t0
    return step

t0 = time.perf_counter(); ref_sampler = mg.ref_sampler(g); t_ref_build = time.perf_counter() - t0
t0 = time.perf_counter(); o_sampler = OracleNeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=N); t_o_build = time.perf_counter() - t0
steps = {"reference": make_runner(mg.ref_model(node_raw, edge_raw, ref_sampler, K, T)),
         "oracle": make_runner(build_oracle_model(node_raw, edge_raw, o_sampler, K, T, sd))}
times = {k: [] for k in steps}
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for it in range(iters):                      # interleaved, so both see the same machine state
    for k in (("reference", "oracle") if it % 2 == 0 else ("oracle", "reference")):
        times[k].append(steps[k](it))
t_ref, t_orc = float(np.median(times["reference"][1:])), float(np.median(times["oracle"][1:]))
print(f"threads {threads}: reference {t_ref*1e3:.0f} ms/iter ({B/t_ref:.0f} edges/s), oracle {t_orc*1e3:.0f} ms/iter ({B/t_orc:.0f} edges/s), "
      f"oracle/reference = {t_orc/t_ref:.3f} (medians of {iters - 1} interleaved iterations);  adjacency build: reference {t_ref_build:.1f} s "
      f"(Python loop), oracle {t_o_build:.1f} s (lexsort)")
print("reference iters ms:", [round(x * 1e3) for x in times["reference"]])
print("oracle    iters ms:", [round(x * 1e3) for x in times["oracle"]])
