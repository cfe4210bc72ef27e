"""Shared builders for the parity tests: the same seeded inputs tests/golden/make_golden.py fed to the reference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lstep_amd import synth  # noqa: E402

# mirrors of the constants in tests/golden/make_golden.py (kept literal so the GPU box needs no reference)
SAMPLER_GRAPHS = {
    "uniform": dict(num_nodes=48, num_edges=1500, seed=10),
    "ties": dict(num_nodes=24, num_edges=1200, seed=11, time_span=12000.0, tie_quantum=50.0),
    "epoch": dict(num_nodes=48, num_edges=1500, seed=12, time_span=1e5, epoch_offset=1.6e9),
}
METHOD_GRAPH = dict(num_nodes=64, num_edges=2000, seed=20)
METHOD_K, METHOD_T = 5, 6
TRACE_GRAPH = dict(num_nodes=64, num_edges=2000, seed=30)
TRACE_K, TRACE_T, TRACE_B, TRACE_G = 5, 4, 16, 2000
TRACE_START, TRACE_BATCHES, GRAD_ROW_STRIDE = 640, 7, 4
EVAL_LOOP = dict(first=1200, edges=6 * 16 + 5, batch=16, stored=2)
LONG_BATCHES, LONG_GRAD_STRIDE = 16, 32          # tests/golden/traces_long.npz
F64_GRAPH = dict(num_nodes=64, num_edges=2000, seed=40, time_span=4096.0, tie_quantum=0.125)   # tests/golden/float64.npz
F64_K, F64_T = 5, 6
WS_GRAPH = dict(num_nodes=40, num_edges=1200, seed=90, time_span=60.0, tie_quantum=0.25)
WS_K, WS_T = 5, 4


def method_inputs():
    g = synth.make_temporal_graph(**METHOD_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=21)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=22)
    return g, node_raw, edge_raw, pe0


def trace_inputs():
    g = synth.make_temporal_graph(**TRACE_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=31)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=32)
    return g, node_raw, edge_raw, pe0


def trace_batches(g):
    out = []
    for b in range(TRACE_BATCHES):
        sl = slice(TRACE_START + b * TRACE_B, TRACE_START + (b + 1) * TRACE_B)
        out.append((g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], synth.make_negatives(g["num_nodes"], TRACE_B, seed=500 + b)))
    return out


def long_trace_batches(g):
    """The 16 consecutive batches of tests/golden/traces_long.npz (the first 7 are ``trace_batches``)."""
    out = []
    for b in range(LONG_BATCHES):
        sl = slice(TRACE_START + b * TRACE_B, TRACE_START + (b + 1) * TRACE_B)
        out.append((g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], synth.make_negatives(g["num_nodes"], TRACE_B, seed=500 + b)))
    return out


def check_long_trace_gradients(model, z, b, atol, digest_atol):
    """Parameter gradients of step b against tests/golden/traces_long.npz (big matrices: every 32nd row + a digest of all of it)."""
    for k, p in model.named_parameters():
        if f"b{b}/grads/{k}" not in z.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        a = p.grad.detach().cpu().numpy()
        a = np.stack([a.real, a.imag], -1) if np.iscomplexobj(a) else a
        got = a[::LONG_GRAD_STRIDE] if a.size > 20000 else a
        np.testing.assert_allclose(got, z[f"b{b}/grads/{k}"], rtol=0, atol=atol, err_msg=f"b{b} {k}")
        np.testing.assert_allclose(a.astype(np.float64).sum(), z[f"b{b}/grads/{k}/digest"][0], rtol=0, atol=digest_atol, err_msg=f"b{b} {k}")


def float64_inputs():
    g = synth.make_temporal_graph(**F64_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=41)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=42).astype(np.float64)
    return g, node_raw, edge_raw, pe0


def eval_batches(g):
    out = []
    for b in range(3):
        s0 = TRACE_START + (TRACE_BATCHES + b) * TRACE_B
        sl = slice(s0, s0 + TRACE_B)
        out.append((g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl],
                    synth.make_negatives(g["num_nodes"], TRACE_B, seed=800 + b), synth.make_negatives(g["num_nodes"], TRACE_B, seed=700 + b)))
    return out


def param_digest(model):
    d = {}
    for k, v in model.state_dict().items():
        a = v.detach().cpu().numpy()
        if np.iscomplexobj(a):
            a = np.stack([a.real, a.imag], -1)
        a = a.astype(np.float64).reshape(-1)
        d[k] = np.asarray([a.sum(), np.abs(a).sum(), (a * np.arange(1, a.size + 1) / a.size).sum()])
    return d


def state_dict_tensors(K, T, seed=3, device="cpu"):
    return {k: torch.from_numpy(v).to(device) for k, v in synth.make_state_dict(K, T, seed=seed).items()}


def eval_loop_batches(g, z, strategy):
    """The batches the reference's own evaluate_model_link_prediction iterated over (tests/golden/eval_loop.npz): its index DataLoader's
    chronological slices incl. the ragged tail, and the negatives its NegativeEdgeSampler drew for each."""
    lo, n, bsz = EVAL_LOOP["first"], EVAL_LOOP["edges"], EVAL_LOOP["batch"]
    out = []
    for b, s0 in enumerate(range(lo, lo + n, bsz)):
        sl = slice(s0, min(s0 + bsz, lo + n))
        out.append((g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], z[f"{strategy}/b{b}/neg_src"], z[f"{strategy}/b{b}/neg_dst"]))
    return out


def eval_loop_expected(z, strategy, b):
    """(probabilities [pos | neg], snapshot, loss) the reference's loop produced for batch b."""
    logits = np.concatenate([z[f"{strategy}/b{b}/pos_logits"], z[f"{strategy}/b{b}/neg_logits"]])
    prob = np.clip(1.0 / (1.0 + np.exp(-logits.astype(np.float64))), 0.0, 1.0)
    return prob, z[f"{strategy}/b{b}/snapshot"], float(z[f"{strategy}/losses"][b])


def variant_inputs():
    """Inputs of tests/golden/variants.npz (weighted_sum ablation, RNG-defined sampling strategies through the model)."""
    g = synth.make_temporal_graph(**WS_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=91)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=92)
    pe0[0] = 0.03
    sl = slice(900, 924)
    return g, node_raw, edge_raw, pe0, (g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl])
