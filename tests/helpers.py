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


KINK_NEAR = 1e-5      # |pre-activation| below which the reference's own relu derivative is decided by float32 rounding (HIP's pre-activations differ by ~1e-6)


def check_long_trace_gradients(model, z, b, atol, digest_atol, loose=5e-5):
    """Parameter gradients of step b against tests/golden/traces_long.npz (big matrices: every 32nd row + a digest of all of it).

    A step whose reference pre-activations come within ``KINK_NEAR`` of a relu kink (``b{b}/kink_distance``, recorded by hooks on the
    reference's three relu inputs) has an ill-defined reference gradient: whichever side of 0 float32 rounding lands on decides whether
    that row contributes to every layer in front of the relu (measured: step 4 of this trace has |h| = 6e-8 and one unit's row of
    d edge_mlp_1.weight moves by 1.9e-5, with identical weights on both sides -- tools/long_trace_diag.py).  Those steps are held to:
    no entry further than ``loose`` (north_star's 1e-4 halved) and at most 3 % of all compared entries beyond ``atol`` (one row of a
    272-wide layer is 1.5 % of the sampled entries); every other step
    to ``atol`` on every entry and ``digest_atol`` on the sum of each whole tensor.  Returns (largest difference, near a kink?)."""
    near = float(z[f"b{b}/kink_distance"][0]) < KINK_NEAR
    worst, total, beyond = 0.0, 0, 0
    for k, p in model.named_parameters():
        if f"b{b}/grads/{k}" not in z.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        a = p.grad.detach().cpu().numpy()
        a = np.stack([a.real, a.imag], -1) if np.iscomplexobj(a) else a
        got = a[::LONG_GRAD_STRIDE] if a.size > 20000 else a
        d = np.abs(got - z[f"b{b}/grads/{k}"])
        worst, total, beyond = max(worst, float(d.max())), total + d.size, beyond + int((d > atol).sum())
        if near:
            assert d.max() <= loose, f"b{b} {k}: {d.max():.3e} (reference within {float(z[f'b{b}/kink_distance'][0]):.1e} of a relu kink)"
        else:
            np.testing.assert_allclose(got, z[f"b{b}/grads/{k}"], rtol=0, atol=atol, err_msg=f"b{b} {k}")
            np.testing.assert_allclose(a.astype(np.float64).sum(), z[f"b{b}/grads/{k}/digest"][0], rtol=0, atol=digest_atol, err_msg=f"b{b} {k}")
    assert beyond <= 0.03 * total, f"b{b}: {beyond} of {total} gradient entries differ by more than {atol}: more than one relu row can explain"
    return worst, near


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


# ------------------------------------------------------------------------------------------------ one oracle training iteration, in row chunks
def oracle_train_step_chunked(om, window_rows, last_table, bn, batch_idx, src, dst, ts, eid, neg, K, G, chunk=256,
                              pe_weight=0.5, neg_sample_weight=0.3, want_emb=None, edges=None):
    """One batch of train_LSTEP_link_prediction.py:204-311 (batch_idx > 0) run by the CPU oracle FROM A GIVEN STATE, the 3 B embedding
    rows processed ``chunk`` edges at a time: the oracle's node channel gathers a dense [rows, time_gap, 172] block (models/LSTEP.py:181),
    5.6 GB per call at the Reddit shape and 22.5 GB at B = 16384, so the full-batch losses and gradients of the big configurations are
    only affordable in slices.  Both losses are means over the batch (BCE over the 2 B probabilities, MSE over B x P entries), so every
    slice back-propagates its share and the shares add up to the full-batch gradient (summation order aside; checked against
    ``protocol.train_iteration`` on the CPU in tests/test_host_cpu.py).

    ``window_rows`` [U, t, P]: the history window of the batch nodes ``bn`` (sorted unique endpoints) -- all the FFT splice reads;
    ``last_table`` [N+1, P]: the newest snapshot.  Leaves the parameter gradients in ``om``'s ``.grad`` (no optimiser step).  Returns
    losses, the 2 B link probabilities, the table update_pe produced from the spliced table, the spliced rows and, for the row indices
    ``want_emb`` into cat[src, dst, neg], their embeddings.  ``edges`` (positions in the batch): only those edges go through the embedding
    stage -- their probabilities and embeddings are exact, the losses and gradients are then partial sums (``"partial": True``) and
    only the state transition (spliced table, update_pe) covers the whole batch."""
    bb, pred = om[0], om[1]
    dt = last_table.dtype
    B, U = len(src), len(bn)
    om.zero_grad()
    filtered = bb.fourier_transform_pe(np.arange(U), window_rows, batch_idx)          # [U, P]; the graph to fft_filter / fft_agg
    leaf = filtered.detach().clone().requires_grad_(True)
    cur = last_table.clone().index_put((torch.from_numpy(bn),), leaf)                 # train:229-230 (only the spliced rows carry gradient)
    lp_sum, pe_pos_sum, pe_neg_sum = 0.0, 0.0, 0.0
    probs = np.empty(2 * B, dtype=np.float64)
    emb_keep = {}
    want = set(int(i) for i in want_emb) if want_emb is not None else set()
    P = cur.shape[1]
    todo = np.arange(B) if edges is None else np.asarray(edges, dtype=np.int64)
    probs[:] = np.nan
    for c0 in range(0, len(todo), chunk):
        sel = todo[c0:c0 + chunk]
        s, d, n_, t = src[sel], dst[sel], neg[sel], ts[sel]
        e_s = bb.combining_pe_raw_feat(cur, s, t, K, G)
        e_d = bb.combining_pe_raw_feat(cur, d, t, K, G)
        e_n = bb.combining_pe_raw_feat(cur, n_, t, K, G)
        p_pos = pred(input_1=e_s, input_2=e_d).squeeze(dim=-1).sigmoid().clamp(0, 1)
        p_neg = pred(input_1=e_s, input_2=e_n).squeeze(dim=-1).sigmoid().clamp(0, 1)      # neg_src = pos_src (train:245)
        i_s, i_d, i_n = (torch.from_numpy(np.ascontiguousarray(a)) for a in (s, d, n_))
        lp = torch.nn.functional.binary_cross_entropy(torch.cat([p_pos, p_neg]), torch.cat([torch.ones_like(p_pos), torch.zeros_like(p_neg)]),
                                                      reduction="sum") / (2 * B)
        pos = ((cur[i_s] - cur[i_d]) ** 2).sum() / (B * P)
        ngt = ((cur[i_s] - cur[i_n]) ** 2).sum() / (B * P)
        loss = (1.0 - pe_weight) * lp + pe_weight * (pos - neg_sample_weight * ngt)
        loss.backward(retain_graph=True)
        lp_sum, pe_pos_sum, pe_neg_sum = lp_sum + float(lp), pe_pos_sum + float(pos), pe_neg_sum + float(ngt)
        probs[sel], probs[B + sel] = p_pos.detach().numpy(), p_neg.detach().numpy()
        for blk, e in enumerate((e_s, e_d, e_n)):
            for j, r in enumerate(sel):
                if blk * B + int(r) in want:
                    emb_keep[blk * B + int(r)] = e[j].detach().numpy().copy()
    filtered.backward(leaf.grad if leaf.grad is not None else torch.zeros_like(leaf))
    pe_loss = pe_pos_sum - neg_sample_weight * pe_neg_sum
    with torch.no_grad():
        spliced = cur.detach().clone()
        table = bb.update_pe(pe=spliced.clone(), node_ids=bn, edge_ids=eid, batch_src_node_ids=src, batch_dst_node_ids=dst,
                             node_interact_times=ts, current_time=ts.max(), num_neighbors=K, time_gap=G)
    return {"lp_loss": lp_sum, "pe_loss": pe_loss, "loss": (1.0 - pe_weight) * lp_sum + pe_weight * pe_loss, "predicts": probs,
            "table": table, "spliced": spliced, "emb": emb_keep, "partial": edges is not None}
