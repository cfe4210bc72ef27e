#!/usr/bin/env python3
"""Generate the golden vectors under ``tests/golden/`` by RUNNING THE REFERENCE (``/root/reference``) on CPU.

Run in the build container only (``python tests/golden/make_golden.py``); the GPU box has no reference and
only reads the committed ``.npz`` files.  The reference's own source is never copied: it is imported from
where it lies, after two import shims for wheels this image lacks:

* ``torch_scatter.scatter(src, index, dim=0, out=, reduce='sum')`` -> ``out.scatter_add_`` (the only call shape
  on the path: reference ``models/LSTEP.py:283-290,320-322``; plain scatter-add is its documented behaviour);
* an empty ``tgb.linkproppred.dataset`` (only the TGB file loader uses it, ``utils/DataLoader.py:96``).

Inputs are NOT stored: they are regenerated from seeds by ``lstep_amd.synth`` (numpy legacy RNG, bit-stable).
Only call arguments that are cheap (ids, times) and the reference's outputs are written.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
REFERENCE = os.environ.get("LSTEP_REFERENCE", "/root/reference")

from lstep_amd import synth  # noqa: E402
from lstep_amd import protocol  # noqa: E402


def import_reference():
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, out=None, reduce="sum"):
        assert dim == 0 and out is not None and reduce == "sum"
        return out.scatter_add_(0, index.view(-1, 1).expand_as(src), src)

    def scatter_mean(src, index, out=None, dim=-1):
        # documented semantics of torch_scatter.scatter_mean for the one call shape on the path (models/LSTEP.py:194: 1-D src / index,
        # zero-initialised `out`): out[i] = sum of the entries with index i / max(count, 1)
        assert out is not None and src.dim() == 1 and index.dim() == 1 and dim in (-1, 0)
        out.scatter_add_(0, index, src)
        cnt = torch.zeros_like(out).scatter_add_(0, index, torch.ones_like(src)).clamp_(min=1)
        return out.div_(cnt)

    ts.scatter, ts.scatter_mean = scatter, scatter_mean
    sys.modules["torch_scatter"] = ts
    tgb = types.ModuleType("tgb")
    lp = types.ModuleType("tgb.linkproppred")
    ds = types.ModuleType("tgb.linkproppred.dataset")
    ds.LinkPropPredDataset = type("LinkPropPredDataset", (), {})
    sys.modules.update({"tgb": tgb, "tgb.linkproppred": lp, "tgb.linkproppred.dataset": ds})
    sys.path.insert(0, REFERENCE)
    from models.LSTEP import LSTEP
    from models.modules import MergeLayer, TimeEncoder
    from utils.DataLoader import Data
    from utils.utils import get_neighbor_sampler

    return LSTEP, MergeLayer, TimeEncoder, Data, get_neighbor_sampler


LSTEP, MergeLayer, TimeEncoder, Data, get_neighbor_sampler = import_reference()


def ref_sampler(g, upto=None):
    sl = slice(0, upto)
    data = Data(g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], np.zeros(len(g["src"][sl])))
    return get_neighbor_sampler(data, sample_neighbor_strategy="recent", seed=None)


def ref_model(node_raw, edge_raw, sampler, K, T, seed=3, weighted_sum=False):
    torch.manual_seed(0)
    bb = LSTEP(node_raw_features=node_raw, edge_raw_features=edge_raw, neighbor_sampler=sampler,
               full_neighbor_sampler=sampler, pe_dim=synth.PE_DIM, num_neighbors=K, time_feat_dim=synth.TIME_DIM,
               num_fft_batches=T, weighted_sum=weighted_sum, device="cpu")
    pred = MergeLayer(input_dim1=synth.FEAT_DIM, input_dim2=synth.FEAT_DIM, hidden_dim=synth.FEAT_DIM, output_dim=1)
    model = torch.nn.Sequential(bb, pred)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(K, T, seed=seed).items()}
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model


# ----------------------------------------------------------------------------------------------- G1 sampler
SAMPLER_GRAPHS = {
    "uniform": dict(num_nodes=48, num_edges=1500, seed=10),
    "ties": dict(num_nodes=24, num_edges=1200, seed=11, time_span=12000.0, tie_quantum=50.0),
    "epoch": dict(num_nodes=48, num_edges=1500, seed=12, time_span=1e5, epoch_offset=1.6e9),
}


def sampler_queries(g, seed):
    """Query (node, time) pairs that hit every edge case of `find_neighbors_before` (utils/utils.py:129-146)."""
    rng = np.random.RandomState(seed)
    n = g["num_nodes"]
    ids, ts = [], []
    pick = rng.randint(0, len(g["ts"]), size=12)
    for e in pick:  # time exactly equal to one of the node's own timestamps: strictly-earlier rule
        ids.append(g["src"][e]); ts.append(g["ts"][e])
        ids.append(g["dst"][e]); ts.append(g["ts"][e])
    for _ in range(10):  # random node at random time
        ids.append(rng.randint(1, n + 1)); ts.append(rng.uniform(g["ts"][0], g["ts"][-1]))
    ids += [0, 0, 1, 2, 3, n, n]
    ts += [g["ts"][-1] + 1.0, g["ts"][0], g["ts"][0], g["ts"][0] - 5.0, g["ts"][-1] + 1e6, g["ts"][-1] + 1.0, np.nextafter(g["ts"][5], np.inf)]
    return np.asarray(ids, dtype=np.int64), np.asarray(ts, dtype=np.float64)


def gen_sampler():
    out = {}
    for name, kw in SAMPLER_GRAPHS.items():
        g = synth.make_temporal_graph(**kw)
        s = ref_sampler(g)
        ids, ts = sampler_queries(g, seed=100 + kw["seed"])
        out[f"{name}/ids"], out[f"{name}/ts"] = ids, ts
        for k in (1, 5, 20, 32, 8, 2000):
            nbr, eid, nt = s.get_historical_neighbors(ids, ts, k)
            assert nbr.dtype == np.int64 and nt.dtype == np.float32
            if k == 2000:  # right-aligned rows: store the last 192 columns plus the non-zero count of the rest (0 here)
                keep = 192
                out[f"{name}/k{k}/head_nnz"] = np.asarray([(nbr[:, : k - keep] != 0).sum()], dtype=np.int64)
                out[f"{name}/k{k}/nbr_tail"], out[f"{name}/k{k}/eid_tail"], out[f"{name}/k{k}/nt_tail"] = nbr[:, -keep:], eid[:, -keep:], nt[:, -keep:]
                out[f"{name}/k{k}/nbr_sum"] = nbr.sum(1)
                out[f"{name}/k{k}/eid_sum"] = eid.sum(1)
            else:
                out[f"{name}/k{k}/nbr"], out[f"{name}/k{k}/eid"], out[f"{name}/k{k}/nt"] = nbr, eid, nt
        # length mismatch: zip() truncates to the shorter of (ids, times) -- utils/utils.py:160-169
        for tag, (a, b) in {"more_ids": (ids, ts[:9]), "more_ts": (ids[:9], ts)}.items():
            nbr, eid, nt = s.get_historical_neighbors(a, b, 5)
            out[f"{name}/{tag}/nbr"], out[f"{name}/{tag}/eid"], out[f"{name}/{tag}/nt"] = nbr, eid, nt
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)
    print("sampler.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- G2 time encoder
def gen_time_encoder():
    dt = np.asarray([0.0, 1e-3, 0.5, 1.0, 3.14159, 17.0, 1e2, 1234.5, 1e4, 86400.0, 1e6, 3.3e7, 1e9, 1.6e9, -1.0, -250.0],
                    dtype=np.float32)
    rng = np.random.RandomState(7)
    dt = np.concatenate([dt, rng.uniform(0, 2e4, size=48).astype(np.float32), (10 ** rng.uniform(-3, 9, size=64)).astype(np.float32)])
    enc = TimeEncoder(synth.TIME_DIM, parameter_requires_grad=False)
    with torch.no_grad():
        y = enc(torch.from_numpy(dt).unsqueeze(0)).squeeze(0).numpy()
    np.savez_compressed(os.path.join(HERE, "time_encoder.npz"), dt=dt, enc=y,
                        w=enc.w.weight.detach().numpy().reshape(-1))
    print("time_encoder.npz", y.shape)


# ----------------------------------------------------------------------------------------------- G3 methods
METHOD_GRAPH = dict(num_nodes=64, num_edges=2000, seed=20)
METHOD_K, METHOD_T = 5, 6


def live_pe(model, g, pe0, K):
    """A PE table whose padding row 0 is non-zero: one update_pe on an early batch (LSTEP.py:317,339)."""
    pe = torch.from_numpy(pe0.copy())
    sl = slice(40, 56)
    bn = protocol.unique_batch_nodes(g["src"][sl], g["dst"][sl])
    with torch.no_grad():
        model[0].update_pe(pe, bn, g["eid"][sl], g["src"][sl], g["dst"][sl], g["ts"][sl], g["ts"][sl].max(), num_neighbors=K)
    return pe


def gen_methods():
    g = synth.make_temporal_graph(**METHOD_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=21)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=22)
    sampler = ref_sampler(g)
    out = {}
    for K in (METHOD_K, 20):
        model = ref_model(node_raw, edge_raw, sampler, K, METHOD_T)
        bb = model[0]
        pe = live_pe(model, g, pe0, K)
        out[f"K{K}/pe_live"] = pe.numpy().copy()
        assert np.abs(pe[0].numpy()).max() > 0, "padding row must be live for this fixture"
        for tag, sl in {"mid": slice(1200, 1216), "early": slice(3, 19)}.items():
            src, dst, t = g["src"][sl], g["dst"][sl], g["ts"][sl]
            with torch.no_grad():
                for G in (8, 2000):
                    out[f"K{K}/{tag}/agg_G{G}"] = bb.aggregated_node_embeddings(src, t, K, G).numpy()
                out[f"K{K}/{tag}/cpe"] = bb.compute_neighborhood_pe(pe, dst, t, K).numpy()
                out[f"K{K}/{tag}/out_src"] = bb.combining_pe_raw_feat(pe, src, t, K, 2000).numpy()
                out[f"K{K}/{tag}/out_dst"] = bb.combining_pe_raw_feat(pe, dst, t, K, 8).numpy()
        # update_pe: U > B (mid batch, 16 edges over 64 nodes) -- zip-truncation leaves rows >= B as padding
        sl = slice(1200, 1216)
        src, dst, t, eid = g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl]
        bn = protocol.unique_batch_nodes(src, dst)
        assert len(bn) > len(src)
        pe_in = pe.clone()
        with torch.no_grad():
            res = bb.update_pe(pe_in, bn, eid, src, dst, t, t.max(), num_neighbors=K)
        assert res is pe_in
        out[f"K{K}/update_UgtB/pe_out"] = pe_in.numpy().copy()
        # update_pe: U < B (48 edges among the 12 lowest-id endpoints of a late window)
        idx = np.nonzero((g["src"] <= 12) & (g["dst"] <= 12))[0]
        idx = idx[idx > 600][:24]
        src, dst, t, eid = g["src"][idx], g["dst"][idx], g["ts"][idx], g["eid"][idx]
        bn = protocol.unique_batch_nodes(src, dst)
        assert len(bn) < len(src), (len(bn), len(src))
        out[f"K{K}/update_UltB/edge_pos"] = idx.astype(np.int64)
        pe_in = pe.clone()
        with torch.no_grad():
            bb.update_pe(pe_in, bn, eid, src, dst, t, t.max(), num_neighbors=K)
        out[f"K{K}/update_UltB/pe_out"] = pe_in.numpy().copy()
        # update_pe from a zero-padding-row table at the very start of the stream (all-padding neighbourhoods)
        sl = slice(0, 16)
        src, dst, t, eid = g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl]
        bn = protocol.unique_batch_nodes(src, dst)
        pe_in = torch.from_numpy(pe0.copy())
        with torch.no_grad():
            bb.update_pe(pe_in, bn, eid, src, dst, t, t.max(), num_neighbors=K)
        out[f"K{K}/update_first/pe_out"] = pe_in.numpy().copy()

    # fourier_transform_pe: masked (stored < T), unmasked (stored == T), eval-style batch_idx != stored length
    model = ref_model(node_raw, edge_raw, sampler, METHOD_K, METHOD_T)
    bb = model[0]
    rng = np.random.RandomState(23)
    hist_full = (0.1 * rng.standard_normal((g["num_nodes"] + 1, METHOD_T, synth.PE_DIM))).astype(np.float32)
    ids = np.asarray([1, 2, 5, 9, 17, 33, 64, 0], dtype=np.int64)
    out["fft/ids"] = ids
    with torch.no_grad():
        for stored, bidx in ((3, 3), (METHOD_T, 9), (4, 2), (2, 0), (1, 1), (METHOD_T, 0)):
            h = torch.from_numpy(hist_full[:, :stored, :].copy())
            out[f"fft/stored{stored}_b{bidx}"] = bb.fourier_transform_pe(ids, h, bidx).numpy()
    np.savez_compressed(os.path.join(HERE, "methods.npz"), **out)
    print("methods.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- G4-G6 traces
TRACE_GRAPH = dict(num_nodes=64, num_edges=2000, seed=30)
TRACE_K, TRACE_T, TRACE_B, TRACE_G = 5, 4, 16, 2000
TRACE_START = 640  # start mid-stream so neighbourhoods are non-trivial
GRAD_ROW_STRIDE = 4
TRACE_BATCHES = 7  # > T + 1 so the history trim (train:224-225) is exercised


def trace_batches(g):
    bs = []
    for b in range(TRACE_BATCHES):
        sl = slice(TRACE_START + b * TRACE_B, TRACE_START + (b + 1) * TRACE_B)
        neg = synth.make_negatives(g["num_nodes"], TRACE_B, seed=500 + b)
        bs.append((g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], neg))
    return bs


def param_digest(model):
    d = {}
    for k, v in model.state_dict().items():
        a = v.detach().numpy()
        if np.iscomplexobj(a):
            a = np.stack([a.real, a.imag], -1)
        a = a.astype(np.float64).reshape(-1)
        d[k] = np.asarray([a.sum(), np.abs(a).sum(), (a * np.arange(1, a.size + 1) / a.size).sum()])
    return d


def gen_traces():
    g = synth.make_temporal_graph(**TRACE_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=31)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=32)
    sampler = ref_sampler(g)
    out = {}

    # ---- G5: training trace (also yields G4: gradients of the first optimised batch)
    model = ref_model(node_raw, edge_raw, sampler, TRACE_K, TRACE_T)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    state = protocol.ProtocolState(history=torch.zeros(g["num_nodes"] + 1, 0, synth.PE_DIM), initial_pe=torch.from_numpy(pe0.copy()))
    for b, (src, dst, t, eid, neg) in enumerate(trace_batches(g)):
        res = protocol.train_iteration(model[0], model[1], opt, state, b, src, dst, t, eid, neg,
                                       TRACE_K, TRACE_G, TRACE_T)
        out[f"train/b{b}/snapshot"] = state.history[:, -1, :].numpy().copy()
        if res is not None:
            out[f"train/b{b}/losses"] = np.asarray([res["lp_loss"], res["pe_loss"], res["loss"]])
            out[f"train/b{b}/predicts"] = res["predicts"]
        if b == 1:
            for k, p in model.named_parameters():
                if p.grad is None:
                    out[f"grads/{k}/none"] = np.zeros(0)
                else:
                    a = p.grad.detach().numpy()
                    a = np.stack([a.real, a.imag], -1) if np.iscomplexobj(a) else a.copy()
                    out[f"grads/{k}/digest"] = np.asarray([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
                    # big matrices: every 4th row is stored (fixture size); the digest covers the rest
                    out[f"grads/{k}"] = a[::GRAD_ROW_STRIDE] if a.size > 20000 else a
        for k, v in param_digest(model).items():
            out[f"train/b{b}/digest/{k}"] = v
    out["train/final_history"] = state.history.numpy().copy()

    # ---- G6: evaluation trace continuing from the trained history (evaluate_model_utils.py:37)
    model.eval()
    ev = protocol.ProtocolState(history=state.history.clone())
    with torch.no_grad():
        for b in range(3):
            s0 = TRACE_START + (TRACE_BATCHES + b) * TRACE_B
            sl = slice(s0, s0 + TRACE_B)
            neg_dst = synth.make_negatives(g["num_nodes"], TRACE_B, seed=700 + b)
            neg_src = synth.make_negatives(g["num_nodes"], TRACE_B, seed=800 + b)
            res = protocol.eval_iteration(model[0], model[1], ev, b, g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl],
                                          neg_src, neg_dst, TRACE_K, TRACE_G, TRACE_T)
            out[f"eval/b{b}/loss"] = np.asarray([res["loss"]])
            out[f"eval/b{b}/predicts"] = res["predicts"]
            out[f"eval/b{b}/snapshot"] = ev.history[:, -1, :].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "traces.npz"), **out)
    print("traces.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- long training trace (graph-replay parity)
LONG_BATCHES = 16     # T = 4: the window is full after batch 3, batches 4 and 5 prime the capture, batch 6 is captured, 7..15 are replays
LONG_GRAD_STRIDE = 32


def gen_traces_long():
    """The training loop body of train_LSTEP_link_prediction.py:204-311 over 16 consecutive batches of the trace graph (same graph, weights
    and first 7 batches as ``gen_traces``), with EVERY step's losses, probabilities, snapshot and parameter gradients: the fixture the
    graph-replayed engine iteration (engine.GraphedTrainStep) is held to."""
    g = synth.make_temporal_graph(**TRACE_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=31)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=32)
    sampler = ref_sampler(g)
    out = {}
    model = ref_model(node_raw, edge_raw, sampler, TRACE_K, TRACE_T)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    state = protocol.ProtocolState(history=torch.zeros(g["num_nodes"] + 1, 0, synth.PE_DIM), initial_pe=torch.from_numpy(pe0.copy()))
    # distance of the step's pre-activations from the relu kinks (edge channel models/LSTEP.py:164-166, neighbourhood PE :241-242, predictor
    # models/modules.py:66): a value within float32 rounding of 0 makes the reference's OWN gradient a coin toss (relu'(0-) = 0, relu'(0+) = 1),
    # and the whole contribution of that row to the layers in front of the relu flips with it.  The tests hold a step's gradients to the
    # tight bar only when the reference is not sitting on a kink.
    kink = []
    hook = lambda mod, inp, out: kink.append(float(out.detach().abs().min()))  # noqa: E731
    for m in (model[0].edge_agg, model[0].pe_neighbor_mlp_1, model[1].fc1):
        m.register_forward_hook(hook)
    for b in range(LONG_BATCHES):
        sl = slice(TRACE_START + b * TRACE_B, TRACE_START + (b + 1) * TRACE_B)
        neg = synth.make_negatives(g["num_nodes"], TRACE_B, seed=500 + b)
        kink.clear()
        res = protocol.train_iteration(model[0], model[1], opt, state, b, g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], neg,
                                       TRACE_K, TRACE_G, TRACE_T)
        out[f"b{b}/snapshot"] = state.history[:, -1, :].numpy().copy()
        if res is None:
            continue
        out[f"b{b}/losses"] = np.asarray([res["lp_loss"], res["pe_loss"], res["loss"]])
        out[f"b{b}/predicts"] = res["predicts"]
        out[f"b{b}/kink_distance"] = np.asarray([min(kink)])
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            a = p.grad.detach().numpy()
            a = np.stack([a.real, a.imag], -1) if np.iscomplexobj(a) else a.copy()
            out[f"b{b}/grads/{k}/digest"] = np.asarray([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
            out[f"b{b}/grads/{k}"] = a[::LONG_GRAD_STRIDE] if a.size > 20000 else a
    out["final_history"] = state.history[:, -TRACE_T:, :].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "traces_long.npz"), **out)
    print("traces_long.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- the reference in float64 (pins oracle.float64_yardstick)
F64_GRAPH = dict(num_nodes=64, num_edges=2000, seed=40, time_span=4096.0, tie_quantum=0.125)    # every time and time difference is exact in float32
F64_K, F64_T = 5, 6


def gen_float64():
    """The reference CLASSES evaluated in float64: parameters and feature tables widened, every nn.Linear input widened by a forward
    pre-hook (the reference casts time differences to float32 explicitly, models/LSTEP.py:153,228,277,314; the float32 VALUES stay,
    only the arithmetic after them is double), torch's default dtype float64 while it runs (the dense scatter targets of :282,319 are
    ``torch.zeros`` of the default dtype).  ``torch.Tensor([current_time])`` (:277) follows the default dtype too, so the graph's
    timestamps are multiples of 1/8 below 4096: every timestamp and every difference is exact in float32 and the reference's float32
    roundings are the identity either way."""
    g = synth.make_temporal_graph(**F64_GRAPH)
    assert np.array_equal(g["ts"], g["ts"].astype(np.float32).astype(np.float64))
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=41)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=42).astype(np.float64)
    sampler = ref_sampler(g)
    model = ref_model(node_raw, edge_raw, sampler, F64_K, F64_T).double()
    bb = model[0]
    bb.fft_filter.weight = torch.nn.Parameter(bb.fft_filter.weight.detach().to(torch.complex128))
    bb.node_raw_features, bb.edge_raw_features = bb.node_raw_features.double(), bb.edge_raw_features.double()
    for m in model.modules():
        if isinstance(m, torch.nn.Linear):
            m.register_forward_pre_hook(lambda mod, inp: (inp[0].to(mod.weight.dtype),))
    out = {}
    torch.set_default_dtype(torch.float64)
    try:
        with torch.no_grad():
            pe = torch.from_numpy(pe0.copy())
            sl = slice(40, 56)        # one early update_pe makes the padding row live (as ``live_pe``)
            bn = protocol.unique_batch_nodes(g["src"][sl], g["dst"][sl])
            bb.update_pe(pe, bn, g["eid"][sl], g["src"][sl], g["dst"][sl], g["ts"][sl], g["ts"][sl].max(), num_neighbors=F64_K)
            out["pe_live"] = pe.numpy().copy()
            for tag, sl in {"mid": slice(1200, 1216), "early": slice(3, 19)}.items():
                src, dst, t = g["src"][sl], g["dst"][sl], g["ts"][sl]
                out[f"{tag}/out_src"] = bb.combining_pe_raw_feat(pe, src, t, F64_K, 2000).numpy()
                out[f"{tag}/out_dst"] = bb.combining_pe_raw_feat(pe, dst, t, F64_K, 8).numpy()
            sl = slice(1200, 1216)
            src, dst, t, eid = g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl]
            bn = protocol.unique_batch_nodes(src, dst)
            pe_in = pe.clone()
            bb.update_pe(pe_in, bn, eid, src, dst, t, t.max(), num_neighbors=F64_K)
            out["update/pe_out"] = pe_in.numpy().copy()
            rng = np.random.RandomState(43)
            hist = 0.1 * rng.standard_normal((g["num_nodes"] + 1, F64_T, synth.PE_DIM))
            ids = np.asarray([1, 2, 5, 9, 17, 33, 64, 0], dtype=np.int64)
            out["fft/ids"] = ids
            for stored, bidx in ((3, 3), (F64_T, 9)):
                out[f"fft/stored{stored}_b{bidx}"] = bb.fourier_transform_pe(ids, torch.from_numpy(hist[:, :stored, :].copy()), bidx).numpy()
    finally:
        torch.set_default_dtype(torch.float32)
    assert all(v.dtype == np.float64 for k, v in out.items() if k != "fft/ids"), {k: v.dtype for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "float64.npz"), **out)
    print("float64.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- row P pinned by the reference's own loop
EVAL_LOOP = dict(first=1200, edges=6 * 16 + 5, batch=16, stored=2)      # a ragged tail batch of 5 edges (drop_last=False)


def gen_eval_loop():
    """The reference's OWN evaluation loop (``evaluate_model_utils.evaluate_model_link_prediction``, an importable function -- unlike the
    training loop, which is a script body) on a tiny ``Data`` with its own index ``DataLoader`` and seeded ``NegativeEdgeSampler``s.
    Nothing of ``lstep_amd.protocol`` is involved: the fixture pins ``protocol.eval_iteration``, the oracle and the engine to
    reference-executed loop code.  Recorded through hooks: the negatives the sampler drew (they are inputs of the hot path), the
    link predictor's logits, the table every ``update_pe`` returned; returned by the function: the per-batch losses."""
    from evaluate_model_utils import evaluate_model_link_prediction
    from utils.DataLoader import get_idx_data_loader
    from utils.utils import NegativeEdgeSampler
    g = synth.make_temporal_graph(**TRACE_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=31)
    sampler = ref_sampler(g)
    lo, n, bsz = EVAL_LOOP["first"], EVAL_LOOP["edges"], EVAL_LOOP["batch"]
    sl = slice(lo, lo + n)
    data = Data(g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], np.zeros(n))
    rng = np.random.RandomState(61)
    hist0 = (0.1 * rng.standard_normal((g["num_nodes"] + 1, EVAL_LOOP["stored"], synth.PE_DIM))).astype(np.float32)
    out = {"history0": hist0}
    for strategy in ("random", "historical"):
        model = ref_model(node_raw, edge_raw, sampler, TRACE_K, TRACE_T)
        neg = NegativeEdgeSampler(src_node_ids=g["src"], dst_node_ids=g["dst"], interact_times=g["ts"], negative_sample_strategy=strategy, seed=2)
        drawn, logits, tables = [], [], []
        sample = neg.sample

        def recording_sample(*a, _sample=sample, **k):
            res = _sample(*a, **k)
            drawn.append(res)
            return res

        neg.sample = recording_sample
        update = model[0].update_pe

        def recording_update(*a, _update=update, **k):
            res = _update(*a, **k)
            tables.append(res.detach().numpy().copy())
            return res

        model[0].update_pe = recording_update
        hook = model[1].register_forward_hook(lambda mod, args, res: logits.append(res.detach().numpy().copy()))
        loader = get_idx_data_loader(indices_list=list(range(n)), batch_size=bsz, shuffle=False)
        losses, metrics = evaluate_model_link_prediction(model_name="LSTEP", model=model, final_trained_positional_encoding=torch.from_numpy(hist0.copy()),
                                                         neighbor_sampler=sampler, evaluate_idx_data_loader=loader, evaluate_neg_edge_sampler=neg,
                                                         evaluate_data=data, loss_func=torch.nn.BCELoss(), num_fft_batches=TRACE_T,
                                                         num_neighbors=TRACE_K, time_gap=TRACE_G)
        hook.remove()
        nb = len(losses)
        assert nb == (n + bsz - 1) // bsz == len(drawn) == len(tables) and len(logits) == 2 * nb
        out[f"{strategy}/losses"] = np.asarray(losses, dtype=np.float64)
        out[f"{strategy}/average_precision"] = np.asarray([m["average_precision"] for m in metrics])
        for b in range(nb):
            out[f"{strategy}/b{b}/neg_src"] = np.asarray(drawn[b][0], dtype=np.int64)
            out[f"{strategy}/b{b}/neg_dst"] = np.asarray(drawn[b][1], dtype=np.int64)
            out[f"{strategy}/b{b}/pos_logits"] = logits[2 * b].reshape(-1)
            out[f"{strategy}/b{b}/neg_logits"] = logits[2 * b + 1].reshape(-1)
            out[f"{strategy}/b{b}/snapshot"] = tables[b]
    np.savez_compressed(os.path.join(HERE, "eval_loop.npz"), **out)
    print("eval_loop.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- weighted_sum + RNG strategies through the model
WS_GRAPH = dict(num_nodes=40, num_edges=1200, seed=90, time_span=60.0, tie_quantum=0.25)    # times a few units apart (exp(-dt) is not all 0), with ties


def gen_variants():
    """(1) the `weighted_sum` ablation of the node channel (models/LSTEP.py:190-206, --ablation weighted_sum) with 'recent' sampling;
    (2) the RNG-defined sampling strategies driving combining_pe_raw_feat / update_pe (three independent draws per combine call)."""
    from utils.utils import get_neighbor_sampler as ref_get
    g = synth.make_temporal_graph(**WS_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=91)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=92)
    pe0[0] = 0.03
    data = Data(g["src"], g["dst"], g["ts"], g["eid"], np.zeros(len(g["src"])))
    K, T = 5, 4
    out = {}
    sl = slice(900, 924)
    src, dst, t, eid = g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl]
    # (1)
    model = ref_model(node_raw, edge_raw, ref_get(data, sample_neighbor_strategy="recent", seed=None), K, T, weighted_sum=True)
    with torch.no_grad():
        for G in (6, 2000):
            out[f"ws/agg_G{G}"] = model[0].aggregated_node_embeddings(src, t, K, G).numpy()
            out[f"ws/out_G{G}"] = model[0].combining_pe_raw_feat(torch.from_numpy(pe0.copy()), dst, t, K, G).numpy()
    # (2)
    for strat, tsf in (("uniform", 0.0), ("time_interval_aware", 1e-2)):
        for ws in (False, True):
            sampler = ref_get(data, sample_neighbor_strategy=strat, time_scaling_factor=tsf, seed=5)
            model = ref_model(node_raw, edge_raw, sampler, K, T, weighted_sum=ws)
            model[0].set_neighbor_sampler(sampler)
            tag = f"{strat}/ws{int(ws)}"
            pe = torch.from_numpy(pe0.copy())
            with torch.no_grad():
                out[f"{tag}/out_src"] = model[0].combining_pe_raw_feat(pe, src, t, K, 7).numpy()
                out[f"{tag}/out_dst"] = model[0].combining_pe_raw_feat(pe, dst, t, K, 7).numpy()
                out[f"{tag}/agg"] = model[0].aggregated_node_embeddings(src, t, K, 7).numpy()
                out[f"{tag}/cpe"] = model[0].compute_neighborhood_pe(pe, dst, t, K).numpy()
                bn = protocol.unique_batch_nodes(src, dst)
                out[f"{tag}/pe_updated"] = model[0].update_pe(pe, bn, eid, src, dst, t, t.max(), num_neighbors=K).numpy().copy()
            # gradients of one loss through the three draws
            pe_g = torch.from_numpy(pe0.copy()).requires_grad_(True)
            w = torch.from_numpy(np.random.RandomState(93).standard_normal((len(src), synth.FEAT_DIM)).astype(np.float32))
            (model[0].combining_pe_raw_feat(pe_g, src, t, K, 7) * w).sum().backward()
            out[f"{tag}/grad_pe"] = pe_g.grad.numpy().copy()
            out[f"{tag}/grad_edge_agg"] = model[0].edge_agg.weight.grad.numpy().copy()
            out[f"{tag}/grad_edge_mlp_1"] = model[0].edge_mlp_1.weight.grad.numpy()[::GRAD_ROW_STRIDE].copy()
    np.savez_compressed(os.path.join(HERE, "variants.npz"), **out)
    print("variants.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- RNG-defined sampling
def gen_random_sampling():
    from utils.utils import get_neighbor_sampler as ref_get
    g = synth.make_temporal_graph(num_nodes=40, num_edges=900, seed=50, tie_quantum=5.0)
    data = Data(g["src"], g["dst"], g["ts"], g["eid"], np.zeros(900))
    rng = np.random.RandomState(51)
    ids = rng.randint(0, 41, size=60).astype(np.int64)
    ts = rng.uniform(g["ts"][0] - 1, g["ts"][-1] + 1, size=60)
    out = {"ids": ids, "ts": ts}
    for strat, tsf in (("uniform", 0.0), ("time_interval_aware", 1e-3)):
        s = ref_get(data, sample_neighbor_strategy=strat, time_scaling_factor=tsf, seed=3)
        for call in range(2):  # the RNG state carries over between calls
            for k in (3, 10):
                nbr, eid, nt = s.get_historical_neighbors(ids, ts, k)
                out[f"{strat}/call{call}/k{k}/nbr"], out[f"{strat}/call{call}/k{k}/eid"], out[f"{strat}/call{call}/k{k}/nt"] = nbr, eid, nt
        s.reset_random_state()
        nbr, eid, nt = s.get_historical_neighbors(ids, ts, 3)
        out[f"{strat}/reset/k3/nbr"] = nbr
    np.savez_compressed(os.path.join(HERE, "random_sampling.npz"), **out)
    print("random_sampling.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- data loader split
def loader_dataset(tmp):
    """Tiny dataset in the reference's on-disk format (regenerated identically by tests/test_data_loader.py)."""
    from lstep_amd import data as ld
    g = synth.make_temporal_graph(num_nodes=120, num_edges=1500, seed=40, tie_quantum=7.0)
    rng = np.random.RandomState(41)
    ld.write_dataset(os.path.join(tmp, "processed_data"), "tiny", g["src"], g["dst"], g["ts"], rng.randint(0, 2, size=1500),
                     rng.standard_normal((1501, 12)), np.zeros((121, 172)))


def gen_loader():
    import tempfile
    from utils.DataLoader import get_link_prediction_data
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        loader_dataset(tmp)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            res = get_link_prediction_data("tiny", 0.15, 0.15)
        finally:
            os.chdir(cwd)
    out["node_shape"], out["edge_shape"] = np.asarray(res[0].shape), np.asarray(res[1].shape)
    out["edge_digest"] = np.asarray([res[1].sum(), np.abs(res[1]).sum()])
    for name, d in zip(("full", "train", "val", "test", "new_val", "new_test"), res[2:]):
        out[f"{name}/edge_ids"] = d.edge_ids
        out[f"{name}/src"] = d.src_node_ids
        out[f"{name}/num_unique_nodes"] = np.asarray([d.num_unique_nodes])
    np.savez_compressed(os.path.join(HERE, "loader.npz"), **out)
    print("loader.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- initial positional encodings
INIT_PE = dict(num_nodes=40, k=6, walk=5, seed=77)


def init_pe_graph():
    """First-batch-shaped input of train_LSTEP_link_prediction.py:168-189: both directions of a batch's edges (sources first), on a
    connected multigraph (a ring plus random chords, some edges repeated) over all nodes but three isolated ones."""
    rng = np.random.RandomState(INIT_PE["seed"])
    n = INIT_PE["num_nodes"] - 3
    ring = np.stack([np.arange(n), (np.arange(n) + 1) % n])
    chords = rng.randint(0, n, size=(2, 30))
    chords = chords[:, chords[0] != chords[1]]
    e = np.concatenate([ring, chords, chords[:, :5]], axis=1)          # five parallel edges
    return np.stack([np.concatenate([e[0], e[1]]), np.concatenate([e[1], e[0]])]).astype(np.int64)


def gen_init_pe():
    """``utils/PositionalEncoding.py`` (RandomWalkPE :69-91, LaplacianPE :42-62) run on ``torch_geometric.utils`` SHIMS: the wheel is not
    installed (nor pinned by the reference), so the six utilities the file calls are restated from their documented semantics, like the
    ``torch_scatter`` shim: ``scatter(reduce='sum')`` = index_add into zeros; ``to_torch_csr_tensor`` = the coalesced (duplicates summed) sparse
    CSR matrix of (edge_index, edge_attr); ``to_edge_index`` = its COO indices and values; ``get_self_loop_attr`` = the diagonal entries, 0
    where there is none; ``get_laplacian(normalization='sym')`` = self-loops removed, then I - D^-1/2 A D^-1/2 as (edge_index + loops,
    [-normalised weights, ones]); ``to_scipy_sparse_matrix`` = scipy COO of those."""
    import scipy.sparse
    tg = types.ModuleType("torch_geometric")
    tu = types.ModuleType("torch_geometric.utils")

    def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
        assert reduce == "sum" and dim == 0
        return torch.zeros(dim_size, dtype=src.dtype).index_add_(0, index, src)

    def to_torch_csr_tensor(edge_index, edge_attr=None, size=None, is_coalesced=False):
        n = size if isinstance(size, int) else size[0]
        return torch.sparse_coo_tensor(edge_index, edge_attr, (n, n)).coalesce().to_sparse_csr()

    def to_edge_index(adj):
        coo = adj.to_sparse_coo().coalesce()
        return coo.indices(), coo.values()

    def get_self_loop_attr(edge_index, edge_attr=None, num_nodes=None):
        mask = edge_index[0] == edge_index[1]
        out = torch.zeros(num_nodes, dtype=edge_attr.dtype)
        out[edge_index[0][mask]] = edge_attr[mask]
        return out

    def get_laplacian(edge_index, edge_weight=None, normalization=None, dtype=None, num_nodes=None):
        assert normalization == "sym" and edge_weight is None
        keep = edge_index[0] != edge_index[1]
        edge_index = edge_index[:, keep]
        w = torch.ones(edge_index.shape[1])
        row, col = edge_index
        deg = torch.zeros(num_nodes).index_add_(0, row, w)
        dis = deg.pow(-0.5)
        dis.masked_fill_(dis == float("inf"), 0)
        w = dis[row] * w * dis[col]
        loops = torch.arange(num_nodes)
        return torch.cat([edge_index, torch.stack([loops, loops])], dim=1), torch.cat([-w, torch.ones(num_nodes)])

    def to_scipy_sparse_matrix(edge_index, edge_attr=None, num_nodes=None):
        return scipy.sparse.coo_matrix((edge_attr.numpy(), (edge_index[0].numpy(), edge_index[1].numpy())), (num_nodes, num_nodes))

    tu.scatter, tu.to_torch_csr_tensor, tu.to_edge_index, tu.get_self_loop_attr = scatter, to_torch_csr_tensor, to_edge_index, get_self_loop_attr
    tu.get_laplacian, tu.to_scipy_sparse_matrix, tu.is_torch_sparse_tensor = get_laplacian, to_scipy_sparse_matrix, (lambda x: x.is_sparse)
    tg.utils = tu
    sys.modules.update({"torch_geometric": tg, "torch_geometric.utils": tu})
    from utils.PositionalEncoding import LaplacianPE, RandomWalkPE
    ei = torch.from_numpy(init_pe_graph())
    N, k, walk = INIT_PE["num_nodes"], INIT_PE["k"], INIT_PE["walk"]
    out = {"edge_index": ei.numpy()}
    out["rwpe"] = RandomWalkPE(ei, N, walk).numpy()
    torch.manual_seed(5)
    pe, edge_weight = LaplacianPE(ei, N, k)
    out["lappe_abs"] = np.abs(pe.numpy())                  # (columns carry a random sign, :57-59; eigenvectors are defined up to sign anyway)
    out["lappe_edge_weight"] = edge_weight.numpy()
    lap = to_scipy_sparse_matrix(*get_laplacian(ei, normalization="sym", num_nodes=N), N).toarray()
    ev = np.linalg.eigvalsh(lap)
    assert np.min(np.diff(ev[:k + 2])) > 1e-3, "the fixture needs non-degenerate small eigenvalues (a unique answer up to sign)"
    out["laplacian"] = lap
    np.savez_compressed(os.path.join(HERE, "init_pe.npz"), **out)
    print("init_pe.npz", len(out), "arrays")


# ----------------------------------------------------------------------------------------------- use_dropout (models/LSTEP.py:131-133,171-172)
def gen_dropout():
    """`use_dropout=True` -- never set by the reference's drivers, pinned where it is deterministic: (1) the model flag with p = 0 (the
    functional dropout of models/LSTEP.py:171-172 is then the identity: the plain, un-premultiplied tail), (2) fourier_transform_pe's
    `use_dropout=True` argument in eval mode (nn.Dropout is the identity there; what remains is the residual `batch_pe += init_pe` of
    models/LSTEP.py:131-133), full and short history."""
    from utils.utils import get_neighbor_sampler as ref_get
    g = synth.make_temporal_graph(**WS_GRAPH)
    node_raw, edge_raw = synth.make_features(g["num_nodes"], len(g["eid"]), seed=91)
    pe0 = synth.make_initial_pe(g["num_nodes"], seed=92)
    pe0[0] = 0.03
    data = Data(g["src"], g["dst"], g["ts"], g["eid"], np.zeros(len(g["src"])))
    K, T = 5, 4
    sl = slice(900, 924)
    src, dst, t = g["src"][sl], g["dst"][sl], g["ts"][sl]
    sampler = ref_get(data, sample_neighbor_strategy="recent", seed=None)
    torch.manual_seed(0)
    bb = LSTEP(node_raw_features=node_raw, edge_raw_features=edge_raw, neighbor_sampler=sampler, full_neighbor_sampler=sampler,
               pe_dim=synth.PE_DIM, num_neighbors=K, time_feat_dim=synth.TIME_DIM, num_fft_batches=T, use_dropout=True, dropout=0.0, device="cpu")
    model = torch.nn.Sequential(bb, MergeLayer(input_dim1=synth.FEAT_DIM, input_dim2=synth.FEAT_DIM, hidden_dim=synth.FEAT_DIM, output_dim=1))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(K, T, seed=3).items()}, strict=True)
    out = {}
    with torch.no_grad():
        for G in (6, 2000):
            out[f"p0/agg_G{G}"] = bb.aggregated_node_embeddings(src, t, K, G).numpy()
            out[f"p0/out_G{G}"] = bb.combining_pe_raw_feat(torch.from_numpy(pe0.copy()), dst, t, K, G).numpy()
    model.eval()
    rng = np.random.RandomState(94)
    ids = np.unique(np.concatenate([src, dst]))
    out["fft/ids"] = ids
    for name, t_len, batch_idx in (("full", T, T + 3), ("short", 2, 2), ("short_idx1", 3, 1)):
        hist = rng.standard_normal((g["num_nodes"] + 1, t_len, synth.PE_DIM)).astype(np.float32) * 0.1
        out[f"fft/{name}/hist"] = hist
        with torch.no_grad():
            out[f"fft/{name}/out"] = bb.fourier_transform_pe(ids, torch.from_numpy(hist), batch_idx, use_dropout=True).numpy()
            out[f"fft/{name}/out_plain"] = bb.fourier_transform_pe(ids, torch.from_numpy(hist), batch_idx).numpy()
    np.savez_compressed(os.path.join(HERE, "dropout.npz"), **out)
    print("dropout.npz", len(out), "arrays")


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["sampler", "time", "methods", "traces", "loader", "random_sampling", "eval_loop", "variants", "traces_long", "float64", "init_pe", "dropout"]
    if "sampler" in which:
        gen_sampler()
    if "time" in which:
        gen_time_encoder()
    if "methods" in which:
        gen_methods()
    if "traces" in which:
        gen_traces()
    if "loader" in which:
        gen_loader()
    if "random_sampling" in which:
        gen_random_sampling()
    if "eval_loop" in which:
        gen_eval_loop()
    if "variants" in which:
        gen_variants()
    if "traces_long" in which:
        gen_traces_long()
    if "float64" in which:
        gen_float64()
    if "init_pe" in which:
        gen_init_pe()
    if "dropout" in which:
        gen_dropout()
