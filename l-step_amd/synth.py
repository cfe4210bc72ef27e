"""Seeded synthetic inputs for the L-STEP hot path (no dataset ships with the reference).

Everything here is drawn from ``numpy.random.RandomState`` (the frozen legacy
generator), so the same seed gives bit-identical arrays in the build container,
on the GPU box and in ``tests/golden/make_golden.py``.  Shapes follow the
reference's data contract: node ids and edge ids start at 1, row 0 of every
feature table is the padding row (reference ``preprocess_data/preprocess_data.py:56-81,108``),
timestamps are float64 and non-decreasing in edge-id order
(``utils/DataLoader.py:199-264`` keeps the chronological order).
"""
from __future__ import annotations

import numpy as np

# model dimensions of the reference's defaults (utils/load_configs.py:36-37; features padded to 172
# in utils/DataLoader.py:185-196)
FEAT_DIM = 172
TIME_DIM = 100
PE_DIM = 172


def make_temporal_graph(num_nodes: int, num_edges: int, seed: int = 0, time_span: float | None = None,
                        tie_quantum: float | None = None, epoch_offset: float = 0.0,
                        zipf: float | None = None, bipartite_split: int | None = None):
    """Random temporal interaction stream.

    :return: dict with ``src``/``dst`` (int64, ids in 1..num_nodes), ``ts`` (float64, sorted), ``eid`` (int64, 1..E)
    ``tie_quantum`` rounds timestamps down to a grid so that many edges share a timestamp (tie-order tests);
    ``epoch_offset`` shifts them to epoch scale (float32 rounding of neighbour times, reference ``utils/utils.py:166``);
    ``zipf`` draws endpoints from a power law (hub skew); ``bipartite_split`` puts src in [1, split], dst in (split, N].
    """
    rng = np.random.RandomState(seed)
    if time_span is None:
        time_span = 1e6 * num_edges / 1e5
    if zipf is not None:
        ranks = np.arange(1, num_nodes + 1, dtype=np.float64)
        prob = ranks ** (-zipf)
        prob /= prob.sum()
        perm = rng.permutation(num_nodes) + 1
        src = perm[rng.choice(num_nodes, size=num_edges, p=prob)]
        dst = perm[rng.choice(num_nodes, size=num_edges, p=prob)]
    elif bipartite_split is not None:
        src = rng.randint(1, bipartite_split + 1, size=num_edges)
        dst = rng.randint(bipartite_split + 1, num_nodes + 1, size=num_edges)
    else:
        src = rng.randint(1, num_nodes + 1, size=num_edges)
        dst = rng.randint(1, num_nodes + 1, size=num_edges)
    ts = np.sort(rng.uniform(0.0, time_span, size=num_edges))
    if tie_quantum is not None:
        ts = np.floor(ts / tie_quantum) * tie_quantum
    ts = ts + epoch_offset
    return {
        "src": src.astype(np.int64),
        "dst": dst.astype(np.int64),
        "ts": ts.astype(np.float64),
        "eid": np.arange(1, num_edges + 1, dtype=np.int64),
        "num_nodes": int(num_nodes),
    }


def make_features(num_nodes: int, num_edges: int, feat_dim: int = FEAT_DIM, seed: int = 1):
    """N(0,1) node and edge feature tables with a zero padding row 0 (float32)."""
    rng = np.random.RandomState(seed)
    node_raw = rng.standard_normal((num_nodes + 1, feat_dim)).astype(np.float32)
    edge_raw = rng.standard_normal((num_edges + 1, feat_dim)).astype(np.float32)
    node_raw[0] = 0.0
    edge_raw[0] = 0.0
    return node_raw, edge_raw


def make_initial_pe(num_nodes: int, pe_dim: int = PE_DIM, seed: int = 2, scale: float = 0.1):
    """Seeded stand-in for the Laplacian initial PE (``utils/PositionalEncoding.py:42-62`` is not importable here)."""
    rng = np.random.RandomState(seed)
    return (scale * rng.standard_normal((num_nodes + 1, pe_dim))).astype(np.float32)


def _uniform(rng, shape, fan_in):
    bound = 1.0 / np.sqrt(fan_in)
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def make_state_dict(num_neighbors: int, num_fft_batches: int, feat_dim: int = FEAT_DIM, time_dim: int = TIME_DIM,
                    pe_dim: int = PE_DIM, seed: int = 3):
    """Weights for ``nn.Sequential(LSTEP, MergeLayer)`` keyed like the reference ``state_dict``
    (``models/LSTEP.py:50-72``, ``models/modules.py:53-54``, wrapped at ``train_LSTEP_link_prediction.py:142``).

    Values are numpy arrays (``0.fft_filter.weight`` is complex64); scale mimics ``nn.Linear``'s default init.
    """
    rng = np.random.RandomState(seed)
    c = feat_dim + time_dim
    sd = {}

    def lin(name, out_f, in_f, bias=True):
        sd[name + ".weight"] = _uniform(rng, (out_f, in_f), in_f)
        if bias:
            sd[name + ".bias"] = _uniform(rng, (out_f,), in_f)

    sd["0.time_encoder.w.weight"] = (1.0 / 10 ** np.linspace(0, 9, time_dim, dtype=np.float32)).reshape(time_dim, 1).astype(np.float32)
    sd["0.time_encoder.w.bias"] = np.zeros(time_dim, dtype=np.float32)
    re = _uniform(rng, (num_fft_batches, pe_dim), pe_dim)
    im = _uniform(rng, (num_fft_batches, pe_dim), pe_dim)
    sd["0.fft_filter.weight"] = (re + 1j * im).astype(np.complex64)
    lin("0.fft_agg", 1, num_fft_batches, bias=False)
    lin("0.edge_mlp_1", c, c)
    lin("0.edge_agg", 1, num_neighbors)
    lin("0.edge_mlp_2", c, c)
    lin("0.node_mlp", feat_dim, c + feat_dim)
    lin("0.self_update_pe", pe_dim, pe_dim)
    lin("0.pe_mlp_1", pe_dim, pe_dim + time_dim)
    lin("0.pe_mlp_2", pe_dim, pe_dim)
    lin("0.self_update_neighbor_pe", pe_dim, pe_dim)
    lin("0.pe_neighbor_mlp_1", pe_dim, pe_dim + time_dim)
    lin("0.pe_neighbor_mlp_2", pe_dim, pe_dim)
    lin("0.out_node_emb", feat_dim, pe_dim + feat_dim)
    lin("1.fc1", feat_dim, 2 * feat_dim)
    lin("1.fc2", 1, feat_dim)
    return sd


def make_negatives(num_nodes: int, size: int, seed: int):
    """Random negative destinations (the reference's ``NegativeEdgeSampler`` is an RNG-defined input generator,
    ``utils/utils.py:304-494``; the hot path only consumes its ids)."""
    rng = np.random.RandomState(seed)
    return rng.randint(1, num_nodes + 1, size=size).astype(np.int64)


def batch_slices(num_edges: int, batch_size: int):
    """Chronological, unshuffled batches of edge positions (``train_LSTEP_link_prediction.py:57-61``)."""
    return [np.arange(s, min(s + batch_size, num_edges)) for s in range(0, num_edges, batch_size)]
