"""On-disk dataset format of the reference (DyGLib ``processed_data/{name}/ml_{name}.csv|.npy|_node.npy``) -> device streams.

SURVEY.md 8(f) rank 3 ("next" row).  Mirrors reference ``utils/DataLoader.py``: ``Data`` (``:68-86``) and
``get_link_prediction_data`` (``:171-279``): feature padding to 172 columns (``:185-196``), chronological 70/15/15 split at
the ts quantiles (``:199``), the inductive hold-out of 10 % of the nodes that appear after the validation time drawn with
``random.seed(2020)`` (``:211-224``), and the six ``Data`` views.  File format (``preprocess_data/preprocess_data.py:49-53,115-117``):
csv columns ``u, i, ts, label, idx`` (node ids and edge ids start at 1), ``ml_{name}.npy`` = edge features ``[E+1, d]``,
``ml_{name}_node.npy`` = node features ``[N+1, d]``, row 0 = padding.

The split is host logic (numpy + Python ``random``, exactly the reference's draw order, so the same nodes are held out); the
arrays then move to the GPU once: ``to_edge_stream`` / ``lstep_amd.sampler.get_neighbor_sampler``.
"""
from __future__ import annotations

import os
import random

import numpy as np
import pandas as pd

FEAT_DIM = 172


class Data:
    """Same attributes as reference ``utils.DataLoader.Data``."""

    def __init__(self, src_node_ids, dst_node_ids, node_interact_times, edge_ids, labels):
        self.src_node_ids = src_node_ids
        self.dst_node_ids = dst_node_ids
        self.node_interact_times = node_interact_times
        self.edge_ids = edge_ids
        self.labels = labels
        self.num_interactions = len(src_node_ids)
        self.unique_node_ids = set(src_node_ids) | set(dst_node_ids)
        self.num_unique_nodes = len(self.unique_node_ids)

    def select(self, mask):
        return Data(self.src_node_ids[mask], self.dst_node_ids[mask], self.node_interact_times[mask], self.edge_ids[mask], self.labels[mask])


def _pad_features(x: np.ndarray, width: int, what: str, name: str) -> np.ndarray:
    assert width >= x.shape[1], f"{what} feature dimension in dataset {name} is bigger than {width}!"
    if x.shape[1] < width:
        x = np.concatenate([x, np.zeros((x.shape[0], width - x.shape[1]))], axis=1)
    return x


def get_link_prediction_data(dataset_name: str, val_ratio: float, test_ratio: float, root: str = "./processed_data"):
    """Returns ``node_raw_features, edge_raw_features, full_data, train_data, val_data, test_data, new_node_val_data,
    new_node_test_data`` exactly like the reference loader."""
    base = os.path.join(root, dataset_name, f"ml_{dataset_name}")
    df = pd.read_csv(base + ".csv")
    edge_raw = _pad_features(np.load(base + ".npy"), FEAT_DIM, "Edge", dataset_name)
    node_raw = _pad_features(np.load(base + "_node.npy"), FEAT_DIM, "Node", dataset_name)

    val_time, test_time = list(np.quantile(df.ts, [1 - val_ratio - test_ratio, 1 - test_ratio]))
    src = df.u.values.astype(np.longlong)
    dst = df.i.values.astype(np.longlong)
    ts = df.ts.values.astype(np.float64)
    full = Data(src, dst, ts, df.idx.values.astype(np.longlong), df.label.values)

    # inductive hold-out: same RNG, same population construction and draw as the reference (DataLoader.py:211-224)
    random.seed(2020)
    node_set = set(src) | set(dst)
    late = ts > val_time
    test_node_set = set(src[late]).union(set(dst[late]))
    held_out = set(random.sample(tuple(test_node_set), int(0.1 * len(node_set))))
    held = np.fromiter(held_out, dtype=np.longlong, count=len(held_out))
    observed = ~np.isin(src, held) & ~np.isin(dst, held)

    train = full.select((ts <= val_time) & observed)
    train_nodes = set(train.src_node_ids).union(train.dst_node_ids)
    assert len(train_nodes & held_out) == 0
    new_nodes = np.fromiter(node_set - train_nodes, dtype=np.longlong)
    touches_new = np.isin(src, new_nodes) | np.isin(dst, new_nodes)
    val_mask = (ts <= test_time) & (ts > val_time)
    test_mask = ts > test_time
    return (node_raw, edge_raw, full, train, full.select(val_mask), full.select(test_mask),
            full.select(val_mask & touches_new), full.select(test_mask & touches_new))


def to_edge_stream(data: Data, device="cuda"):
    """Chronological device-resident edge arrays for ``lstep_amd.engine`` (one host->device copy per split)."""
    from .engine import EdgeStream
    return EdgeStream.from_numpy(data.src_node_ids, data.dst_node_ids, data.node_interact_times, data.edge_ids, device)


def write_dataset(root: str, name: str, src, dst, ts, labels, edge_feat: np.ndarray, node_feat: np.ndarray):
    """Write arrays in the reference's processed format (used by tests and for exporting synthetic graphs)."""
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    e = len(src)
    pd.DataFrame({"u": src, "i": dst, "ts": ts, "label": labels, "idx": np.arange(1, e + 1)}).to_csv(os.path.join(d, f"ml_{name}.csv"))
    np.save(os.path.join(d, f"ml_{name}.npy"), edge_feat)
    np.save(os.path.join(d, f"ml_{name}_node.npy"), node_feat)
