"""CPU ORACLE for the L-STEP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-fp32 / numpy restatement of the reference algorithm (kthrn22/L-STEP), used ONLY as the checker:
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``l-step_amd/`` does, and the product path raises when the HIP library is missing instead of falling back here.

Parity pin: every function below is checked against golden vectors produced by running the reference itself
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).  The reference has no
tests or fixtures of its own, so those vectors are the pin.

The op ORDER deliberately follows the reference (dense ``[B, time_gap, F]`` gather, dense ``[N+1, P+D]`` scatter
targets, a Python loop over rows in the sampler): this file doubles as the CPU baseline the GPU path is timed
against, so it must cost what the reference costs.

Reference lines followed (``/root/reference``):
  sampler        utils/utils.py:72-109 (adjacency), :129-146 (search), :148-213 (recent strategy), :282-301 (build)
  time encoder   models/modules.py:7-39 (frozen by models/LSTEP.py:50)
  merge layer    models/modules.py:42-68
  backbone       models/LSTEP.py:29-74 (parameters), :104-137 (FFT filter), :139-220 (edge + node aggregation),
                 :222-249 (neighbourhood PE), :251-266 (combine), :268-340 (update_pe, both phases)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


# --------------------------------------------------------------------------------------------------- sampler
class OracleNeighborSampler:
    """Time-sorted undirected adjacency + 'recent' historical-neighbour lookup.

    Storage is CSR (three flat arrays + offsets) instead of the reference's per-node Python lists; the content and
    the order inside each node's slice are the reference's: every edge is appended to both endpoints
    (src's list first, then dst's: utils/utils.py:297-299) and each list is STABLY sorted by timestamp (:99).
    """

    sample_neighbor_strategy = "recent"

    def __init__(self, src, dst, eid, ts, num_nodes=None, seed=None, _force_lexsort=False):
        src = np.asarray(src, dtype=np.int64)
        dst = np.asarray(dst, dtype=np.int64)
        eid = np.asarray(eid, dtype=np.int64)
        ts = np.asarray(ts, dtype=np.float64)
        e = len(src)
        top = int(max(src.max(), dst.max())) if e else 0
        self.num_rows = max(top, int(num_nodes) if num_nodes is not None else 0) + 1
        owner = np.empty(2 * e, dtype=np.int64)
        other = np.empty(2 * e, dtype=np.int64)
        owner[0::2], owner[1::2] = src, dst
        other[0::2], other[1::2] = dst, src
        eids = np.repeat(eid, 2)
        tss = np.repeat(ts, 2)
        if e and bool(np.all(ts[1:] >= ts[:-1])) and not _force_lexsort:
            # chronological input (every data file of the reference is): insertion order already IS time order with ties in
            # insertion order, so one stable sort by owner gives the same permutation as the three-key sort below
            # (tests/test_oracle_golden.py checks the two against each other); minutes -> seconds at 2 x 20 M entries
            order = np.argsort(owner, kind="stable")
        else:
            order = np.lexsort((np.arange(2 * e), tss, owner))  # owner major, then time, then insertion order (= stable)
        self.nbr = other[order]
        self.eid = eids[order]
        self.ts = tss[order]
        self.indptr = np.zeros(self.num_rows + 1, dtype=np.int64)
        np.cumsum(np.bincount(owner, minlength=self.num_rows), out=self.indptr[1:])
        self.seed = seed

    def reset_random_state(self):  # API parity with utils/utils.py:274-279; 'recent' draws nothing
        pass

    def count_before(self, node_id: int, t: float) -> int:
        """Number of the node's interactions strictly earlier than t (``np.searchsorted`` side='left', :140)."""
        lo, hi = self.indptr[node_id], self.indptr[node_id + 1]
        return int(np.searchsorted(self.ts[lo:hi], t))

    def get_historical_neighbors(self, node_ids, node_interact_times, num_neighbors=20):
        assert num_neighbors > 0, "Number of sampled neighbors for each node should be greater than 0!"
        rows = len(node_ids)
        nbr = np.zeros((rows, num_neighbors), dtype=np.int64)
        eid = np.zeros((rows, num_neighbors), dtype=np.int64)
        nts = np.zeros((rows, num_neighbors), dtype=np.float32)
        # zip() stops at the shorter input: rows beyond it stay all-padding (:169)
        for r, (node, t) in enumerate(zip(node_ids, node_interact_times)):
            lo = self.indptr[node]
            cnt = self.count_before(node, t)
            take = min(cnt, num_neighbors)
            if take:
                a, b = lo + cnt - take, lo + cnt
                nbr[r, num_neighbors - take:] = self.nbr[a:b]  # right-aligned (:206-208)
                eid[r, num_neighbors - take:] = self.eid[a:b]
                nts[r, num_neighbors - take:] = self.ts[a:b]   # float64 -> float32 (:166)
        return nbr, eid, nts


def oracle_neighbor_sampler(src, dst, eid, ts, num_nodes=None):
    return OracleNeighborSampler(src, dst, eid, ts, num_nodes=num_nodes)


# --------------------------------------------------------------------------------------------------- small modules
class OracleTimeEncoder(nn.Module):
    """cos(t * w + b) with w_i = 10^(-9 i / (D-1)), b = 0, both frozen (models/modules.py:19-25, LSTEP.py:50)."""

    def __init__(self, time_dim: int):
        super().__init__()
        self.w = nn.Linear(1, time_dim)
        with torch.no_grad():
            self.w.weight.copy_(torch.from_numpy(1.0 / 10 ** np.linspace(0, 9, time_dim, dtype=np.float32)).reshape(time_dim, 1))
            self.w.bias.zero_()
        self.w.weight.requires_grad_(False)
        self.w.bias.requires_grad_(False)

    def forward(self, dt: torch.Tensor) -> torch.Tensor:  # [..] -> [.., D]
        # (the cast is a no-op in the fp32 model; the float64 yardstick of ``float64_yardstick`` keeps the reference's float32
        # rounding of dt and only then widens)
        return torch.cos(self.w(dt.unsqueeze(-1).to(self.w.weight.dtype)))


class OracleMergeLayer(nn.Module):
    """fc2(relu(fc1(cat[a, b])))  (models/modules.py:53-68)."""

    def __init__(self, input_dim1, input_dim2, hidden_dim, output_dim):
        super().__init__()
        self.fc1 = nn.Linear(input_dim1 + input_dim2, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)
        self.act = nn.ReLU()

    def forward(self, input_1, input_2):
        return self.fc2(self.act(self.fc1(torch.cat([input_1, input_2], dim=1))))


# --------------------------------------------------------------------------------------------------- backbone
class OracleLSTEP(nn.Module):
    """Same parameters (names, shapes, dtypes) and same method surface as reference ``models.LSTEP.LSTEP``."""

    def __init__(self, node_raw_features, edge_raw_features, neighbor_sampler, full_neighbor_sampler=None, pe_dim=172,
                 num_neighbors=20, time_feat_dim=100, num_fft_batches=100, device="cpu", weighted_sum=False, use_dropout=False, dropout=0.1):
        super().__init__()
        self.weighted_sum = weighted_sum      # the `weighted_sum` ablation (models/LSTEP.py:74,190-206; train_LSTEP_link_prediction.py:126)
        self.use_dropout, self.dropout = use_dropout, dropout      # models/LSTEP.py:40-41 (no reference driver sets use_dropout)
        self.fft_dropout = nn.Dropout(p=dropout)                   # models/LSTEP.py:55
        f_edge = edge_raw_features.shape[-1]
        f_node = node_raw_features.shape[-1]
        self.num_fft_batches = num_fft_batches
        self.pe_dim = pe_dim
        self.device = device
        self.node_raw_features = torch.from_numpy(np.asarray(node_raw_features, dtype=np.float32)).to(device)
        self.edge_raw_features = torch.from_numpy(np.asarray(edge_raw_features, dtype=np.float32)).to(device)
        self.neighbor_sampler = neighbor_sampler
        self.full_neighbor_sampler = full_neighbor_sampler
        self.time_encoder = OracleTimeEncoder(time_feat_dim)
        c = f_edge + time_feat_dim
        self.fft_filter = nn.Linear(pe_dim, num_fft_batches, bias=False).to(torch.complex64)  # weight [T, P] complex
        self.fft_agg = nn.Linear(num_fft_batches, 1, bias=False)
        self.edge_mlp_1 = nn.Linear(c, c)
        self.edge_agg = nn.Linear(num_neighbors, 1)
        self.edge_mlp_2 = nn.Linear(c, c)
        self.node_mlp = nn.Linear(c + f_node, f_node)
        self.self_update_pe = nn.Linear(pe_dim, pe_dim)
        self.pe_mlp_1 = nn.Linear(pe_dim + time_feat_dim, pe_dim)
        self.pe_mlp_2 = nn.Linear(pe_dim, pe_dim)
        self.self_update_neighbor_pe = nn.Linear(pe_dim, pe_dim)
        self.pe_neighbor_mlp_1 = nn.Linear(pe_dim + time_feat_dim, pe_dim)
        self.pe_neighbor_mlp_2 = nn.Linear(pe_dim, pe_dim)
        self.out_node_emb = nn.Linear(pe_dim + f_node, f_node)

    def set_neighbor_sampler(self, neighbor_sampler):
        self.neighbor_sampler = neighbor_sampler

    # ---- helpers
    def _idx(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def _masked_time_features(self, dt64: torch.Tensor, nbr_ids: np.ndarray) -> torch.Tensor:
        """Time-encode float64 deltas after the float32 cast and zero the padded slots (LSTEP.py:153-154, 228-231)."""
        tf = self.time_encoder(dt64.float().to(self.device))
        tf[self._idx(nbr_ids == 0)] = 0.0
        return tf

    # ---- F: FFT filter over the PE history (LSTEP.py:104-137)
    def fourier_transform_pe(self, node_ids, pe, batch_idx, use_dropout=False, use_mixer=False):
        x = pe[self._idx(node_ids)]  # [U, t, P]
        mask = None
        t_len = x.shape[1]
        if t_len < self.num_fft_batches:
            x = torch.cat([x, x.new_zeros(x.shape[0], self.num_fft_batches - t_len, x.shape[2])], dim=1)
            mask = torch.zeros_like(x)
            mask[:, :batch_idx, :] = 1.0  # keyed on batch_idx, not on the stored length (:113)
        _unused = torch.clone(x)  # the reference keeps an (unused) copy here (:115); kept so the CPU baseline pays for it too
        z = torch.fft.fftn(x.to(torch.complex64), dim=1)                               # complex64, whatever the parameters' width (:116)
        if mask is not None:
            z = z * mask
        z = self.fft_filter.weight.unsqueeze(0) * z
        if mask is not None:
            z = z * mask
        z = torch.fft.ifftn(z, dim=1)
        if mask is not None:
            z = z * mask
        y = z.real.to(torch.float32)  # float32; imaginary part dropped (:129)
        if use_dropout:               # (:131-133: dropout on the filtered window, the (padded) window itself added back)
            y = self.fft_dropout(y) + x
        # (the widening is a no-op in the fp32 model; ``float64_yardstick`` keeps the reference's explicit complex64 / float32 casts)
        return self.fft_agg(y.permute(0, 2, 1).to(self.fft_agg.weight.dtype)).squeeze()

    # ---- A + N: edge/time channel and node channel (LSTEP.py:139-220)
    def aggregated_node_embeddings(self, node_ids, node_interact_times, num_neighbors=20, time_gap=2000, testing=False):
        nbr, eid, nts = self.neighbor_sampler.get_historical_neighbors(node_ids, node_interact_times, num_neighbors)
        edge_rows = self.edge_raw_features[self._idx(eid)]                                        # [B, K, F]
        dt = torch.from_numpy(node_interact_times[:, None] - nts)                                 # float64
        tf = self._masked_time_features(dt, nbr)                                                  # [B, K, D]
        x = torch.cat([tf, edge_rows], dim=-1)                                                    # time first (:158)
        _unused = torch.clone(x)                                                                  # dead copy of the reference (:159), baseline cost
        x = self.edge_mlp_1(x)
        x = self.edge_agg(x.permute(0, 2, 1)).squeeze()                                           # Linear over the K axis
        x = self.edge_mlp_2(torch.relu(x))
        if self.use_dropout:          # (:171-172: the FUNCTIONAL dropout, i.e. active whatever the module's mode)
            x = torch.nn.functional.dropout(x, p=self.dropout)

        nbr_g, _, nts_g = self.neighbor_sampler.get_historical_neighbors(node_ids, node_interact_times, time_gap)
        node_rows = self.node_raw_features[self._idx(nbr_g)]                                      # [B, G, F] dense
        m = torch.from_numpy((nbr_g > 0).astype(np.float32))
        m[m == 0] = -1e10
        scores = torch.softmax(m, dim=1).to(self.device)                                          # 1/valid on valid slots
        if self.weighted_sum:
            # (:190-206) per row, one weight per DISTINCT neighbour time u: exp(-(t - mean of the row's slots with time u)), 0 for the
            # padding time 0.0, normalised over the row's distinct times, handed back to the slots and clamped to [0, 1]
            tg = torch.from_numpy(nts_g)                                                          # float32 [B, G]
            uniq, inv = torch.unique(tg, return_inverse=True)
            off = inv + (torch.arange(inv.shape[0]) * uniq.shape[0]).unsqueeze(-1)
            but = scatter_mean_restated(tg.flatten(), off.flatten(), inv.shape[0] * uniq.shape[0]).view(inv.shape[0], uniq.shape[0])
            w = torch.exp(-(torch.from_numpy(node_interact_times).unsqueeze(-1) - but)) * (but != 0.0)     # float64
            sw = torch.sum(w, dim=-1)
            sw = sw + (sw == 0)
            w = w / sw.unsqueeze(-1)
            w = w.flatten()[off.flatten()].view(tg.shape).clamp(0, 1).to(torch.float32)
            pooled = torch.mean(node_rows * scores.unsqueeze(-1) * w.to(self.device).unsqueeze(-1), dim=1)
        else:
            pooled = torch.mean(node_rows * scores.unsqueeze(-1), dim=1)                          # divides by G again (:208)
        node_part = pooled + self.node_raw_features[self._idx(node_ids)]
        return self.node_mlp(torch.cat([node_part, x], dim=-1))

    # ---- C: neighbourhood PE (LSTEP.py:222-249)
    def compute_neighborhood_pe(self, pe, node_ids, node_interact_times, num_neighbors=30):
        nbr, _, nts = self.neighbor_sampler.get_historical_neighbors(node_ids, node_interact_times, num_neighbors)
        dt = torch.from_numpy(node_interact_times).unsqueeze(-1) - torch.from_numpy(nts)          # float64
        tf = self._masked_time_features(dt, nbr)
        own = pe[self._idx(node_ids)]
        agg = torch.cat([pe[self._idx(nbr)], tf], dim=-1).sum(dim=1)                              # PE first (:238)
        agg = self.pe_neighbor_mlp_2(torch.relu(self.pe_neighbor_mlp_1(agg)))
        return own + torch.tanh(self.self_update_neighbor_pe(own) + agg)

    # ---- O (LSTEP.py:251-266)
    def combining_pe_raw_feat(self, pe, node_ids, node_interact_times, num_neighbors=30, time_gap=2000, testing=False):
        h = self.aggregated_node_embeddings(node_ids, node_interact_times, num_neighbors, time_gap)
        q = self.compute_neighborhood_pe(pe, node_ids, node_interact_times, num_neighbors)
        return self.out_node_emb(torch.cat([h, q], dim=-1))

    # ---- U1 + U2 (LSTEP.py:268-340); mutates `pe` in place and returns it
    def update_pe(self, pe, node_ids, edge_ids, batch_src_node_ids, batch_dst_node_ids, node_interact_times, current_time,
                  num_neighbors=30, time_gap=2000):
        ids = self._idx(node_ids)
        src = self._idx(batch_src_node_ids)
        dst = self._idx(batch_dst_node_ids)
        now32 = torch.tensor([current_time], dtype=torch.float32)  # float32-rounded FIRST (:277)
        own = pe[ids]
        dt = (now32 - torch.from_numpy(node_interact_times)).float().to(self.device)              # f32 - f64 -> f64 -> f32
        tf = self.time_encoder(dt)                                                                # [B, D]
        acc = pe.new_zeros(pe.shape[0], pe.shape[1] + tf.shape[1])                                # dense [N+1, P+D]
        acc.index_add_(0, src, torch.cat([pe[dst], tf], dim=-1))
        acc.index_add_(0, dst, torch.cat([pe[src], tf], dim=-1))
        msg = self.pe_mlp_2(torch.relu(self.pe_mlp_1(acc[ids])))
        pe[ids] = own + torch.tanh(self.self_update_pe(own) + msg)

        # phase 2: push to the K most recent neighbours; `node_ids` (U rows) is zipped with the B edge times (:306-308)
        nbr, _, nts = self.neighbor_sampler.get_historical_neighbors(node_ids, node_interact_times, num_neighbors)
        rep = ids.unsqueeze(-1).expand(nbr.shape).reshape(-1)
        nbr_flat = nbr.reshape(-1)
        dt2 = (now32 - torch.from_numpy(nts.reshape(-1))).float().to(self.device)                 # f32 - f32 (:314)
        tf2 = self.time_encoder(dt2)
        tf2[self._idx(nbr_flat == 0)] = 0.0
        pe[0] = 0.0                                                                               # (:317)
        acc2 = pe.new_zeros(pe.shape[0], pe.shape[1] + tf2.shape[1])
        acc2.index_add_(0, self._idx(nbr_flat), torch.cat([pe[rep], tf2], dim=-1))
        touched = torch.unique(self._idx(nbr_flat))                                               # sorted; may contain 0
        own2 = pe[touched]
        msg2 = self.pe_mlp_2(torch.relu(self.pe_mlp_1(acc2[touched])))
        pe[touched] = own2 + torch.tanh(msg2)                                                     # self_update_pe term is dead (:334-335)
        return pe


def scatter_mean_restated(src: torch.Tensor, index: torch.Tensor, size: int) -> torch.Tensor:
    """``torch_scatter.scatter_mean(src, index, out=zeros(size))`` (third-party wheel, absent here; version unpinned by the reference,
    which has no requirements file): its documented definition -- out[i] = (sum of the src entries with index i) / max(their count, 1),
    summed in src's dtype in memory order.  Only call site: models/LSTEP.py:194."""
    out = torch.zeros(size, dtype=src.dtype).scatter_add_(0, index, src)
    cnt = torch.zeros(size, dtype=src.dtype).scatter_add_(0, index, torch.ones_like(src)).clamp_(min=1)
    return out / cnt


def build_oracle_model(node_raw, edge_raw, sampler, num_neighbors, num_fft_batches, state_dict=None, feat_dim=172,
                       time_dim=100, pe_dim=172, weighted_sum=False, use_dropout=False, dropout=0.1):
    """``nn.Sequential(backbone, link_predictor)`` as the reference wraps it (train_LSTEP_link_prediction.py:140-142)."""
    bb = OracleLSTEP(node_raw, edge_raw, sampler, sampler, pe_dim=pe_dim, num_neighbors=num_neighbors,
                     time_feat_dim=time_dim, num_fft_batches=num_fft_batches, weighted_sum=weighted_sum, use_dropout=use_dropout, dropout=dropout)
    pred = OracleMergeLayer(feat_dim, feat_dim, feat_dim, 1)
    model = nn.Sequential(bb, pred)
    if state_dict is not None:
        model.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)
    return model


def float64_yardstick(model: nn.Sequential, tables: bool = True) -> nn.Sequential:
    """The same model evaluated in float64 / complex128 (a deep copy; feature tables and parameters widened, the reference's float32
    roundings of time differences kept).  NOT a second oracle: a yardstick for the places where thousands of fp32 terms are summed --
    there the reference's own result depends on its summation order (sequential ``index_add_`` on the CPU, atomics on a GPU), and a test
    may accept a HIP value that differs from the fp32 oracle by more than the bar only if it is no further from this float64 value
    than the fp32 oracle itself is (plus the bar).  ``tables=False``: without the feature tables (enough for ``update_pe`` and
    ``fourier_transform_pe``, which never read them)."""
    import copy
    src = model[0]
    node_raw, edge_raw = src.node_raw_features, src.edge_raw_features
    src.node_raw_features = src.edge_raw_features = None     # (not copied twice: a 20 M-edge table is 13.8 GB)
    try:
        m = copy.deepcopy(model).double()      # .double() leaves complex parameters alone
    finally:
        src.node_raw_features, src.edge_raw_features = node_raw, edge_raw
    bb = m[0]
    bb.fft_filter.weight = nn.Parameter(bb.fft_filter.weight.detach().to(torch.complex128))
    if tables:
        bb.node_raw_features, bb.edge_raw_features = node_raw.double(), edge_raw.double()
    return m
