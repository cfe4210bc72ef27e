"""Device-resident temporal neighbour sampler: the host-side mirror of reference ``utils.utils.NeighborSampler``.

Same public surface as the reference (``utils/utils.py:70-279``, built by ``get_neighbor_sampler`` at ``:282-301``):
``sample_neighbor_strategy``, ``seed``, ``reset_random_state()``, ``get_historical_neighbors(node_ids,
node_interact_times, num_neighbors) -> (int64[M, K], int64[M, K], float32[M, K])`` with numpy in / numpy out.
What changes is where the data lives: the per-node Python lists become one time-sorted CSR in HBM
(``int64 indptr``, ``int32 nbr``, ``int32 eid``, ``float64 ts`` -- 16 bytes per adjacency entry) and the per-row
Python loop + ``np.searchsorted`` becomes ``lstep_sample_recent`` (one wave per row).

Only the ``'recent'`` strategy runs on the device (and only it can feed the fused LSTEP kernels); ``'uniform'`` /
``'time_interval_aware'`` are defined by ``np.random.RandomState.choice`` call order (``utils/utils.py:175-198``) and are
replayed on the host for API parity.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native as nat


def build_csr_arrays(src, dst, eid, ts, num_nodes=None):
    """Host-side CSR build with the reference's ordering: each edge goes to both endpoints (src's list first,
    ``utils/utils.py:297-299``); each list is stably sorted by timestamp (``:99``)."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    eid = np.asarray(eid, dtype=np.int64)
    ts = np.asarray(ts, dtype=np.float64)
    e = len(src)
    top = int(max(src.max(), dst.max())) if e else 0
    if e and (min(src.min(), dst.min()) < 0):
        raise ValueError("negative node id")
    rows = max(top, int(num_nodes) if num_nodes is not None else 0) + 1
    if rows >= 2 ** 31 or (e and eid.max() >= 2 ** 31):
        raise ValueError("node / edge ids must fit int32")
    owner = np.empty(2 * e, dtype=np.int64)
    owner[0::2], owner[1::2] = src, dst
    other = np.empty(2 * e, dtype=np.int32)
    other[0::2], other[1::2] = dst, src
    tss = np.repeat(ts, 2)
    # stable sort by (owner, time); ties keep insertion order = the reference's `sorted(..., key=ts)` on append order
    order = np.lexsort((tss, owner))  # lexsort is stable: equal (owner, ts) keep their original (insertion) order
    indptr = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(owner, minlength=rows), out=indptr[1:])
    return indptr, other[order], np.repeat(eid, 2).astype(np.int32)[order], tss[order], rows


class NeighborSampler:
    def __init__(self, src_node_ids, dst_node_ids, edge_ids, node_interact_times, num_nodes=None,
                 sample_neighbor_strategy: str = "recent", time_scaling_factor: float = 0.0, seed: int = None,
                 device="cuda"):
        if sample_neighbor_strategy not in ("recent", "uniform", "time_interval_aware"):
            raise ValueError(f"Not implemented error for sample_neighbor_strategy {sample_neighbor_strategy}!")
        nat.load_library()  # fail loudly if the HIP library is missing
        self.sample_neighbor_strategy = sample_neighbor_strategy
        self.time_scaling_factor = time_scaling_factor
        self.seed = seed
        self.device = torch.device(device)
        indptr, nbr, eid, ts, rows = build_csr_arrays(src_node_ids, dst_node_ids, edge_ids, node_interact_times, num_nodes)
        self.num_rows = rows
        self.nnz = int(len(nbr))
        self.indptr = torch.from_numpy(indptr).to(self.device)
        self.nbr = torch.from_numpy(nbr).to(self.device)
        self.eid = torch.from_numpy(eid).to(self.device)
        self.ts = torch.from_numpy(ts).to(self.device)
        if sample_neighbor_strategy != "recent":
            # uniform / time_interval_aware are DEFINED by numpy's RandomState.choice call order (utils/utils.py:175-198):
            # they replay on the host from a host copy of the CSR; the device path serves 'recent'.
            self._host = (indptr, nbr.astype(np.int64), eid.astype(np.int64), ts)
            if seed is not None:
                self.random_state = np.random.RandomState(seed)
        if self.nnz == 0:  # keep valid device pointers for empty graphs
            self.nbr = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.eid = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.ts = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.max_degree = int(np.diff(indptr).max()) if len(indptr) > 1 else 0
        self._csr = nat.CsrStruct(self.indptr.data_ptr(), self.nbr.data_ptr(), self.eid.data_ptr(), self.ts.data_ptr(),
                                  self.num_rows, self.nnz, self.max_degree)

    @classmethod
    def from_device_edges(cls, src: torch.Tensor, dst: torch.Tensor, eid: torch.Tensor, ts: torch.Tensor, num_nodes: int,
                          seed: int = None, sample_neighbor_strategy: str = "recent", time_scaling_factor: float = 0.0):
        """Build the CSR on the GPU from device-resident edge arrays (large graphs: the reference's Python loop over
        all edges, ``utils/utils.py:296-299``, takes minutes at 10^7-10^8 edges).  Same ordering rule as
        :func:`build_csr_arrays`: two stable sorts, by time then by owner, keep insertion order among ties."""
        nat.load_library()
        self = cls.__new__(cls)
        if sample_neighbor_strategy not in ("recent", "uniform", "time_interval_aware"):
            raise ValueError(f"Not implemented error for sample_neighbor_strategy {sample_neighbor_strategy}!")
        self.sample_neighbor_strategy, self.time_scaling_factor, self.seed = sample_neighbor_strategy, time_scaling_factor, seed
        dev = src.device
        self.device = dev
        e = src.numel()
        rows = int(num_nodes) + 1
        owner = torch.stack([src, dst], dim=1).reshape(-1)
        other = torch.stack([dst, src], dim=1).reshape(-1).to(torch.int32)
        tss = ts.to(torch.float64).repeat_interleave(2)
        if e and bool((tss[1:] < tss[:-1]).any()):
            o1 = torch.sort(tss, stable=True).indices
            owner, other, tss = owner[o1], other[o1], tss[o1]
            eids = eid.repeat_interleave(2)[o1]
        else:  # already chronological (the reference's data files are): one stable sort by owner is enough
            eids = eid.repeat_interleave(2)
        o2 = torch.sort(owner, stable=True).indices
        counts = torch.bincount(owner, minlength=rows)
        self.indptr = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
        self.indptr[1:] = torch.cumsum(counts, 0)
        self.nbr = other[o2].contiguous()
        self.eid = eids[o2].to(torch.int32).contiguous()
        self.ts = tss[o2].contiguous()
        self.num_rows, self.nnz = rows, int(2 * e)
        self.max_degree = int(counts.max().item()) if e else 0          # (set-up time: the one host read of the build)
        self._csr = nat.CsrStruct(self.indptr.data_ptr(), self.nbr.data_ptr(), self.eid.data_ptr(), self.ts.data_ptr(),
                                  self.num_rows, self.nnz, self.max_degree)
        if sample_neighbor_strategy != "recent":       # the RNG-defined strategies replay numpy's generator on the host, from a host copy of the CSR
            self._host = (self.indptr.cpu().numpy(), self.nbr.cpu().numpy().astype(np.int64), self.eid.cpu().numpy().astype(np.int64),
                          self.ts.cpu().numpy())
            if seed is not None:
                self.random_state = np.random.RandomState(seed)
        return self

    @property
    def csr(self):
        return self._csr

    def reset_random_state(self):
        """API parity (``utils/utils.py:274-279``); the 'recent' strategy draws no random numbers."""
        if self.seed is not None:
            self.random_state = np.random.RandomState(self.seed)

    # ---- device-side entry point: tensors in, tensors out, no host sync
    def sample_device(self, node_ids: torch.Tensor, times: torch.Tensor, num_neighbors: int, want_count: bool = False):
        assert num_neighbors > 0, "Number of sampled neighbors for each node should be greater than 0!"
        node_ids = node_ids.to(device=self.device, dtype=torch.int64).contiguous()
        times = times.to(device=self.device, dtype=torch.float64).contiguous()
        m = node_ids.numel()
        nbr = torch.empty((m, num_neighbors), dtype=torch.int64, device=self.device)
        eid = torch.empty((m, num_neighbors), dtype=torch.int64, device=self.device)
        nt = torch.empty((m, num_neighbors), dtype=torch.float32, device=self.device)
        cnt = torch.empty((m,), dtype=torch.int32, device=self.device) if want_count else None
        lib = nat.load_library()
        with torch.cuda.device(self.device):
            nat.check(lib.lstep_sample_recent(self._csr, nat.ptr(node_ids), m, nat.ptr(times), times.numel(), int(num_neighbors),
                                              nat.ptr(nbr), nat.ptr(eid), nat.ptr(nt), nat.ptr(cnt), nat.current_stream()))
        return (nbr, eid, nt, cnt) if want_count else (nbr, eid, nt)

    def _sampled_probabilities(self, times: np.ndarray) -> np.ndarray:
        """CAWN-style time-interval-aware weights of one node's full history (utils/utils.py:111-127)."""
        if len(times) == 0:
            return np.array([])
        rel = times - np.max(times)
        w = np.exp(self.time_scaling_factor * rel)
        p = w / np.cumsum(w)
        p[np.isnan(p)] = -1e10
        return p

    def _row_probabilities(self, node: int, cnt: int) -> np.ndarray:
        """``torch.softmax(float32(weights of the node's FULL history)[:cnt])`` (utils/utils.py:180-182), computed exactly as the reference
        computes it (one 1-D torch.softmax per row) but kept: the weights depend on the node only, the softmax on (node, cnt), and the
        model asks for the same rows three times per call (edge, node and PE channel).  Both caches are bounded (the CSR never changes)."""
        if self.__dict__.get("_p_tsf") != self.time_scaling_factor:        # (a public attribute: a change invalidates what was kept)
            self._p_soft, self._p_node, self._p_soft_bytes, self._p_node_bytes = {}, {}, 0, 0
            self._p_tsf = self.time_scaling_factor
        soft = self._p_soft
        p = soft.get((node, cnt))
        if p is not None:
            return p
        indptr, _, _, tss = self._host
        weights = self._p_node
        w = weights.get(node)
        if w is None:
            w = self._sampled_probabilities(tss[indptr[node]:indptr[node + 1]])
            if self._p_node_bytes + w.nbytes > (512 << 20):
                weights.clear()
                self._p_node_bytes = 0
            weights[node] = w
            self._p_node_bytes = self._p_node_bytes + w.nbytes
        p = torch.softmax(torch.from_numpy(w[:cnt]).float(), dim=0).numpy()
        if self._p_soft_bytes + p.nbytes > (256 << 20):
            soft.clear()
            self._p_soft_bytes = 0
        soft[(node, cnt)] = p
        self._p_soft_bytes = self._p_soft_bytes + p.nbytes
        return p

    def _sample_random_host(self, node_ids, node_interact_times, num_neighbors, out=None):
        """'uniform' / 'time_interval_aware' (utils/utils.py:175-198): per row, in row order, one RandomState.choice over the interactions
        strictly before the query time, then a re-sort of the sampled slots by (float32) time.  The draws are made by the native replay of
        numpy's legacy generator (``lstep_sample_random_host``: the row loop, MT19937 and both paths of ``RandomState.choice`` in C++, on
        numpy's OWN generator state, which is read before and stored back after the call -- Python-side and native draws share one
        stream).  The re-sort of every sampled row by its float32 times (utils/utils.py:192-196) is an UNSTABLE numpy ``argsort`` in the
        reference, whose order among equal times is an implementation detail of numpy (introsort or a SIMD sort, by version, length and
        CPU).  The node's history is time-sorted, so that order is the order of the drawn POSITIONS wherever no two distinct positions share
        a float32 time: ``lstep_sample_random_sorted_host`` (round 5) draws on one thread and then sorts the positions and gathers the
        triples on ``LSTEP_HOST_THREADS`` workers (default: the cores there are, at most 16), flags the rows that do hold such a tie and
        leaves those in draw order; only they go through numpy's 1-D ``argsort`` here (the same call the reference makes).
        LSTEP_RNG_NUMPY_SORT=1: round 4's path (native draws, every row re-sorted by numpy; the A/B of the native sort).
        LSTEP_PY_RNG_SAMPLER=1: the interpreter loop of rounds 1-3 (A/B).
        ``out``: three C-contiguous arrays [rows, K] (int64, int64, float32) to fill instead of fresh ones (the model hands in slices of
        pinned staging buffers: no concatenate, no pageable copy); rows without history are zeroed here."""
        import os
        if os.environ.get("LSTEP_PY_RNG_SAMPLER") == "1":
            res = self._sample_random_python(node_ids, node_interact_times, num_neighbors)
            if out is not None:
                for dst, src in zip(out, res):
                    dst[...] = src
                return out
            return res
        import ctypes
        indptr, nbrs, eids, tss = self._host
        rows = len(node_ids)
        K = int(num_neighbors)
        m = min(rows, len(node_interact_times))          # (the reference zips the two arrays: utils/utils.py:169)
        if out is None:
            out_n = np.zeros((rows, K), dtype=np.longlong)
            out_e = np.zeros((rows, K), dtype=np.longlong)
            out_t = np.zeros((rows, K), dtype=np.float32)
        else:
            out_n, out_e, out_t = out
            for a, dt in ((out_n, np.int64), (out_e, np.int64), (out_t, np.float32)):
                if a.shape != (rows, K) or a.dtype != dt or not a.flags.c_contiguous:
                    raise ValueError("out: three C-contiguous [rows, num_neighbors] arrays (int64, int64, float32)")
            if m < rows:
                out_n[m:], out_e[m:], out_t[m:] = 0, 0, 0
        if m == 0:
            return out_n, out_e, out_t
        ids = np.ascontiguousarray(np.asarray(node_ids)[:m], dtype=np.int64)
        ts = np.ascontiguousarray(np.asarray(node_interact_times)[:m], dtype=np.float64)
        lib = nat.load_library()
        vp = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
        p_vals = p_off = None
        cnt = np.empty(m, dtype=np.int64)
        nat.check(lib.lstep_count_before_host(vp(indptr), vp(tss), self.num_rows, vp(ids), vp(ts), m, vp(cnt)))
        if out is not None:
            empty = np.nonzero(cnt == 0)[0]               # (a reused buffer: the rows the native code leaves untouched)
            if empty.size:
                out_n[empty], out_e[empty], out_t[empty] = 0, 0, 0
        if self.sample_neighbor_strategy == "time_interval_aware":
            # the probabilities are torch.softmax's float32 output (utils/utils.py:182), row by row as the reference computes them: their
            # last bits depend on torch's vectorised exp, which only torch reproduces; everything behind them is replayed natively
            parts = [self._row_probabilities(int(ids[r]), int(cnt[r])) for r in np.nonzero(cnt)[0]]
            p_vals = np.ascontiguousarray(np.concatenate(parts) if parts else np.zeros(0), dtype=np.float32)
            p_off = np.zeros(m + 1, dtype=np.int64)
            np.cumsum(cnt, out=p_off[1:])
            if p_vals.size and (np.isnan(p_vals).any() or (p_vals < 0).any()):
                raise ValueError("probabilities contain NaN" if np.isnan(p_vals).any() else "probabilities are not non-negative")
        owner = np.random if self.seed is None else self.random_state
        kind, key, pos, has_gauss, cached = owner.get_state()
        key = np.ascontiguousarray(key, dtype=np.uint32).copy()
        cpos = ctypes.c_int32(int(pos))
        args = (vp(indptr), vp(nbrs), vp(eids), vp(tss), self.num_rows, vp(ids), vp(ts), m, K, None if p_vals is None else vp(p_vals),
                None if p_off is None else vp(p_off), vp(key), ctypes.byref(cpos), vp(out_n), vp(out_e), vp(out_t))
        if os.environ.get("LSTEP_RNG_NUMPY_SORT") == "1":
            nat.check(lib.lstep_sample_random_host(*args))
            resort = np.nonzero(cnt)[0]                   # (rows without history stay all zeros)
        else:
            tied = np.zeros(m, dtype=np.uint8)
            threads = int(os.environ.get("LSTEP_HOST_THREADS", "0")) or min(16, os.cpu_count() or 1)
            nat.check(lib.lstep_sample_random_sorted_host(*args, vp(tied), threads))
            resort = np.nonzero(tied)[0]
        owner.set_state((kind, key, int(cpos.value), has_gauss, cached))
        self.last_numpy_sorted_rows = int(resort.size)
        for r in resort:
            order = out_t[r].argsort()
            out_n[r], out_e[r], out_t[r] = out_n[r][order], out_e[r][order], out_t[r][order]
        return out_n, out_e, out_t

    def sample_random_into(self, node_ids, node_interact_times, num_neighbors, out):
        """``get_historical_neighbors`` of an RNG-defined strategy, written into the caller's arrays (same checks, same draws)."""
        assert num_neighbors > 0, "Number of sampled neighbors for each node should be greater than 0!"
        if self.sample_neighbor_strategy == "recent":
            raise ValueError("sample_random_into serves the RNG-defined strategies ('uniform', 'time_interval_aware')")
        ids = np.asarray(node_ids)
        if ids.size and (ids.min() < 0 or ids.max() >= self.num_rows):
            raise IndexError("list index out of range")
        return self._sample_random_host(ids, np.asarray(node_interact_times), num_neighbors, out=out)

    def _sample_random_python(self, node_ids, node_interact_times, num_neighbors):
        """The same as an interpreter loop around ``RandomState.choice`` itself (rounds 1-3; kept as the A/B reference of the native replay)."""
        indptr, nbrs, eids, tss = self._host
        rows = len(node_ids)
        out_n = np.zeros((rows, num_neighbors), dtype=np.longlong)
        out_e = np.zeros((rows, num_neighbors), dtype=np.longlong)
        out_t = np.zeros((rows, num_neighbors), dtype=np.float32)
        rng = np.random if self.seed is None else self.random_state
        for r, (node, t) in enumerate(zip(node_ids, node_interact_times)):
            lo, hi = indptr[node], indptr[node + 1]
            cnt = int(np.searchsorted(tss[lo:hi], t))
            if cnt == 0:
                continue
            p = None
            if self.sample_neighbor_strategy == "time_interval_aware":
                p = torch.softmax(torch.from_numpy(self._sampled_probabilities(tss[lo:hi])[:cnt]).float(), dim=0).numpy()
            pick = rng.choice(a=cnt, size=num_neighbors, p=p)
            out_n[r, :], out_e[r, :], out_t[r, :] = nbrs[lo + pick], eids[lo + pick], tss[lo + pick]
            order = out_t[r, :].argsort()
            out_n[r, :], out_e[r, :], out_t[r, :] = out_n[r, :][order], out_e[r, :][order], out_t[r, :][order]
        return out_n, out_e, out_t

    # ---- reference-shaped entry point: numpy in, numpy out (utils/utils.py:148-213)
    def get_historical_neighbors(self, node_ids: np.ndarray, node_interact_times: np.ndarray, num_neighbors: int = 20):
        assert num_neighbors > 0, "Number of sampled neighbors for each node should be greater than 0!"
        if self.sample_neighbor_strategy != "recent":
            ids = np.asarray(node_ids)
            if ids.size and (ids.min() < 0 or ids.max() >= self.num_rows):
                raise IndexError("list index out of range")
            return self._sample_random_host(ids, np.asarray(node_interact_times), num_neighbors)
        ids = np.ascontiguousarray(node_ids, dtype=np.int64)
        if ids.size and (ids.min() < 0 or ids.max() >= self.num_rows):
            raise IndexError("list index out of range")  # what the reference's list lookup raises
        ts = np.ascontiguousarray(node_interact_times, dtype=np.float64)
        nbr, eid, nt = self.sample_device(torch.from_numpy(ids), torch.from_numpy(ts), num_neighbors)
        return nbr.cpu().numpy().astype(np.longlong, copy=False), eid.cpu().numpy().astype(np.longlong, copy=False), nt.cpu().numpy()


def get_neighbor_sampler(data, sample_neighbor_strategy: str = "uniform", time_scaling_factor: float = 0.0, seed: int = None,
                         device="cuda", num_nodes=None):
    """Mirror of reference ``utils.utils.get_neighbor_sampler(data, ...)``: ``data`` needs ``src_node_ids``,
    ``dst_node_ids``, ``edge_ids``, ``node_interact_times`` (``utils/DataLoader.py:68-86``)."""
    return NeighborSampler(data.src_node_ids, data.dst_node_ids, data.edge_ids, data.node_interact_times, num_nodes=num_nodes,
                           sample_neighbor_strategy=sample_neighbor_strategy, time_scaling_factor=time_scaling_factor, seed=seed,
                           device=device)
