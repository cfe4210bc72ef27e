"""Multi-GPU execution of the L-STEP per-batch protocol: one process per GPU, ``torch.distributed`` (RCCL over xGMI).

The reference is single-process (no collective anywhere); this module is new design.  Within a batch the
path is independent per destination row, across batches it is strictly sequential (the PE state evolves), so the W ranks
cooperate on ONE global batch of ``W * B`` edges at a time (weak scaling: B per GPU fixed):

  state       node n is OWNED by rank ``n % W``.  The PE history ring -- the only O(N * T) state (69 GB at 1 M nodes,
              275 GB at 4 M) -- is sharded by owner: rank r keeps ``[T+2, ceil((N+1-r)/W), P]``.  The current PE table
              ``[N+1, P]`` (688 MB per 1 M nodes), CSR, feature tables and weights are replicated.
  FFT splice  rank r filters the history of the batch nodes it owns; the ``[U, P]`` filtered rows are ALL-GATHERED
              (padded, <= 22 MB per 32 K nodes) and written into every replica of the current table.
  combine     rank r runs the fused gather + dense tails for ITS B edges (3B rows); the loss is the mean over the
              global batch, i.e. the mean of the rank means.
  backward    gradients w.r.t. the spliced rows are REDUCE-SCATTERED by owner, each rank back-propagates its owned rows through its
              history shard into the filter coefficients; parameter gradients are ALL-REDUCED (one flat bucket, 2.3 MB).
  update_pe   three forms (``DistributedLstep.form``, LSTEP_PHASE2):
              "pull"      OWNER-SHARDED PE TABLE.  A rank is authoritative for the rows it owns only; both phases are computed by the
                          owner of the UPDATED row (phase 1: batch nodes, their new rows ALL-GATHERED -- "updated positional encodings
                          at snapshot boundaries", BASELINE.json north_star -- because phase 2's messages carry them; phase 2: touched
                          neighbours, nothing sent).  Before its next gather a rank PULLS the rows that gather will read and it does
                          not own (its 3 B rows, their K most recent neighbours, row 0: ~23 % of a 4 M-node table) from their owners:
                          an all-to-all-v whose requests are computed one step ahead from the look-ahead batch, so the rows travel
                          underneath the backward pass on a communicator of their own (``RowPull``).  Per-rank bytes and update
                          FLOPs are flat in W: the form for W = 8.  The full-size table of a rank is a CACHE outside its owned rows.
              "replicate" every rank recomputes update_pe for the whole global batch on a replicated table: no update collective, but
                          work that grows with W (default up to W = 4).
              "allgather" owner-computes with an all-gather of ALL updated rows into replicated tables (round 1's form; kept for A/B).

All collectives are small-to-medium one-shot gathers/reductions (no ring-pipelined bulk transfer is needed); the data
path itself (gathers, GEMMs) has no collective inside.  Results equal the single-GPU engine on the same global batch up
to fp32 summation order (``tests/test_parallel.py``).
"""
from __future__ import annotations

import contextlib
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _native as nat
from .engine import HistoryRing, LstepEngine, _LinkLoss, _backward_unit, _lookup_rows
from .model import SplicedRows


quiesce_collectives = nat.quiesce_collectives      # (lives beside the stream helpers: the single-GPU engine's capture needs it too)


class LstepCapacityError(RuntimeError):
    """A fixed-capacity block of the device-driven multi-GPU iteration was too small for the data (see ``DistributedLstep.check_capacity``)."""


# ---------------------------------------------------------------------------------------------- collectives
# ONE code path for RCCL and gloo: equal padded blocks in the flat [W * rows, ...] layout both backends accept for
# all_gather_into_tensor / reduce_scatter_tensor, so the world-size-2 gloo tests execute exactly the branches RCCL executes on
# the GPUs.  The only backend difference left: gloo has no device collectives, so CUDA tensors are staged through the host
# (tests with several ranks on one GPU).  LSTEP_FORCE_COLLECTIVES=1 keeps the collectives even at world size 1 (the -m gpu test
# that runs the RCCL calls on a one-GPU box).
def _staged(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _skip_single(w: int) -> bool:
    return w == 1 and os.environ.get("LSTEP_FORCE_COLLECTIVES") != "1"


def _pad_rows(t: torch.Tensor, rows: int) -> torch.Tensor:
    if t.shape[0] == rows and t.is_contiguous():
        return t
    pad = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    return pad


def _unpad_blocks(flat: torch.Tensor, counts, mx: int) -> torch.Tensor:
    """[W * mx, ...] of equal padded blocks -> the counts[r] valid rows of every block, concatenated in rank order."""
    if all(c == mx for c in counts):
        return flat
    return torch.cat([flat[i * mx:i * mx + c] for i, c in enumerate(counts)], dim=0)


def exchange_counts(n: int, device, group=None):
    """Row count of every rank (one small all-gather + a host read: avoid it where the counts can be derived locally)."""
    w = dist.get_world_size(group)
    if _skip_single(w):
        return [int(n)]
    staged = torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"
    mine = torch.tensor([int(n)], dtype=torch.int64, device="cpu" if staged else device)
    ns = torch.empty((w,), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(ns, mine, group=group)
    return [int(x) for x in ns.tolist()]


class PendingGather:
    """All-gather of row blocks with different row counts, left in flight (``async_op=True``: on RCCL's stream, or in gloo's worker
    thread) while the caller keeps enqueueing compute; ``wait()`` makes the current stream wait for it and returns the rows of all
    ranks concatenated in rank order.  ``counts`` (rows per rank) may be passed when every rank can derive it locally: that saves the
    size exchange and its host sync."""

    def __init__(self, t: torch.Tensor, group=None, counts=None, sync: bool = False):
        """``sync``: issue the collective synchronously (the current stream waits for it at once).  Needed inside a graph capture on a SIDE
        stream: an asynchronous collective captured there faults in torch 2.10 / RCCL 2.26 (tools/rccl_capture_probe.py); on the capturing
        stream itself the asynchronous all-gather captures fine."""
        self.done, self.work = None, None
        w = dist.get_world_size(group)
        self.counts = counts if counts is not None else exchange_counts(t.shape[0], t.device, group)
        if _skip_single(w):
            self.done = t
            return
        self.dev, self.mx = t.device, max(self.counts)
        self.staged = _staged(t, group)
        pad = _pad_rows(t, self.mx)
        if self.staged:
            pad = pad.cpu()
        self.pad = pad                                   # (kept alive until the collective has finished)
        self.out = torch.empty((w * self.mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=pad.device)
        if sync:
            dist.all_gather_into_tensor(self.out, pad, group=group)
        else:
            self.work = dist.all_gather_into_tensor(self.out, pad, group=group, async_op=True)

    def wait(self) -> torch.Tensor:
        if self.done is None:
            if self.work is not None:
                self.work.wait()
            out = _unpad_blocks(self.out, self.counts, self.mx)
            self.done = out.to(self.dev) if self.staged else out
            self.pad = self.out = self.work = None
        return self.done


def all_gather_var(t: torch.Tensor, group=None, counts=None):
    """Blocking form of ``PendingGather``.  Returns (concatenated rows in rank order, counts list)."""
    g = PendingGather(t, group, counts, sync=t.is_cuda and torch.cuda.is_current_stream_capturing())
    return g.wait(), g.counts


def all_reduce_sum(t: torch.Tensor, group=None):
    if _skip_single(dist.get_world_size(group)):
        return t
    if _staged(t, group):
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, group=group)
    return t


def reduce_scatter_var(t: torch.Tensor, counts, group=None):
    """Sum ``t`` (rows ordered rank-major, ``counts[r]`` rows for rank r) over all ranks and return THIS rank's block: one
    reduce_scatter of equal padded blocks (each rank receives only what it owns: half the traffic of an all-reduce)."""
    w, r = dist.get_world_size(group), dist.get_rank(group)
    assert len(counts) == w and sum(counts) == t.shape[0]
    if _skip_single(w):
        return t
    mx = max(counts)
    if all(c == mx for c in counts):
        padded = t.contiguous()
    else:
        padded = torch.zeros((w * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        off = 0
        for i, c in enumerate(counts):
            padded[i * mx:i * mx + c] = t[off:off + c]
            off += c
    staged = _staged(t, group)
    if staged:
        padded = padded.cpu()
    out = torch.empty((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=padded.device)
    dist.reduce_scatter_tensor(out, padded, group=group)
    out = out[:counts[r]]
    return out.to(t.device) if staged else out


def all_reduce_gradients(params, group=None, extra: torch.Tensor = None, scale: float = None):
    """One flat bucket for all parameter gradients (complex ones viewed as real); missing grads count as zero.  ``extra`` (1-D float32,
    optional) rides in the same bucket and is returned summed over the ranks (the three loss scalars: one collective less per step).
    ``scale``: the reduced bucket (gradients and ``extra``) is multiplied by it (1 / world: the mean over the ranks) in one launch."""
    views = []
    for p in params:
        if not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        views.append(torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad)
    parts = [v.reshape(-1) for v in views] + ([extra.reshape(-1).to(views[0].dtype)] if extra is not None else [])
    flat = torch.cat(parts)
    all_reduce_sum(flat, group)
    if scale is not None:
        flat.mul_(scale)
    chunks = torch.split(flat, [p.numel() for p in parts])
    torch._foreach_copy_(views, [c.view_as(v) for c, v in zip(chunks, views)])
    return chunks[-1] if extra is not None else None


def pack_ids(z: torch.Tensor, ids: torch.Tensor, width: int) -> torch.Tensor:
    """Hide the int64 row ids in two of z's zero padding columns (bit patterns), so ids and rows travel in ONE collective.
    z is [n, ld] with ld >= width + 2; the residual kernel only reads the first ``width`` columns."""
    assert z.shape[1] >= width + 2 and z.dtype == torch.float32
    z = z.contiguous()
    z[:, width:width + 2] = ids.to(torch.int64).view(torch.int32).view(-1, 2).view(torch.float32)
    return z


def unpack_ids(z: torch.Tensor, width: int) -> torch.Tensor:
    return z[:, width:width + 2].contiguous().view(torch.int32).view(torch.int64).reshape(-1)


def nat_branch(name: str) -> int:
    from . import _native as nat
    return nat.BRANCH_EDGE_NODE if name == "edge_node" else nat.BRANCH_PE


def owned_rows(num_rows: int, world: int, rank: int) -> int:
    """Number of node ids in [0, num_rows) with id % world == rank."""
    return (num_rows - rank + world - 1) // world if num_rows > rank else 0


def exchange_rows(send: torch.Tensor, send_counts, recv_counts, group=None, async_op: bool = False):
    """All-to-all-v of row blocks: ``send`` holds ``send_counts[p]`` rows for rank p, in rank order; returns the rows this rank receives,
    ``recv_counts[p]`` from rank p, in rank order (``async_op``: an object whose ``wait()`` returns them).  RCCL: one
    ``all_to_all_single`` with split sizes; gloo (the CPU / one-GPU rehearsals -- it has no all-to-all): point-to-point sends and
    receives of the same blocks, device tensors staged through the host."""
    w, r = dist.get_world_size(group), dist.get_rank(group)
    assert len(send_counts) == w and len(recv_counts) == w and sum(send_counts) == send.shape[0]
    send = send.contiguous()

    class _Done:
        def __init__(self, t):
            self.t = t

        def wait(self):
            return self.t

    if _skip_single(w):
        return _Done(send) if async_op else send
    out_shape = (sum(recv_counts),) + tuple(send.shape[1:])
    if dist.get_backend(group) != "gloo":
        out = torch.empty(out_shape, dtype=send.dtype, device=send.device)
        if send.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a graph capture: synchronous, and only ever issued from the capturing stream itself (an asynchronous all-to-all, or one
            # issued from a side stream that joined the capture, faults in torch 2.10 / RCCL 2.26: tools/rccl_capture_probe.py)
            assert len(set(send_counts) | set(recv_counts)) == 1, "captured exchanges move equal blocks"
            dist.all_to_all_single(out, send, group=group)
            return _Done(out) if async_op else out
        if len(set(send_counts) | set(recv_counts)) == 1:      # equal blocks (the device-driven pull): the plain equal-split exchange
            work = dist.all_to_all_single(out, send, group=group, async_op=True)
        else:
            work = dist.all_to_all_single(out, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts), group=group,
                                          async_op=True)

        class _Pending:
            def wait(self_inner):
                work.wait()
                return out
        pend = _Pending()
        pend._keep = send
        return pend if async_op else pend.wait()
    dev = send.device
    s = send.cpu() if send.is_cuda else send
    o = torch.empty(out_shape, dtype=send.dtype)
    ops, so, ro = [], 0, 0
    for p in range(w):
        sb, rb = s[so:so + send_counts[p]], o[ro:ro + recv_counts[p]]
        so, ro = so + send_counts[p], ro + recv_counts[p]
        if p == r:
            rb.copy_(sb)
            continue
        if send_counts[p]:
            ops.append(dist.P2POp(dist.isend, sb, p, group))
        if recv_counts[p]:
            ops.append(dist.P2POp(dist.irecv, rb, p, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    res = o.to(dev) if dev.type == "cuda" else o
    return _Done(res) if async_op else res


class TensorsKey:
    """Identity of a tuple of tensors (address, length, layout, version) -- like ``engine.BatchKey``, for any number of tensors; holds
    them, so an equal address means the same buffer."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.sig = self._sig(tensors)

    @staticmethod
    def _sig(tensors):
        return tuple((t.data_ptr(), t.numel(), t.stride(), t.dtype, t._version) for t in tensors)

    def matches(self, *tensors) -> bool:
        return self.sig == self._sig(tensors) and self.sig == self._sig(self.tensors)


class RowPull:
    """The rows of the owner-sharded PE table one gather will read and this rank does not own, fetched from their owners.

    ``request`` (any time after the batch's ids are known -- one step ahead with a look-ahead): sample the K most recent neighbours of
    the rows, take the distinct ids (neighbours, the rows themselves, the padding row 0) this rank does not own, group them by owner
    and send every owner its list in a fixed-capacity block (ids are 4 bytes: the padding is noise), together with the counts, which
    travel to the host asynchronously.  ``fetch`` (once the owners' rows are final: after the previous update_pe): every rank gathers
    the rows it was asked for, one all-to-all-v moves them (exact sizes: the counts have long arrived), the receiver scatters them
    into its full-size table.  ``wait``: the current stream waits for that scatter.  All on the caller's current stream / the
    communicator ``group`` (a second communicator, so the transfer is not queued behind the backward pass's collectives)."""

    def __init__(self, dl: "DistributedLstep", ids: torch.Tensor, times: torch.Tensor, key: TensorsKey = None):
        from . import _native as nat
        self.dl, self.key = dl, key
        W, rank, dev, rows = dl.W, dl.rank, dl.device, dl.num_rows
        nbr = dl.bb.neighbor_sampler.sample_device(ids, times, dl.K)[0]
        lib = nat.load_library()
        n = nbr.numel() + ids.numel() + 1
        sentinel = W * rows
        keys = torch.empty(n, dtype=torch.int32, device=dev)
        ids = ids.contiguous()
        with torch.cuda.device(dev):
            nat.check(lib.lstep_pull_keys(nat.ptr(nbr), nbr.numel(), nat.ptr(ids), ids.numel(), W, rank, rows, nat.ptr(keys), nat.current_stream()))
        _, _, _, uniq, summary = nat.group_by_key(keys, max(1, int(sentinel + 1).bit_length()), sentinel, wait=None)
        self._parts = (uniq, summary, n)
        self._send_requests(dl._pull_capacity(n))
        self.done = None

    def _send_requests(self, C: int):
        """Every owner receives its block of ``C`` id slots (-1 = unused) and every rank the whole count matrix; nothing here waits for the GPU."""
        from . import _native as nat
        dl = self.dl
        W, dev, rows = dl.W, dl.device, dl.num_rows
        uniq, summary, n = self._parts
        self.req = torch.empty((W, C), dtype=torch.int32, device=dev)
        cnt = torch.empty(W, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            nat.check(nat.load_library().lstep_pull_blocks(nat.ptr(uniq), nat.ptr(summary), W, rows, C, nat.ptr(self.req), nat.ptr(cnt),
                                                           nat.current_stream()))
        self.C = C
        group = dl.pull_group
        self.asked = exchange_rows(self.req.reshape(W * C), [C] * W, [C] * W, group, async_op=True)       # [W * C]: block p = what rank p wants from me
        cm = all_gather_var(cnt.reshape(1, W), group, counts=[1] * W)[0]                                      # [W, W]: cm[p, q] = rows p wants from q
        if cm.is_cuda:
            self.cm_host = torch.empty((W, W), dtype=torch.int32, pin_memory=True)
            self.cm_host.copy_(cm, non_blocking=True)
            self.cm_event = torch.cuda.Event()
            self.cm_event.record()
        else:
            self.cm_host, self.cm_event = cm.clone(), None

    def fetch(self):
        """Serve and receive (call when the owned rows are final on the current stream)."""
        dl = self.dl
        W, rank, C = dl.W, dl.rank, self.C
        if self.cm_event is not None:
            self.cm_event.synchronize()
        cm = self.cm_host.tolist()
        mine = [int(cm[rank][q]) for q in range(W)]            # rows I receive from q
        serve = [int(cm[p][rank]) for p in range(W)]           # rows I send to p
        if max(max(max(row) for row in cm), 0) > C:
            # some owner was asked for more ids than a block holds (ids concentrated on one owner: every rank sees the same matrix and
            # takes this branch together): send the lists again in blocks that hold any list, then go on
            self.asked.wait()
            self._send_requests(self._parts[2])
            dl._pull_min_capacity = max(dl._pull_min_capacity, max(max(row) for row in cm))
            C = self.C
            if self.cm_event is not None:
                self.cm_event.synchronize()
        asked = self.asked.wait().reshape(W, C)
        serve_ids = torch.cat([asked[p, :serve[p]] for p in range(W)]).long()
        rows_out = dl.table.index_select(0, serve_ids)
        pend = exchange_rows(rows_out, serve, mine, dl.pull_group, async_op=True)
        my_ids = torch.cat([self.req[q, :mine[q]] for q in range(W)]).long()
        rows_in = pend.wait()
        if my_ids.numel():
            dl.table.index_copy_(0, my_ids, rows_in)
        self.rows_pulled = int(my_ids.numel())
        if dl.table.is_cuda:
            self.done = torch.cuda.Event()
            self.done.record()
        self.req = self.asked = self._parts = None

    def wait(self):
        if self.done is not None:
            torch.cuda.current_stream(self.dl.device).wait_event(self.done)


# ---------------------------------------------------------------------------------------------- engine
class ShardedSparseRing(HistoryRing):
    """The sparse change-mask ring of the single-GPU engine (``engine.HistoryRing``, sparse mode) over the rows ONE rank owns: local row
    id // W for the nodes with id % W == rank.  The current PE is the REPLICATED full table (``full``), updated in place by every rank;
    a slot receives the rows its batch wrote straight from the update kernels (their ``mirror``, which maps id -> id // W and skips
    rows of other owners), ``oldest`` follows the window.  No shard-sized copy of the current table, no per-batch clone: per batch a
    rank appends only the rows it owns among the ~20 % of the table the batch wrote."""

    def __init__(self, full_table: torch.Tensor, world: int, rank: int, num_fft_batches: int):
        super().__init__(owned_rows(full_table.shape[0], world, rank), full_table.shape[1], num_fft_batches, full_table.device, sparse=True)
        if not self.sparse:
            raise RuntimeError("the owner-sharded sparse ring needs the change mask (at most 126 snapshots, no LSTEP_DENSE/CLONE_HISTORY)")
        self.full, self.world, self.rank = full_table, int(world), int(rank)
        self.table = None

    def owned(self) -> torch.Tensor:
        """Strided view of the rows this rank owns, in local order."""
        return self.full[self.rank::self.world]

    def last(self) -> torch.Tensor:
        assert self.len > 0
        return self.owned()

    def spare(self) -> torch.Tensor:
        return self.full

    base_for_next = spare

    def building(self):
        return self.buf[(self.start + self.len) % self.S]

    def written(self, ids: torch.Tensor, mirrored: bool = False):
        self.mark(ids, self.world, self.rank, copy=not mirrored)

    def commit(self):
        slot = (self.start + self.len) % self.S
        dst = self.buf[slot]
        if self._all_written or self.len == 0:
            dst.copy_(self.owned())
        else:
            if self._written:
                assert self.dev_start is None, "writers without a mirror need the host-resident ring position"
            for ids in self._written:          # writers without a mirror (none on the device-count path): their owned rows, copied now
                own = ids[ids % self.world == self.rank]
                dst.index_copy_(0, own // self.world, self.full[own])
            # local row 0 carries the always-set change bit (lstep_history_slot_bits: it is the padding row on rank 0): keep it valid.
            # (lstep_copy_rows copies row r of the source to row r of the slot: the source pointer is moved to full[rank], r = 0; the slot is
            # picked on the device when the ring position lives there)
            from . import _native as nat
            ref = self._ref(slot)
            with torch.cuda.device(self.buf.device):
                nat.check(nat.load_library().lstep_copy_rows(nat.ptr(self.buf if ref is not None else dst), nat.ptr(self.full[self.rank:]), self.P, self.P,
                                                             nat.ptr(self._row0), 1, self.rows, ref, nat.current_stream()))
        if self.len == 0:
            self.oldest.copy_(self.owned())
        self._written, self._all_written = [], False
        if self.len < self.T:
            self.len += 1
        else:
            self.start = (self.start + 1) % self.S
            assert self._advance is None, "HistoryRing.apply_advance() must follow every commit()"
            self._advance = self.start
        self.begin_slot()

    def adopt_full_slots(self):
        self.recompute_mask()
        self._reposition()
        self._written, self._all_written, self._advance, self._advanced = [], False, None, [None, None]
        if self.len:
            self.oldest.copy_(self.buf[self.start])
        self.begin_slot()


class DistributedLstep:
    """Drives one ``LstepEngine``'s model over a global batch shared by all ranks of ``group``."""

    def __init__(self, engine: LstepEngine, optimizer=None, group=None, probe: dict = None):
        """``probe`` = {"captured": bool, "pull": bool}: what a rehearsal of captured collectives ACROSS the ranks of ``group`` found
        (tools/rccl_graph_probe.py, run by ``bench.py`` in a child process per rank before the rank initialises HIP, agreed by an
        all-reduce).  It is what lets a multi-rank job take the two things this torch / RCCL pair can get wrong in a way that kills the
        process: the whole-step HIP graph with the collectives inside (W > 1: only with probe["captured"], or LSTEP_DIST_GRAPH=1), and the
        owner-sharded "pull" form as the default beyond four ranks (only with probe["pull"]).  Without a probe a multi-rank job runs the
        device-driven iteration launch by launch in the "replicate" form."""
        self.eng = engine
        self.probe = dict(probe) if probe else None
        self.bb, self.predictor = engine.backbone, engine.predictor
        self.group = group
        self.W, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.K, self.G = engine.K, engine.G
        dev = engine.device
        self.device = dev
        rows = self.bb.node_raw_features.shape[0]
        self.num_rows = rows
        self.table = torch.zeros((rows, self.bb.pe_dim), dtype=torch.float32, device=dev)  # replicated current PE
        self.slot_of = engine.slot_of
        engine.ring = None  # the unsharded ring is not used (and must not be allocated at scale)
        # Three ways to run update_pe across ranks (``_choose_form``; module docstring): "pull" -- owner-sharded table, owner-computes,
        # pulled gather rows; "replicate" -- every rank recomputes the whole update on a replicated table; "allgather" -- owner-computes with
        # all-gathers of every new row into replicated tables (clone ring shards, host-sized).
        self.form = self._choose_form()
        self.replicated = self.form == "replicate"
        self.pull_group = None
        self._pending_pull, self._pull_min_capacity = None, 0
        # the pull's own stream: its request (a sampler launch and a ~1 M-key sort) runs beside the forward pass, its row exchange behind
        # update_pe -- on neither the critical stream nor in front of update_pe on the update stream
        self._pull_stream = nat.role_stream(dev, "pull", priority=-1 if os.environ.get("LSTEP_PULL_PRIORITY", "1") == "1" else 0)
        if self.form in ("replicate", "pull"):
            self._ring = ShardedSparseRing(self.table, self.W, self.rank, self.bb.num_fft_batches)
            engine.ring = self._ring        # (lets the engine's grouping helpers take their device-count branch; its own iterations are not used)
            if engine.use_aux:
                from .model import _aux_stream
                self._ring.advance_stream = _aux_stream(dev)
            if self.form == "pull":
                # launch by launch (gloo; LSTEP_PULL_COMM=own; LSTEP_DIST_GRAPH=0) the pulled rows travel on a communicator of their own: on the
                # main one they would queue behind the backward pass's reduce-scatter / all-reduce (a communicator runs its collectives in
                # issue order) instead of underneath them.  A captured iteration keeps every collective on the main communicator (below).
                captured = (os.environ.get("LSTEP_DIST_HOST_COUNTS") != "1" and self._graph_allowed(group)
                            and os.environ.get("LSTEP_PULL_COMM", "main") != "own")
                self.pull_group = dist.new_group(backend=dist.get_backend(group)) if not (_skip_single(self.W) or captured) else group
        else:
            self._ring = HistoryRing(owned_rows(rows, self.W, self.rank), self.bb.pe_dim, self.bb.num_fft_batches, dev)
        self._copy_stream = nat.role_stream(dev, "dist-copy")
        # DEVICE-DRIVEN iteration (round 4): every collective has a fixed capacity and its live count travels on the device
        # (``lstep_owner_partition``), so the host never waits for the GPU inside an iteration and -- over RCCL -- the whole training
        # iteration, collectives included, is captured once and replayed as ONE HIP graph (``GraphedDistStep``), like the single-GPU engine's.
        # LSTEP_DIST_HOST_COUNTS=1: the host-sized iterations of rounds 2-3 (A/B; also what "allgather" always takes).
        self.device_driven = self.form in ("replicate", "pull") and os.environ.get("LSTEP_DIST_HOST_COUNTS") != "1"
        self.use_step_graph = self.device_driven and self._graph_allowed(group)
        self._graphed, self._steady_steps = {}, 0
        self._overflow = torch.zeros(1, dtype=torch.int32, device=dev)      # sticky: some fixed-capacity block was too small
        self._overflow_poll = None
        self.comm_log = None        # bench.py: a list that receives (name, bytes, start event, end event) of every collective issued eagerly
        if self.form == "pull" and self.use_step_graph and os.environ.get("LSTEP_PULL_COMM", "main") == "own":
            # LSTEP_PULL_COMM=own: the pull keeps its own communicator and stream and the iteration is issued launch by launch (capturing an
            # all-to-all on a second communicator from a side stream faulted in RCCL 2.26: tools/rccl_capture_probe.py)
            self.use_step_graph = False
        # update_pe's layers are forward-only (no gradient ever reaches them): they stay out of the bucket
        frozen = {id(p) for m in (self.bb.pe_mlp_1, self.bb.pe_mlp_2, self.bb.self_update_pe) for p in m.parameters()}
        self._trainable = [p for p in list(self.bb.parameters()) + list(self.predictor.parameters()) if id(p) not in frozen]

    def _graph_allowed(self, group) -> bool:
        """May the iteration be captured as one HIP graph with its RCCL collectives inside?  One rank: yes (rehearsed on hardware: world
        size 1 with every collective forced).  More ranks: capturing collectives that really cross links has never run on this pool, and
        the patterns this torch / RCCL pair gets wrong SIGSEGV the process (tools/rccl_capture_probe.py) -- so only when a cross-rank
        rehearsal came back clean (``probe["captured"]``) or the caller insists (LSTEP_DIST_GRAPH=1).  LSTEP_DIST_GRAPH=0: never."""
        flag = os.environ.get("LSTEP_DIST_GRAPH")
        if flag == "0" or torch.device(self.device).type != "cuda" or dist.get_backend(group) == "gloo":
            return False
        if self.W == 1 or flag == "1":
            return True
        return bool(self.probe and self.probe.get("captured"))

    # ---- state import/export (tests, checkpoints)
    def load_history(self, history: torch.Tensor):
        """Adopt a reference-shaped ``[N+1, t, P]`` history: this rank keeps its owned rows."""
        if history.shape[1]:
            self.table.copy_(history[:, -1, :])
        self.ring.load(history[self.rank::self.W])

    def _device_update_ok(self) -> bool:
        bb = self.bb
        return (torch.device(self.device).type == "cuda" and bb._fused_tail_ok() and bb.num_fft_batches <= 126
                and getattr(bb.neighbor_sampler, "sample_neighbor_strategy", "recent") == "recent"
                and not any(os.environ.get(v) == "1" for v in ("LSTEP_TORCH_UPDATE", "LSTEP_TORCH_ENTRIES", "LSTEP_HOST_COUNTS",
                                                                 "LSTEP_DENSE_HISTORY", "LSTEP_CLONE_HISTORY")))

    def _append_snapshot(self):
        """This rank's rows of the finished table become the newest snapshot of its ring shard.  The strided copy (N / W rows of 688 B)
        runs on a copy stream; the next history read waits for it (``_wait_snapshot``)."""
        if self.table.is_cuda:
            main = torch.cuda.current_stream(self.device)
            self._copy_stream.wait_stream(main)
            with torch.cuda.stream(self._copy_stream):
                self._ring.spare().copy_(self.table[self.rank::self.W])
            self.table.record_stream(self._copy_stream)
            self._snapshot_pending = True
        else:
            self._ring.spare().copy_(self.table[self.rank::self.W])
        self._ring.commit()

    @property
    def ring(self) -> HistoryRing:
        """The ring shard, safe to read on the current stream (a pending snapshot copy is waited for first)."""
        self._wait_snapshot()
        return self._ring

    def _wait_snapshot(self):
        if getattr(self, "_snapshot_pending", False):
            torch.cuda.current_stream(self.device).wait_stream(self._copy_stream)
            self._snapshot_pending = False

    # ---- pieces
    def _prefetch(self, lookahead):
        """Group the next global batch's endpoints now (``LstepEngine.prefetch_batch_nodes``) and send the number of its batch nodes per
        owner rank to the host behind it, so the next iteration starts without waiting for the GPU."""
        self.eng.prefetch_batch_nodes(*lookahead)
        key, (_, _, uniq, pending) = self.eng._prefetched_group
        n_unique = pending[0] if isinstance(pending, torch.Tensor) else pending._keep[0]     # (device summary / pending host count)
        valid = torch.arange(uniq.numel(), device=uniq.device) < n_unique               # entries past n_unique are uninitialised
        ranks = torch.arange(self.W, device=uniq.device, dtype=uniq.dtype)
        # (not torch.bincount: it reads the largest value back to size its output, a host synchronisation)
        per_owner = ((torch.remainder(uniq, self.W).unsqueeze(1) == ranks) & valid.unsqueeze(1)).sum(dim=0)
        host = torch.empty(self.W, dtype=torch.int64, pin_memory=True)
        host.copy_(per_owner, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._owner_prefetch = (key, host, done, per_owner)

    def _prefetched_owner_counts(self, src, dst):
        pre = self.__dict__.pop("_owner_prefetch", None)
        if pre is None or not pre[0].matches(src, dst):
            return None
        pre[2].synchronize()
        return [int(c) for c in pre[1].tolist()]

    def _splice(self, bn: torch.Tensor, batch_idx: int, owner_counts=None):
        """Owner-sharded FFT filter + all-gather of the filtered rows; returns (local rows with grad, leaf of all rows, perm)."""
        return self._splice_finish(self._splice_start(bn, batch_idx, owner_counts))

    def _splice_start(self, bn: torch.Tensor, batch_idx: int, owner_counts=None):
        """The local half of the splice: filter the history of the batch nodes this rank owns and put the all-gather of the filtered rows
        IN FLIGHT (``PendingGather``).  Whatever does not read the PE table -- the edge / node channels of the gather stage, 70 % of
        its bytes -- can be enqueued before ``_splice_finish`` and runs underneath the collective."""
        self._wait_snapshot()
        owner = bn % self.W
        # every rank derives the same counts: no size exchange.  With a look-ahead they were computed one iteration ago and are already
        # on the host (``_prefetch``); otherwise this is a host synchronisation.
        counts = owner_counts if owner_counts is not None else torch.bincount(owner, minlength=self.W).tolist()
        order = torch.argsort(owner, stable=True)                      # bn is sorted by id: this orders it by (owner rank, node id)
        first = sum(counts[:self.rank])
        owned_idx = order[first:first + counts[self.rank]]             # positions in bn of the nodes this rank owns
        self._owned_idx, self._owner_order = owned_idx, order
        mine = bn[owned_idx]                                           # (sizes known on the host: no boolean-mask compaction)
        self.ring.wait_window()
        rows_mine = self.bb.filter_history(self.ring.buf, self.ring.geom(), mine // self.W, batch_idx, mask=self.ring.mask,
                                           oldest=self.ring.oldest)
        return bn, rows_mine, PendingGather(rows_mine.detach(), self.group, counts=counts), order, counts

    def _splice_finish(self, started):
        bn, rows_mine, pending, order, counts = started
        gathered = pending.wait()
        # gathered is ordered by (owner rank, node id); bn is ordered by node id
        rows_all = torch.empty_like(gathered)
        rows_all[order] = gathered
        self.table.index_copy_(0, bn, rows_all)
        self.slot_of[bn] = torch.arange(bn.numel(), dtype=torch.int32, device=self.device)
        leaf = rows_all.detach().requires_grad_(True)
        return rows_mine, leaf, (order, counts)

    def _probabilities(self, a, b):
        return self.predictor(input_1=a, input_2=b).squeeze(dim=-1).sigmoid().clamp(0, 1)

    def _rows_with_ids(self, ids: torch.Tensor) -> torch.Tensor:
        """[n, P + 4] block: the current table rows of ``ids`` with the ids packed into the padding columns (one collective)."""
        P = self.bb.pe_dim
        z = torch.zeros((ids.numel(), P + 4), dtype=torch.float32, device=self.device)
        z[:, :P] = self.table[ids]
        return pack_ids(z, ids, P)

    def _write_rows(self, z_all: torch.Tensor):
        P = self.bb.pe_dim
        self.table.index_copy_(0, unpack_ids(z_all, P), z_all[:, :P])

    def _update_phase1(self, bn, src, dst, ts, presorted=None, owner_counts=None, first: bool = False, owned_idx=None):
        """update_pe phase 1 completely (its rows feed phase 2); no host synchronisation on the engine's fast path.
        Fused path (default widths): every rank updates the rows it owns IN PLACE (``lstep_update_rows``) and the new rows are
        all-gathered; library path: the pre-activation rows z are gathered and every replica applies pe += tanh(z)."""
        now32 = ts.max().to(torch.float32)       # torch.Tensor([current_time]) of the reference: float32-rounded, kept on the device
        shard = (self.W, self.rank)
        P = self.bb.pe_dim
        fused = self.bb._fused_tail_ok() and os.environ.get("LSTEP_TORCH_UPDATE") != "1"
        # change bits of the snapshot this batch builds: the owned rows among the batch nodes (spliced rows, phase 1) and the owned rows
        # phase 2 touches (each rank computes exactly the rows it owns)
        ring = self._ring
        ring.begin_slot(all_changed=first)
        ring.mark(bn, self.W, self.rank)
        if fused:
            ids = self.bb.update_pe_phase1(self.table, bn, src, dst, ts, now32, shard=shard, presorted=presorted, fused=True,
                                           owned_idx=owned_idx)
            if not _skip_single(self.W):
                self._write_rows(all_gather_var(self._rows_with_ids(ids), self.group, counts=owner_counts)[0])
        else:
            ids, z = self.bb.update_pe_phase1(self.table, bn, src, dst, ts, now32, shard=shard, presorted=presorted)
            # ids ride in z's padding columns: one collective per phase; phase-1 row counts are known locally
            z_all, _ = all_gather_var(pack_ids(z, ids, P), self.group, counts=owner_counts)
            self.bb.apply_residual_tanh(self.table, unpack_ids(z_all, P), z_all)       # every replica applies the same update
        return now32, fused

    def _choose_form(self) -> str:
        """LSTEP_PHASE2 = auto | replicate | allgather | pull (module docstring).
        ``auto``: "replicate" up to W = 4 -- zero update bytes and the single-GPU engine's host-free path; the redundant work (phase 2
        touches 0.72 / 0.91 M rows at W = 2 / 4 on c4) still fits under the backward pass -- and "pull" beyond (when the job's probe of its
        exchange pattern is clean, below): at W = 8 on c5 the
        replicated update would touch 2.9 M rows per step (~6.7 ms of update kernels against a ~3.4 ms step) and an all-gather of them
        would move 1.77 GB per rank, while the pull moves the ~0.55 GB a rank's next gather reads and keeps the update at 1 / W of the
        rows.  Configurations the device-count update does not cover (non-default widths, RNG-defined sampling, T > 126) take
        "allgather", the host-sized owner-computes form."""
        policy = os.environ.get("LSTEP_PHASE2", "auto")
        if policy not in ("auto", "replicate", "allgather", "pull"):
            raise ValueError("LSTEP_PHASE2 must be auto, replicate, allgather or pull")
        if not self._device_update_ok():
            if policy in ("replicate", "pull"):
                raise RuntimeError(f"LSTEP_PHASE2={policy} needs the device-count update path (default widths, 'recent' sampling, T <= 126, a GPU)")
            return "allgather"
        if policy == "auto":
            # "pull" is the form whose bytes and FLOPs are flat in W (expected efficiency ~0.6-0.7 at W = 8 against
            # ~0.45 for "replicate"), but its exchange pattern -- two all-to-alls per step, captured from the capturing stream -- has only
            # ever run on ONE real RCCL rank.  So it is taken beyond LSTEP_PULL_MIN_WORLD - 1 = four ranks exactly when a rehearsal of that
            # pattern across the job's real ranks came back clean on every rank (``probe["pull"]``, tools/rccl_graph_probe.py through
            # bench.py), and "replicate" -- three plain equal-block collectives per step on one communicator -- otherwise.
            min_world = int(os.environ.get("LSTEP_PULL_MIN_WORLD", "5"))
            if self.W >= min_world and self.probe and self.probe.get("pull"):
                return "pull"
            return "replicate"
        return policy

    def _phase2_replicated(self) -> bool:
        return self.form == "replicate"

    def _update_phase2(self, bn, ts, state):
        """update_pe phase 2 up to the all-gather of its rows, which is left in flight."""
        now32, fused = state
        shard = (self.W, self.rank)
        ring = self._ring
        if fused:
            replicate = self._phase2_replicated()
            ids = self.bb.update_pe_phase2(self.table, bn, ts, now32, self.K, shard=None if replicate else shard, fused=True)
            ring.mark(ids, self.W, self.rank)          # (marks the rows of ``ids`` this rank owns)
            if replicate:
                return ("rows", None)
            return ("rows", PendingGather(self._rows_with_ids(ids), self.group) if not _skip_single(self.W) else None)
        ids, z = self.bb.update_pe_phase2(self.table, bn, ts, now32, self.K, shard=shard)
        ring.mark(ids, self.W, self.rank)
        return ("z", PendingGather(pack_ids(z, ids, self.bb.pe_dim), self.group))

    def _update_start(self, bn, src, dst, ts, presorted=None, owner_counts=None, first: bool = False):
        return self._update_phase2(bn, ts, self._update_phase1(bn, src, dst, ts, presorted, owner_counts, first))

    def _update_finish(self, pending):
        """Apply the gathered phase-2 rows on every replica and append the snapshot to this rank's ring shard."""
        kind, gather = pending
        if kind == "rows":
            if gather is not None:
                self._write_rows(gather.wait())
        else:
            z_all = gather.wait()
            self.bb.apply_residual_tanh(self.table, unpack_ids(z_all, self.bb.pe_dim), z_all)
        self._append_snapshot()

    # ---- train:204-311 on a global batch of W*B edges (every rank passes the SAME arrays)
    def train_iteration(self, optimizer, batch_idx: int, src, dst, ts, eid, neg_dst, initial_pe: torch.Tensor = None, lookahead=None):
        if self.device_driven:
            return self._train_iteration_device_driven(optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead)
        with self.eng.aux_streams():
            if self.form == "pull":
                return self._train_iteration_pull(optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead)
            if self.replicated:
                return self._train_iteration_replicated(optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead)
            return self._train_iteration(optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead)

    # ---- the replicated-update form: forward / backward exactly as below, update_pe = the single-GPU engine's, on the global batch
    def _global_batch_nodes(self, src, dst, batch_idx):
        """(exact sorted batch nodes, capacity-sized list with its device count, grouping, rows per owner).  The per-owner counts come
        from the look-ahead of the previous iteration when there was one (no host wait); otherwise the host waits for the grouping."""
        owner_counts = self._prefetched_owner_counts(src, dst)      # (before the grouping below consumes the engine's prefetch)
        bn_cap, n_live, presorted = self.eng.batch_nodes_device(src, dst)
        if owner_counts is None:
            u = int(n_live.item())
            owner_counts = torch.bincount(bn_cap[:u] % self.W, minlength=self.W).tolist()
        u = sum(owner_counts)
        return bn_cap[:u], bn_cap, n_live, presorted, owner_counts

    def _forward_on_slice(self, bn, batch_idx, owner_counts, src, dst, neg_dst, ts, pull=None):
        """FFT splice (owner-sharded filter + all-gather) and this rank's slice through gather, dense tail, predictor and loss.
        ``pull`` (owner-sharded table): the fetched rows this gather reads; their scatter must precede the write of the spliced rows."""
        n_glob = src.numel()
        b = n_glob // self.W
        sl = slice(self.rank * b, (self.rank + 1) * b)
        started = self._splice_start(bn, batch_idx, owner_counts)
        s_, d_, n_, t_ = src[sl], dst[sl], neg_dst[sl], ts[sl]
        ids3, t3 = torch.cat([s_, d_, n_]), torch.cat([t_, t_, t_])
        fused = self.bb._fused_tail_ok()
        # edge + node channels first: they read no PE row, so their launch overlaps the all-gather of the filtered rows
        x_edge, x_node, _, _, _ = self.bb._gather(None, ids3, t3, self.K, self.G, nat_branch("edge_node"), wide=fused, row_blocks=3)
        if pull is not None:
            pull.wait()      # (rows of this batch's nodes were fetched BEFORE their splice: the all-gathered spliced rows overwrite them next)
        rows_mine, leaf, (owner_order, owner_counts) = self._splice_finish(started)
        spliced = SplicedRows(leaf, self.slot_of)
        _, _, x_pe, own, _ = self.bb._gather(self.table, ids3, t3, self.K, self.G, nat_branch("pe"), spliced, wide=fused, row_blocks=3)
        emb_p = self.bb._combined_tail(x_edge, x_node, x_pe, own, fused)
        logits = self.predictor.pair_logits(emb_p, b, (0, b, 0, 2 * b))
        loss, lp_loss, pe_loss, predicts = _LinkLoss.apply(logits, leaf, self.table, self.slot_of, ids3, self.eng.pe_weight,
                                                           self.eng.neg_sample_weight)
        out = {"lp_loss": lp_loss.detach(), "pe_loss": pe_loss.detach(), "loss": loss.detach(), "predicts": predicts.detach()}
        return out, loss, rows_mine, leaf, owner_order, owner_counts

    # ---- the owner-sharded form ("pull"): module docstring, ``RowPull``
    def _slice_rows(self, blocks, ts):
        """This rank's gather rows of a global batch: cat of its B-edge slice of every id block, with the slice's times repeated."""
        b = ts.numel() // self.W
        sl = slice(self.rank * b, (self.rank + 1) * b)
        return torch.cat([x[sl] for x in blocks]), torch.cat([ts[sl]] * len(blocks))

    def _pull_capacity(self, n: int) -> int:
        """Id slots per owner in the request exchange: 1.5 x an even split of the n candidate ids (or 1.25 x the longest list seen),
        never more than n; a list that does not fit makes every rank resend in blocks of n (``RowPull.fetch``)."""
        even = -(-n // self.W)
        want = max(even + even // 2, self._pull_min_capacity + self._pull_min_capacity // 4, 1024)
        return min(n, -(-want // 1024) * 1024)

    def _pull_capacity_dev(self, n: int) -> int:
        """Id slots per owner of the device-driven pull (``RowPullDev``).  The rows travel in blocks of this size too, so every slot is paid
        in bytes: the capacity follows the LARGEST request list seen so far (the per-owner counts of every request travel to the host
        asynchronously, ``_note_pull_counts``; never waited for) times LSTEP_PULL_SLACK (default 1.25), in steps of 4096.  Until a count has
        arrived: an even split of the n candidate ids (an upper bound of the distinct ones).  An overflow is detected on the device and
        reported by ``check_capacity``; a captured iteration whose capacity has become too small for what was seen since is captured again
        (``GraphedDistStep.step``)."""
        slack = float(os.environ.get("LSTEP_PULL_SLACK", "1.25"))
        seen = self._pull_seen_max()
        if seen is None:
            want = int(-(-n // self.W) * slack) + 256
        else:
            want = int(seen * slack) + 1024
        return max(4096, min(n, -(-want // 4096) * 4096))

    def _pull_seen_max(self):
        from .model import _LiveCount
        tr = self.__dict__.setdefault("_pull_track", _LiveCount())
        # (no event query while this thread captures: hipEventQuery is one of the calls a capture forbids)
        got = None if (self.table.is_cuda and torch.cuda.is_current_stream_capturing()) else tr.poll()
        if got is not None:
            self._pull_seen = max(self.__dict__.get("_pull_seen", 0), int(got))
        return self.__dict__.get("_pull_seen")

    def _note_pull_counts(self, cnt_max: torch.Tensor):
        """Send the largest per-owner request count of one pull to the host (asynchronously; not inside a capture)."""
        from .model import _LiveCount
        if cnt_max.is_cuda and not torch.cuda.is_current_stream_capturing():
            self.__dict__.setdefault("_pull_track", _LiveCount()).send(cnt_max)
        elif not cnt_max.is_cuda:
            self._pull_seen = max(self.__dict__.get("_pull_seen", 0), int(cnt_max))

    def _pull_now(self, blocks, ts):
        """No look-ahead had the rows fetched: request and fetch on the current stream (two host waits)."""
        pull = RowPull(self, *self._slice_rows(blocks, ts))
        self._poison_foreign_rows()
        pull.fetch()
        return pull

    def _take_pull(self, *tensors):
        pend, self._pending_pull = self._pending_pull, None
        return pend if (pend is not None and pend.key is not None and pend.key.matches(*tensors)) else None

    def _owned_positions(self, bn, owner_counts):
        order = torch.argsort(bn % self.W, stable=True)
        self._owner_order = order
        first = sum(owner_counts[:self.rank])
        return order[first:first + owner_counts[self.rank]]

    def _share_phase1_rows(self, owner_counts, bn):
        """``update_pe_device(after_phase1=...)``: phase 2's messages carry the phase-1 rows of ALL batch nodes (models/LSTEP.py:311-320),
        so every owner hands out the ones it just computed -- the "all-gather of updated positional encodings" of north_star: U x 688 B.
        Every rank knows which rows arrive in which order (the batch nodes by (owner, id): ``_owner_order``): plain rows travel."""
        def share(ids1):
            if not _skip_single(self.W):
                rows_all, _ = all_gather_var(self.table.index_select(0, ids1), self.group, counts=owner_counts)
                self.table.index_copy_(0, bn.index_select(0, self._owner_order), rows_all)
        return share

    def full_table(self) -> torch.Tensor:
        """The whole current PE table assembled from its owners (all-gather of the owned rows) -- a new tensor; this rank's own table,
        a cache outside its owned rows, is left as it is."""
        if _skip_single(self.W):
            return self.table.clone()
        mx = owned_rows(self.num_rows, self.W, 0)
        cat, _ = all_gather_var(_pad_rows(self.table[self.rank::self.W].contiguous(), mx), self.group, counts=[mx] * self.W)
        out = torch.empty_like(self.table)
        for r in range(self.W):
            out[r::self.W] = cat[r * mx:r * mx + owned_rows(self.num_rows, self.W, r)]
        return out

    def sync_full_table(self):
        """Make every row of this rank's table valid: checkpoints, and batch 0, where the reference hands the updated table back through
        ``initial_positional_encoding`` (train:281,286)."""
        if not _skip_single(self.W):
            self.table.copy_(self.full_table())

    def _poison_foreign_rows(self):
        """LSTEP_PULL_POISON=1 (tests): every row this rank does not own becomes NaN right before the rows of the next gather are
        fetched -- whatever the next iteration reads must then have arrived through the pull or one of the two all-gathers."""
        if os.environ.get("LSTEP_PULL_POISON") != "1":
            return
        if getattr(self, "_foreign", None) is None:
            self._foreign = (torch.remainder(torch.arange(self.num_rows, device=self.device), self.W) != self.rank).unsqueeze(1)
        self.table.masked_fill_(self._foreign, float("nan"))

    def _train_iteration_pull(self, optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead):
        assert src.numel() % self.W == 0, "global batch must divide by the world size"
        bb, ring = self.bb, self._ring
        bb.prepare_step()
        bn, bn_cap, n_live, presorted, owner_counts = self._global_batch_nodes(src, dst, batch_idx)
        out, loss = None, None
        if batch_idx == 0:
            self.table.copy_(initial_pe)         # every row valid on every rank
            self._pending_pull = None
            ring.begin_slot(all_changed=True)
            owned_idx = self._owned_positions(bn, owner_counts)
        else:
            pull = self._take_pull(src, dst, ts, neg_dst) or self._pull_now((src, dst, neg_dst), ts)
            out, loss, rows_mine, leaf, owner_order, owner_counts = self._forward_on_slice(bn, batch_idx, owner_counts, src, dst, neg_dst, ts, pull=pull)
            owned_idx = self._owned_idx
        if lookahead is not None:
            self._prefetch(lookahead[:2])
        ahead = lookahead if (lookahead is not None and len(lookahead) >= 4) else None
        nxt = None
        ps = self._pull_stream
        if ahead is not None:       # the next batch's requests: independent of everything this iteration computes
            if ps is not None:
                ps.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                nxt = RowPull(self, *self._slice_rows((ahead[0], ahead[1], ahead[3]), ahead[2]), key=TensorsKey(ahead[0], ahead[1], ahead[2], ahead[3]))

        def update_and_append():
            bb.update_pe_device(self.table, bn_cap, n_live, src, dst, ts, self.K, presorted, changed=ring.written, mirror=ring.building(),
                                mirror_shard=(self.W, self.rank), owner=(self.W, self.rank), owned_idx=owned_idx,
                                after_phase1=self._share_phase1_rows(owner_counts, bn))
            if batch_idx == 0 and initial_pe is not None:
                self.sync_full_table()
                initial_pe.copy_(self.table)
            ring.commit()

        def fetch_next(after=None):
            """Serve and receive the next gather's rows on the pull stream, behind ``after`` (update_pe's end) or the current stream."""
            if ps is not None:
                if after is not None:
                    ps.wait_event(after)
                else:
                    ps.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                self._poison_foreign_rows()
                nxt.fetch()
            self._pending_pull = nxt

        if loss is None:
            update_and_append()
            ring.apply_advance()
            if nxt is not None:
                fetch_next()
            return out
        main, side = torch.cuda.current_stream(self.device), self.eng._update_stream
        overlap = self.eng.overlap_update
        updated = None
        if overlap:
            side.wait_stream(main)       # after the forward pass: it reads the table update_pe is about to rewrite
            with torch.cuda.stream(side):
                update_and_append()
                updated = torch.cuda.Event()
                updated.record()
        else:
            update_and_append()
        optimizer.zero_grad()
        (loss / self.W).backward()                       # global mean = mean of the rank means
        g_rows = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
        g_mine = reduce_scatter_var(g_rows[owner_order].contiguous(), owner_counts, self.group)
        if rows_mine.numel():
            rows_mine.backward(g_mine)                   # -> fft_filter / fft_agg through this rank's history shard
        bb.join_aux_stream()
        v = all_reduce_gradients(self._trainable, self.group, extra=torch.stack([out["lp_loss"], out["pe_loss"], out["loss"]]))
        ring.apply_advance()             # the backward pass is enqueued: the window's oldest snapshot may move on behind it
        if nxt is not None:
            # the rows of the NEXT gather: behind update_pe (the owners' rows are final there), underneath the backward pass on the GPU;
            # the host only reads counts that arrived while it was enqueueing the backward pass
            fetch_next(updated)
        if updated is not None:
            main.wait_event(updated)     # the optimiser may only step once update_pe has read its weights; the ring shard is appended
        optimizer.step()
        self.slot_of.index_fill_(0, bn, -1)
        out["lp_loss"], out["pe_loss"], out["loss"] = (v / self.W).unbind(0)       # global means (they travelled with the gradient bucket)
        return out

    def _eval_iteration_pull(self, batch_idx, src, dst, ts, eid, neg_src, neg_dst, lookahead):
        n_glob = src.numel()
        b = n_glob // self.W
        bb, ring = self.bb, self._ring
        bn, bn_cap, n_live, presorted, owner_counts = self._global_batch_nodes(src, dst, batch_idx)
        pull = self._take_pull(src, dst, ts, neg_src, neg_dst) or self._pull_now((src, dst, neg_src, neg_dst), ts)
        started = self._splice_start(bn, batch_idx, owner_counts)
        pull.wait()
        self._splice_finish(started)
        owned_idx = self._owned_idx
        self.slot_of.index_fill_(0, bn, -1)
        ids, t4 = self._slice_rows((src, dst, neg_src, neg_dst), ts)
        emb_p = bb.combining_pe_raw_feat(self.table, ids, t4, self.K, self.G, padded=True, row_blocks=4)
        if self.predictor.fused_ok(emb_p):
            predicts = self.predictor.pair_logits(emb_p, b, (0, b, 2 * b, 3 * b)).sigmoid().clamp(0, 1)
        else:
            emb = emb_p[:, :bb.feat_dim]
            predicts = torch.cat([self._probabilities(emb[:b], emb[b:2 * b]), self._probabilities(emb[2 * b:3 * b], emb[3 * b:])], dim=0)
        labels = torch.cat([torch.ones(b, device=self.device), torch.zeros(b, device=self.device)])
        if lookahead is not None:
            self._prefetch(lookahead[:2])
        nxt = None
        ps = self._pull_stream
        if lookahead is not None and len(lookahead) >= 5:
            if ps is not None:
                ps.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                nxt = RowPull(self, *self._slice_rows((lookahead[0], lookahead[1], lookahead[3], lookahead[4]), lookahead[2]), key=TensorsKey(*lookahead[:5]))
        bb.update_pe_device(self.table, bn_cap, n_live, src, dst, ts, self.K, presorted, changed=ring.written, mirror=ring.building(),
                            mirror_shard=(self.W, self.rank), owner=(self.W, self.rank), owned_idx=owned_idx,
                            after_phase1=self._share_phase1_rows(owner_counts, bn))
        ring.commit()
        ring.apply_advance()
        if nxt is not None:
            if ps is not None:
                ps.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                self._poison_foreign_rows()
                nxt.fetch()
            self._pending_pull = nxt
        return {"loss": F.binary_cross_entropy(predicts, labels), "predicts": predicts}

    def _train_iteration_replicated(self, optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead):
        assert src.numel() % self.W == 0, "global batch must divide by the world size"
        bb, ring = self.bb, self._ring
        bb.prepare_step()
        bn, bn_cap, n_live, presorted, owner_counts = self._global_batch_nodes(src, dst, batch_idx)
        out, loss = None, None
        if batch_idx == 0:
            self.table.copy_(initial_pe)
            ring.begin_slot(all_changed=True)
        else:
            out, loss, rows_mine, leaf, owner_order, owner_counts = self._forward_on_slice(bn, batch_idx, owner_counts, src, dst, neg_dst, ts)
        if lookahead is not None:
            self._prefetch(lookahead[:2])

        def update_and_append():
            # every rank updates ALL rows of its replica (same inputs everywhere) and mirrors the ones it owns into its ring shard
            bb.update_pe_device(self.table, bn_cap, n_live, src, dst, ts, self.K, presorted, changed=ring.written, mirror=ring.building(),
                                mirror_shard=(self.W, self.rank))
            if batch_idx == 0 and initial_pe is not None:
                initial_pe.copy_(self.table)
            ring.commit()

        if loss is None:
            update_and_append()
            ring.apply_advance()
            return out
        main, side = torch.cuda.current_stream(self.device), self.eng._update_stream
        overlap = self.eng.overlap_update
        if overlap:
            side.wait_stream(main)       # after the forward pass: it reads the table update_pe is about to rewrite
            with torch.cuda.stream(side):
                update_and_append()
        else:
            update_and_append()
        optimizer.zero_grad()
        (loss / self.W).backward()                       # global mean = mean of the rank means
        g_rows = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
        g_mine = reduce_scatter_var(g_rows[owner_order].contiguous(), owner_counts, self.group)
        if rows_mine.numel():
            rows_mine.backward(g_mine)                   # -> fft_filter / fft_agg through this rank's history shard
        bb.join_aux_stream()
        v = all_reduce_gradients(self._trainable, self.group, extra=torch.stack([out["lp_loss"], out["pe_loss"], out["loss"]]))
        ring.apply_advance()             # the backward pass is enqueued: the window's oldest snapshot may move on behind it
        if overlap:
            main.wait_stream(side)       # the optimiser may only step once update_pe has read its weights
        optimizer.step()
        self.slot_of.index_fill_(0, bn, -1)
        out["lp_loss"], out["pe_loss"], out["loss"] = (v / self.W).unbind(0)       # global means (they travelled with the gradient bucket)
        return out

    def _eval_iteration_replicated(self, batch_idx, src, dst, ts, eid, neg_src, neg_dst, lookahead):
        n_glob = src.numel()
        b = n_glob // self.W
        sl = slice(self.rank * b, (self.rank + 1) * b)
        bb, ring = self.bb, self._ring
        bn, bn_cap, n_live, presorted, owner_counts = self._global_batch_nodes(src, dst, batch_idx)
        self._splice(bn, batch_idx, owner_counts)
        self.slot_of.index_fill_(0, bn, -1)
        ids = torch.cat([src[sl], dst[sl], neg_src[sl], neg_dst[sl]])
        emb_p = bb.combining_pe_raw_feat(self.table, ids, torch.cat([ts[sl]] * 4), self.K, self.G, padded=True, row_blocks=4)
        if self.predictor.fused_ok(emb_p):
            predicts = self.predictor.pair_logits(emb_p, b, (0, b, 2 * b, 3 * b)).sigmoid().clamp(0, 1)
        else:
            emb = emb_p[:, :bb.feat_dim]
            predicts = torch.cat([self._probabilities(emb[:b], emb[b:2 * b]), self._probabilities(emb[2 * b:3 * b], emb[3 * b:])], dim=0)
        labels = torch.cat([torch.ones(b, device=self.device), torch.zeros(b, device=self.device)])
        if lookahead is not None:
            self._prefetch(lookahead[:2])
        bb.update_pe_device(self.table, bn_cap, n_live, src, dst, ts, self.K, presorted, changed=ring.written, mirror=ring.building(),
                            mirror_shard=(self.W, self.rank))
        ring.commit()
        ring.apply_advance()
        return {"loss": F.binary_cross_entropy(predicts, labels), "predicts": predicts}

    def _train_iteration(self, optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead):
        n_glob = src.numel()
        assert n_glob % self.W == 0, "global batch must divide by the world size"
        b = n_glob // self.W
        sl = slice(self.rank * b, (self.rank + 1) * b)
        self.bb.prepare_step()
        owner_counts = self._prefetched_owner_counts(src, dst)      # (before the grouping below consumes the engine's prefetch)
        bn, presorted = self.eng.batch_nodes_and_segments(src, dst)
        out, loss = None, None
        if batch_idx == 0:
            self.table.copy_(initial_pe)
            owner_counts = None
        else:
            started = self._splice_start(bn, batch_idx, owner_counts)
            s_, d_, n_, t_ = src[sl], dst[sl], neg_dst[sl], ts[sl]
            ids3, t3 = torch.cat([s_, d_, n_]), torch.cat([t_, t_, t_])
            fused = self.bb._fused_tail_ok()
            split = self.W > 1 or os.environ.get("LSTEP_FORCE_COLLECTIVES") == "1"
            if split:
                # edge + node channels first: they read no PE row, so their launch overlaps the all-gather of the filtered rows
                x_edge, x_node, _, _, _ = self.bb._gather(None, ids3, t3, self.K, self.G, nat_branch("edge_node"), wide=fused, row_blocks=3)
            rows_mine, leaf, (owner_order, owner_counts) = self._splice_finish(started)
            spliced = SplicedRows(leaf, self.slot_of)
            if split:
                _, _, x_pe, own, _ = self.bb._gather(self.table, ids3, t3, self.K, self.G, nat_branch("pe"), spliced, wide=fused, row_blocks=3)
                emb_p = self.bb._combined_tail(x_edge, x_node, x_pe, own, fused)
            else:
                emb_p = self.bb.combining_pe_raw_feat(self.table, ids3, t3, self.K, self.G, spliced=spliced, padded=True, row_blocks=3)
            emb = emb_p[:, :self.bb.feat_dim]
            pos_src, pos_dst, neg_emb = emb[:b], emb[b:2 * b], emb[2 * b:]
            if self.eng.fused_loss and self.predictor.fused_ok(emb_p):   # predictor + loss terms + their gradient: three launches
                logits = self.predictor.pair_logits(emb_p, b, (0, b, 0, 2 * b))
                loss, lp_loss, pe_loss, predicts = _LinkLoss.apply(logits, leaf, self.table, self.slot_of, ids3, self.eng.pe_weight,
                                                                   self.eng.neg_sample_weight)
            else:
                p_pos = self._probabilities(pos_src, pos_dst)
                p_neg = self._probabilities(pos_src, neg_emb)
                predicts = torch.cat([p_pos, p_neg], dim=0)
                labels = torch.cat([torch.ones_like(p_pos), torch.zeros_like(p_neg)], dim=0)
                lp_loss = F.binary_cross_entropy(predicts, labels)
                e_src = _lookup_rows(self.table, spliced, s_)
                pe_loss = F.mse_loss(e_src, _lookup_rows(self.table, spliced, d_)) - self.eng.neg_sample_weight * F.mse_loss(e_src, _lookup_rows(self.table, spliced, n_))
                loss = (1.0 - self.eng.pe_weight) * lp_loss + self.eng.pe_weight * pe_loss
            out = {"lp_loss": lp_loss.detach(), "pe_loss": pe_loss.detach(), "loss": loss.detach(), "predicts": predicts.detach()}
        if lookahead is not None:
            self._prefetch(lookahead[:2])                  # the next global batch's endpoints, grouped while this one runs
        # update_pe: the all-gather of the phase-2 rows (the largest collective, ~0.7 KB per touched node) stays in flight
        # while the backward pass runs; neither reads what the other writes
        # With a loss to differentiate, update_pe's kernels go to a side stream, issued from this same thread (so every rank still posts
        # its collectives in the same order): they read and write only the PE table, which the backward pass never touches.  The host
        # waits once inside phase 2 (segment counts), i.e. until the forward pass, phase 1 and the neighbour grouping have run; the
        # rest of update_pe (and the all-gather of its rows) then runs underneath the backward pass.  (Enqueueing phase 2 after the
        # backward pass was measured too, 4.9-5.0 ms/step either way at W = 1: with one host thread the step is bound by the ~230
        # launches it issues, not by the GPU.)
        overlap = loss is not None and self.eng.overlap_update
        if overlap:
            main, side = torch.cuda.current_stream(self.device), self.eng._update_stream
            side.wait_stream(main)       # after the forward pass: it reads the table update_pe is about to rewrite
            with torch.cuda.stream(side):
                state = self._update_phase1(bn, src, dst, ts, presorted=presorted, owner_counts=owner_counts, first=batch_idx == 0,
                                            owned_idx=self._owned_idx)
                pending = self._update_phase2(bn, ts, state)
        else:
            pending = self._update_start(bn, src, dst, ts, presorted=presorted, owner_counts=owner_counts, first=batch_idx == 0)
        if loss is None:
            self._update_finish(pending)
            if batch_idx == 0 and initial_pe is not None:
                initial_pe.copy_(self.table)
        if loss is not None:
            optimizer.zero_grad()
            (loss / self.W).backward()                       # global mean = mean of the rank means
            g_rows = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
            # every rank's loss touches every spliced row; each rank needs the summed gradient of the rows it owns.  The reduce-scatter
            # only needs the critical stream's activation backward: the weight-gradient products are still running on the auxiliary
            # stream underneath it, and are joined only where the flat parameter all-reduce needs them
            g_mine = reduce_scatter_var(g_rows[owner_order].contiguous(), owner_counts, self.group)
            if rows_mine.numel():
                rows_mine.backward(g_mine)                   # -> fft_filter / fft_agg through this rank's history shard
            self.bb.join_aux_stream()
            all_reduce_gradients(self._trainable, self.group)
            if overlap:
                with torch.cuda.stream(side):
                    self._update_finish(pending)
                main.wait_stream(side)       # the optimiser may only step once update_pe has read its weights
            else:
                self._update_finish(pending)
            optimizer.step()
            self.slot_of.index_fill_(0, bn, -1)
            # losses reported as global means (one collective for the three scalars)
            v = torch.stack([out["lp_loss"], out["pe_loss"], out["loss"]])
            all_reduce_sum(v, self.group)
            out["lp_loss"], out["pe_loss"], out["loss"] = (v / self.W).unbind(0)
        return out

    # ---- evaluate_model_utils.py:38-142 on a global batch (call under torch.no_grad())
    def eval_iteration(self, batch_idx: int, src, dst, ts, eid, neg_src, neg_dst, lookahead=None):
        if self.device_driven and self._ring.len > 0:
            return self._eval_iteration_dev(batch_idx, src, dst, ts, eid, neg_src, neg_dst, lookahead)
        if self.form == "pull":
            return self._eval_iteration_pull(batch_idx, src, dst, ts, eid, neg_src, neg_dst, lookahead)
        if self.replicated:
            return self._eval_iteration_replicated(batch_idx, src, dst, ts, eid, neg_src, neg_dst, lookahead)
        n_glob = src.numel()
        b = n_glob // self.W
        sl = slice(self.rank * b, (self.rank + 1) * b)
        bn = torch.unique(torch.cat([src, dst]))
        self._splice(bn, batch_idx)
        self.slot_of.index_fill_(0, bn, -1)
        ids = torch.cat([src[sl], dst[sl], neg_src[sl], neg_dst[sl]])
        emb = self.bb.combining_pe_raw_feat(self.table, ids, torch.cat([ts[sl]] * 4), self.K, self.G)
        p_pos = self._probabilities(emb[:b], emb[b:2 * b])
        p_neg = self._probabilities(emb[2 * b:3 * b], emb[3 * b:])
        predicts = torch.cat([p_pos, p_neg], dim=0)
        labels = torch.cat([torch.ones_like(p_pos), torch.zeros_like(p_neg)], dim=0)
        self._update_finish(self._update_start(bn, src, dst, ts))
        return {"loss": F.binary_cross_entropy(predicts, labels), "predicts": predicts}

    # =====================================================================================================================================
    # DEVICE-DRIVEN iteration (round 4).  Same algorithm as the host-sized iterations above, but nothing in it depends on a
    # count the host would have to read: the batch-node list is capacity-sized (2 x global batch, live count on the device), its split by
    # owner is W blocks of C slots (``lstep_owner_partition``: ids, positions, per-owner counts -- all on the device), and every collective
    # moves whole blocks:  all_gather_into_tensor of [C, P] filtered rows -> [W * C, P];  reduce_scatter_tensor of the [W * C, P] gradient
    # (the gradient buffer IS laid out by (owner, slot): slot_of numbers the spliced rows that way, so no re-ordering copy on either side);
    # the flat all-reduce;  "pull": all_gather of the [C, P] phase-1 rows, all_to_all_single of [W, Cp] request ids and of [W, Cp, P] rows.
    # Unused slots carry node 0 / garbage rows that no consumer reads (every consumer takes the per-owner count on the device).
    # A block that is too small sets a sticky device flag which the host polls WITHOUT waiting (``check_capacity``): the step that
    # overflowed is invalid and the next call raises ``LstepCapacityError`` (capacity: LSTEP_DIST_SLACK, default 1.25 x an even split;
    # ids are spread over owners by id % W, so a block is off an even split by ~1 / sqrt(U / W)).
    # An iteration is then a fixed launch sequence: over RCCL it is captured ONCE, collectives included, and replayed (``GraphedDistStep``).
    # =====================================================================================================================================
    def _owner_capacity(self, cap: int) -> int:
        """Slots per owner block for a capacity-sized list of ``cap`` ids."""
        if self.W == 1:
            return cap
        slack = float(os.environ.get("LSTEP_DIST_SLACK", "1.25"))
        want = int(-(-cap // self.W) * slack) + 64
        return min(cap, -(-want // 64) * 64)

    def _partition(self, bn_cap: torch.Tensor, n_live: torch.Tensor):
        """``lstep_owner_partition`` of the batch-node list: (ids [W * C] int64, positions in bn [W * C] int32, counts [W] int32, C)."""
        from . import _native as nat
        lib = nat.load_library()
        cap, W, dev = bn_cap.numel(), self.W, self.device
        C = self._owner_capacity(cap)
        ids = torch.empty(W * C, dtype=torch.int64, device=dev)
        pos = torch.empty(W * C, dtype=torch.int32, device=dev)
        counts = torch.empty(W, dtype=torch.int32, device=dev)
        ws = nat._workspace(dev, int(lib.lstep_owner_partition_workspace(cap, W)))
        with torch.cuda.device(dev):
            nat.check(lib.lstep_owner_partition(nat.ptr(bn_cap), cap, nat.ptr(n_live), W, C, nat.ptr(ws), ws.numel(), nat.ptr(ids), nat.ptr(pos),
                                                nat.ptr(counts), nat.ptr(self._overflow), nat.current_stream()))
        return ids, pos, counts, C

    def check_capacity(self, wait: bool = False):
        """Raise if a fixed-capacity block overflowed in an EARLIER iteration.  The flag travels to the host asynchronously; without ``wait``
        this never blocks (an overflow is reported one or two iterations late -- the results since then are invalid either way)."""
        if not self._overflow.is_cuda:
            bad = bool(self._overflow.item())
        else:
            if wait:
                bad = bool(self._overflow.item())
            else:
                bad = False
                pend = self._overflow_poll
                if pend is not None and pend[1].query():
                    bad = bool(pend[0][0])
                    self._overflow_poll = pend = None
                if pend is None and not torch.cuda.is_current_stream_capturing():
                    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                    host.copy_(self._overflow, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    self._overflow_poll = (host, ev)
        if bad:
            raise LstepCapacityError("a fixed-capacity owner block of the device-driven multi-GPU iteration overflowed (node ids are very unevenly "
                                     "spread over id % world): results since then are invalid; raise LSTEP_DIST_SLACK (default 1.25) or set "
                                     "LSTEP_DIST_HOST_COUNTS=1 for the host-sized iteration")

    def _log(self, name: str, nbytes: int):
        """bench.py's ``comm`` object: a pair of timed events around an eagerly issued collective (never inside a capture)."""
        if self.comm_log is None or torch.cuda.is_current_stream_capturing() or not self.table.is_cuda:
            return contextlib.nullcontext()
        log = self.comm_log

        @contextlib.contextmanager
        def timed():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            yield
            e1.record()
            log.append((name, int(nbytes), e0, e1))
        return timed()

    def _scatter_owner_rows(self, rows_all: torch.Tensor, part, number: bool):
        """Block p, slot i < counts[p] of the gathered rows -> table[ids[p, i]]; ``number``: slot_of[id] = p * C + i as well."""
        from . import _native as nat
        ids, _, counts, C = part
        with torch.cuda.device(self.device):
            nat.check(nat.load_library().lstep_scatter_owner_rows(nat.ptr(rows_all), int(rows_all.stride(0)), nat.ptr(ids), nat.ptr(counts), self.W, C,
                                                                  nat.ptr(self.table), self.bb.pe_dim, nat.ptr(self.slot_of) if number else None,
                                                                  nat.current_stream()))

    def _splice_start_dev(self, part, batch_idx: int):
        """Filter the history of the batch nodes this rank owns (its block of the partition, count on the device) and put the all-gather of
        the [C, P] block IN FLIGHT."""
        ids, _, counts, C = part
        r = self.rank
        self._wait_snapshot()
        ring = self._ring
        ring.wait_window()
        mine_local = torch.div(ids[r * C:(r + 1) * C], self.W, rounding_mode="floor")      # local ring rows (dead slots: node 0 -> row 0)
        rows_mine = self.bb.filter_history(ring.buf, ring.geom(), mine_local, batch_idx, mask=ring.mask, oldest=ring.oldest,
                                           live=counts[r:r + 1], ring=ring.window_ref())
        with self._log("all_gather filtered rows", self.W * C * self.bb.pe_dim * 4):
            pending = PendingGather(rows_mine.detach(), self.group, counts=[C] * self.W)
            if self.comm_log is not None:
                pending.wait()                        # (being timed: not left in flight)
        return rows_mine, pending

    def _splice_finish_dev(self, started, part):
        rows_mine, pending = started
        gathered = pending.wait()                     # [W * C, P], block p = rank p's filtered rows
        self._scatter_owner_rows(gathered, part, number=True)
        return rows_mine, gathered.detach().requires_grad_(True)

    def _forward_dev(self, part, batch_idx, ids3, t3, b, pull=None):
        """FFT splice (owner-sharded filter + fixed-capacity all-gather) and this rank's slice through gather, dense tail, predictor, loss."""
        started = self._splice_start_dev(part, batch_idx)
        fused = self.bb._fused_tail_ok()
        # edge + node channels first: they read no PE row, so their launch overlaps the all-gather of the filtered rows
        x_edge, x_node, _, _, _ = self.bb._gather(None, ids3, t3, self.K, self.G, nat_branch("edge_node"), wide=fused, row_blocks=3)
        if pull is not None:
            pull.wait()      # (rows of this batch's nodes were fetched BEFORE their splice: the all-gathered spliced rows overwrite them next)
        rows_mine, leaf = self._splice_finish_dev(started, part)
        spliced = SplicedRows(leaf, self.slot_of)
        _, _, x_pe, own, _ = self.bb._gather(self.table, ids3, t3, self.K, self.G, nat_branch("pe"), spliced, wide=fused, row_blocks=3)
        emb_p = self.bb._combined_tail(x_edge, x_node, x_pe, own, fused)
        logits = self.predictor.pair_logits(emb_p, b, (0, b, 0, 2 * b))
        loss, lp_loss, pe_loss, predicts = _LinkLoss.apply(logits, leaf, self.table, self.slot_of, ids3, self.eng.pe_weight,
                                                           self.eng.neg_sample_weight)
        out = {"lp_loss": lp_loss.detach(), "pe_loss": pe_loss.detach(), "loss": loss.detach(), "predicts": predicts.detach()}
        return out, loss, rows_mine, leaf

    def _share_phase1_rows_dev(self, part):
        """``update_pe_device(after_phase1=...)`` with fixed blocks: every owner's [C, P] block of freshly written phase-1 rows is all-gathered
        and scattered into every table (north_star's "all-gather of updated positional encodings")."""
        ids, _, _, C = part
        r = self.rank

        def share(ids1):
            if _skip_single(self.W):
                return
            mine = self.table.index_select(0, ids[r * C:(r + 1) * C])
            with self._log("all_gather phase-1 rows", self.W * C * self.bb.pe_dim * 4):
                rows_all, _ = all_gather_var(mine, self.group, counts=[C] * self.W)
            self._scatter_owner_rows(rows_all, part, number=False)
        return share

    def _reduce_gradients_dev(self, leaf, rows_mine, part, out):
        """Backward tail shared by both forms: reduce-scatter of the spliced rows' gradient by owner block, the history-filter backward on this
        rank's shard, the flat all-reduce (with the three loss scalars), everything scaled to the mean over the ranks."""
        C, W, P = part[3], self.W, self.bb.pe_dim
        g_rows = leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)
        with self._log("reduce_scatter row gradient", W * C * P * 4):
            g_mine = reduce_scatter_var(g_rows, [C] * W, self.group)
        # g_mine is the SUM over the ranks of their local gradients (each seeded with 1): the filter's parameter gradients computed from it are
        # this OWNER's share of W x the global-mean gradient; the bucket below adds the owners' shares and scales everything by 1 / W once
        rows_mine.backward(g_mine)                       # -> fft_filter / fft_agg through this rank's history shard (dead slots: zero gradient)
        self.bb.join_aux_stream()
        extra = torch.stack([out["lp_loss"], out["pe_loss"], out["loss"]])
        with self._log("all_reduce parameter gradients", sum(p.numel() * (2 if p.is_complex() else 1) for p in self._trainable) * 4):
            v = all_reduce_gradients(self._trainable, self.group, extra=extra, scale=(1.0 / W) if W > 1 else None)
        return v

    def _batch_dev(self, src, dst, neg_dst, ts):
        """(global grouping keys / float32(max t), this rank's gather rows and times, B per rank)."""
        eng, W, r = self.eng, self.W, self.rank
        b = src.numel() // W
        prep = eng.prepare_batch(src, dst, neg_dst, ts)
        if W == 1:
            return prep, prep[0], prep[1], b
        sl = slice(r * b, (r + 1) * b)
        loc = eng.prepare_batch(src[sl], dst[sl], neg_dst[sl], ts[sl])
        return prep, loc[0], loc[1], b

    def _train_iteration_device_driven(self, optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead):
        from .model import drain_dead_graphs
        drain_dead_graphs()
        self.check_capacity()
        assert src.numel() % self.W == 0, "global batch must divide by the world size"
        ring = self._ring
        if self._ring_generation != ring.generation:
            self.drop_captured_iterations()
            self._ring_generation = ring.generation
        if self._graph_ready(optimizer, batch_idx, src, ts):
            gs = self._graphed.get(src.numel())
            if gs is None or gs.optimizer is not optimizer:
                if gs is not None:
                    gs.close()
                gs = self._graphed[src.numel()] = GraphedDistStep(self, optimizer, src.numel())
            return gs.step(batch_idx, src, dst, ts, eid, neg_dst, lookahead)
        if ring.len == ring.T and batch_idx > 0:
            self._steady_steps += 1
        with self.eng.aux_streams():
            return self._train_iteration_dev(optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead)

    _ring_generation = 0

    def _graph_ready(self, optimizer, batch_idx, src, ts) -> bool:
        from .optim import FusedAdam
        ring = self._ring
        return (self.use_step_graph and batch_idx > 0 and ring.len == ring.T and isinstance(optimizer, FusedAdam) and self._steady_steps >= 2
                and self.eng.overlap_update and src.dtype == torch.int64 and ts.dtype == torch.float64)

    def drop_captured_iterations(self):
        for gs in self._graphed.values():
            gs.close()
        self._graphed, self._steady_steps = {}, 0

    def close(self):
        self.drop_captured_iterations()
        self.bb.close()

    def _train_iteration_dev(self, optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead):
        """train:204-311 on a global batch of W * B edges, no host synchronisation anywhere (module comment above)."""
        eng, bb, ring = self.eng, self.bb, self._ring
        W, rank = self.W, self.rank
        pull_form = self.form == "pull"
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            ring.early_advance()             # the previous iteration's slide: beside this iteration
        else:
            ring.apply_advance()             # (a slide left pending by a replayed iteration)
        bb.prepare_step()
        prep, ids3, t3, b = self._batch_dev(src, dst, neg_dst, ts)
        bn_cap, n_live, presorted = eng.batch_nodes_device(src, dst, keys=prep[2])
        part = self._partition(bn_cap, n_live)
        out = loss = None
        if batch_idx == 0:
            self.table.copy_(initial_pe)         # every row valid on every rank
            self._pending_pull = None
            ring.begin_slot(all_changed=True)
        else:
            pull = None
            if pull_form and not capturing:
                # (a captured iteration finds the rows of ITS gather in the table: the previous replay fetched them from its look-ahead
                # buffers, or ``GraphedDistStep.step`` fetched them launch by launch; every replay ends with all its streams joined)
                pull = self._take_pull(src, dst, ts, neg_dst) or self._pull_now_dev((src, dst, neg_dst), ts)
            out, loss, rows_mine, leaf = self._forward_dev(part, batch_idx, ids3, t3, b, pull=pull)
        nxt = None
        ps = self._pull_stream
        ahead = lookahead if (pull_form and lookahead is not None and len(lookahead) >= 4) else None
        # A captured iteration issues the pull's two all-to-all exchanges from the CAPTURING stream, last: torch 2.10 / RCCL 2.26 fault when
        # all_to_all_single is captured from a side stream that joined the capture (all_gather / reduce_scatter / all_reduce are fine there:
        # tools/rccl_capture_probe.py, profiles/r04_rccl_capture_probe.txt).  Launch by launch the requests go out beside the forward pass
        # and the rows travel underneath the backward pass, on the pull's own stream and communicator.
        pull_last = capturing and ahead is not None
        if ahead is not None:       # the next batch's requests: independent of everything this iteration computes
            if ps is not None:
                ps.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                # (captured: the request lists are built here, beside the forward pass; their exchange goes out from the capturing stream, last)
                nxt = RowPullDev(self, *self._slice_rows((ahead[0], ahead[1], ahead[3]), ahead[2]), key=TensorsKey(ahead[0], ahead[1], ahead[2], ahead[3]),
                                 defer_exchange=pull_last)

        def update_and_append():
            base, ref = ring.building_ref()
            kw = dict(changed=ring.written, mirror=base if ref is not None else ring.building(), mirror_ring=ref, mirror_shard=(W, rank),
                      now32=prep[3])
            if pull_form:
                C = part[3]
                kw.update(owner=(W, rank), owned_idx=part[1][rank * C:(rank + 1) * C].long(), owned_live=part[2][rank:rank + 1],
                          after_phase1=self._share_phase1_rows_dev(part))
            bb.update_pe_device(self.table, bn_cap, n_live, src, dst, ts, self.K, presorted, **kw)
            if batch_idx == 0 and initial_pe is not None:
                if pull_form:
                    self.sync_full_table()
                initial_pe.copy_(self.table)
            ring.commit()

        def fetch_next(after=None):
            """Serve and receive the next gather's rows on the pull stream, behind ``after`` (update_pe's end) or the current stream."""
            if ps is not None:
                if after is not None:
                    ps.wait_event(after)
                else:
                    ps.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                self._poison_foreign_rows()
                nxt.fetch()
            self._pending_pull = nxt

        if loss is None:
            update_and_append()
            ring.apply_advance()
            if nxt is not None:
                fetch_next()
            ring.tick()
            return out
        main, side = torch.cuda.current_stream(self.device), eng._update_stream
        side.wait_stream(main)           # after the forward pass: it reads the table update_pe is about to rewrite
        with torch.cuda.stream(side):
            update_and_append()
            updated = torch.cuda.Event()
            updated.record()
        optimizer.zero_grad()
        _backward_unit(loss)
        v = self._reduce_gradients_dev(leaf, rows_mine, part, out)
        ring.apply_advance()             # the backward pass is enqueued: the window's oldest snapshot may move on behind it
        if nxt is not None and not pull_last:
            # the rows of the NEXT gather: behind update_pe (the owners' rows are final there).  Launch by launch with a communicator of its
            # own the exchange runs underneath the backward pass; on the main communicator (captured iterations) it is issued LAST, so the
            # backward pass's reduce-scatter / all-reduce never queue behind it
            fetch_next(updated)
        main.wait_event(updated)         # the optimiser may only step once update_pe has read its weights; the ring shard is appended
        if pull_last:
            main.wait_stream(ps)         # the request blocks are ready (built on the pull stream beside the forward pass)
            nxt.send_requests()
            self._poison_foreign_rows()
            nxt.fetch()
            self._pending_pull = nxt
        optimizer.step()
        self.slot_of.index_fill_(0, bn_cap, -1)      # (the dead tail is node 0, whose entry is -1 anyway)
        ring.tick()
        out["lp_loss"], out["pe_loss"], out["loss"] = v.unbind(0)       # global means (they travelled with the gradient bucket)
        return out

    def _pull_now_dev(self, blocks, ts):
        """No look-ahead had the rows fetched: request and fetch on the current stream (no host wait either: fixed blocks)."""
        pull = RowPullDev(self, *self._slice_rows(blocks, ts))
        self._poison_foreign_rows()
        pull.fetch()
        return pull

    def _eval_iteration_dev(self, batch_idx, src, dst, ts, eid, neg_src, neg_dst, lookahead):
        """evaluate_model_utils.py:38-142 on a global batch, device-driven (call under torch.no_grad())."""
        self.check_capacity()
        eng, bb, ring = self.eng, self.bb, self._ring
        W, rank = self.W, self.rank
        pull_form = self.form == "pull"
        b = src.numel() // W
        ring.apply_advance()
        bn_cap, n_live, presorted = eng.batch_nodes_device(src, dst)
        part = self._partition(bn_cap, n_live)
        pull = None
        if pull_form:
            pull = self._take_pull(src, dst, ts, neg_src, neg_dst) or self._pull_now_dev((src, dst, neg_src, neg_dst), ts)
        started = self._splice_start_dev(part, batch_idx)
        if pull is not None:
            pull.wait()
        self._splice_finish_dev(started, part)
        self.slot_of.index_fill_(0, bn_cap, -1)
        ids, t4 = self._slice_rows((src, dst, neg_src, neg_dst), ts)
        emb_p = bb.combining_pe_raw_feat(self.table, ids, t4, self.K, self.G, padded=True, row_blocks=4)
        if self.predictor.fused_ok(emb_p):
            predicts = self.predictor.pair_logits(emb_p, b, (0, b, 2 * b, 3 * b)).sigmoid().clamp(0, 1)
        else:
            emb = emb_p[:, :bb.feat_dim]
            predicts = torch.cat([self._probabilities(emb[:b], emb[b:2 * b]), self._probabilities(emb[2 * b:3 * b], emb[3 * b:])], dim=0)
        labels = torch.cat([torch.ones(b, device=self.device), torch.zeros(b, device=self.device)])
        nxt = None
        ps = self._pull_stream
        if pull_form and lookahead is not None and len(lookahead) >= 5:
            if ps is not None:
                ps.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                nxt = RowPullDev(self, *self._slice_rows((lookahead[0], lookahead[1], lookahead[3], lookahead[4]), lookahead[2]),
                                 key=TensorsKey(*lookahead[:5]))
        base, ref = ring.building_ref()
        kw = dict(changed=ring.written, mirror=base if ref is not None else ring.building(), mirror_ring=ref, mirror_shard=(W, rank))
        if pull_form:
            C = part[3]
            kw.update(owner=(W, rank), owned_idx=part[1][rank * C:(rank + 1) * C].long(), owned_live=part[2][rank:rank + 1],
                      after_phase1=self._share_phase1_rows_dev(part))
        bb.update_pe_device(self.table, bn_cap, n_live, src, dst, ts, self.K, presorted, **kw)
        ring.commit()
        ring.apply_advance()
        ring.tick()
        if nxt is not None:
            if ps is not None:
                ps.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(ps) if ps is not None else contextlib.nullcontext():
                self._poison_foreign_rows()
                nxt.fetch()
            self._pending_pull = nxt
        return {"loss": F.binary_cross_entropy(predicts, labels), "predicts": predicts}


class RowPullDev:
    """``RowPull`` with fixed blocks in BOTH directions (device-driven: nothing here reads a count on the host): ``request`` sends every owner a
    block of ``Cp`` id slots (-1 = unused), ``fetch`` has every owner copy the rows it was asked for into a [W, Cp, P] buffer
    (``lstep_rows_by_id``), ONE all_to_all_single of equal blocks moves them, and the requester writes the rows of its own id blocks into
    its table.  A list that does not fit its block sets the engine's sticky overflow flag (``DistributedLstep.check_capacity``)."""

    def __init__(self, dl: "DistributedLstep", ids: torch.Tensor, times: torch.Tensor, key: TensorsKey = None, defer_exchange: bool = False):
        """``defer_exchange``: only build the request blocks (kernels: they may run on a side stream of a capture); the caller issues
        ``send_requests()`` later, from the stream the exchange must be issued on."""
        from . import _native as nat
        self.dl, self.key = dl, key
        W, rank, dev, rows = dl.W, dl.rank, dl.device, dl.num_rows
        nbr = dl.bb.neighbor_sampler.sample_device(ids, times, dl.K)[0]
        lib = nat.load_library()
        n = nbr.numel() + ids.numel() + 1
        sentinel = W * rows
        keys = torch.empty(n, dtype=torch.int32, device=dev)
        ids = ids.contiguous()
        with torch.cuda.device(dev):
            nat.check(lib.lstep_pull_keys(nat.ptr(nbr), nbr.numel(), nat.ptr(ids), ids.numel(), W, rank, rows, nat.ptr(keys), nat.current_stream()))
        _, _, _, uniq, summary = nat.group_by_key(keys, max(1, int(sentinel + 1).bit_length()), sentinel, wait=None)
        C = self.C = dl._pull_capacity_dev(n)
        self.req = torch.empty((W, C), dtype=torch.int32, device=dev)
        cnt = torch.empty(W, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_pull_blocks(nat.ptr(uniq), nat.ptr(summary), W, rows, C, nat.ptr(self.req), nat.ptr(cnt), nat.current_stream()))
        dl._overflow.bitwise_or_((cnt > C).any().to(torch.int32))
        self.cnt_max = cnt.max().reshape(1)            # (kept: a replayed graph rewrites it in place, GraphedDistStep reads it back)
        dl._note_pull_counts(self.cnt_max)
        self.done, self.asked = None, None
        if not defer_exchange:
            self.send_requests()

    def send_requests(self):
        dl, W, C = self.dl, self.dl.W, self.C
        with dl._log("all_to_all pull requests", W * C * 4):
            self.asked = exchange_rows(self.req.reshape(W * C), [C] * W, [C] * W, dl.pull_group, async_op=True)     # block p = what rank p wants from me
            if dl.comm_log is not None:
                self.asked.wait()                     # (being timed: not left in flight)

    def fetch(self):
        """Serve and receive (call when the owned rows are final on the current stream)."""
        from . import _native as nat
        dl = self.dl
        W, C, P, dev = dl.W, self.C, dl.bb.pe_dim, dl.device
        lib = nat.load_library()
        asked = self.asked.wait().contiguous()
        rows_out = torch.empty((W * C, P), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_rows_by_id(nat.ptr(asked), W * C, nat.ptr(dl.table), P, nat.ptr(rows_out), 0, nat.current_stream()))
        with dl._log("all_to_all pulled rows", W * C * P * 4):
            rows_in = exchange_rows(rows_out, [C] * W, [C] * W, dl.pull_group, async_op=True).wait().contiguous()
        req = self.req.reshape(W * C)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_rows_by_id(nat.ptr(req), W * C, nat.ptr(dl.table), P, nat.ptr(rows_in), 1, nat.current_stream()))
        if dl.table.is_cuda:
            self.done = torch.cuda.Event()
            self.done.record()
        self._keep = (asked, rows_out, rows_in, req)
        self.asked = None

    def wait(self):
        if self.done is not None:
            torch.cuda.current_stream(self.dl.device).wait_event(self.done)


class GraphedDistStep:
    """One steady-state training iteration of ``DistributedLstep`` (device-driven form) -- FFT splice with its all-gather, this rank's slice
    through gather / tail / predictor / loss, update_pe, the backward pass with its reduce-scatter and all-reduce, Adam, the ring tick --
    captured ONCE as a HIP graph, RCCL collectives included (torch's NCCL process group records them into the capture; verified on this
    torch / RCCL by tools/rccl_capture_probe.py), and replayed per batch: per step the host copies the batch (and, "pull", the look-ahead
    batch) into fixed buffers and launches the graph -- like ``engine.GraphedTrainStep`` on one GPU."""

    def __init__(self, dl: DistributedLstep, optimizer, batch: int):
        dev = dl.device
        self.dl, self.optimizer, self.B = dl, optimizer, int(batch)
        i64 = lambda: torch.zeros(self.B, dtype=torch.int64, device=dev)  # noqa: E731
        f64 = lambda: torch.zeros(self.B, dtype=torch.float64, device=dev)  # noqa: E731
        self.cur = (i64(), i64(), f64(), i64(), i64())          # src, dst, ts, eid, neg
        self.nxt = (i64(), i64(), f64(), i64())                 # src, dst, ts, neg of the look-ahead batch ("pull")
        self.graph, self.out, self.replays, self.hyper = None, None, 0, None
        self._ahead_key = None
        self._pull_cap, self._cnt_max = None, None      # ("pull") block size the captured requests use / their largest count, on the device
        self.pull_graph, self._pull_now_cap, self._pull_now_keep, self.pull_now_replays = None, None, None, 0    # (``_pull_now_graphed``)

    def _hyper(self):
        o = self.optimizer
        return (o.lr, o.weight_decay, tuple(o.betas), o.eps)

    def close(self):
        from .model import drain_dead_graphs, retire_graph
        for g in (self.graph, self.pull_graph):
            if g is not None:
                retire_graph(g)
        self.graph, self.out, self.pull_graph, self._pull_now_keep = None, None, None, None
        drain_dead_graphs()

    def step(self, batch_idx, src, dst, ts, eid, neg_dst, lookahead):
        dl = self.dl
        pull = dl.form == "pull"
        fresh = True
        if pull:
            # the rows of THIS gather were requested and fetched by the previous replay from its look-ahead buffers: valid only if that
            # look-ahead named exactly this batch (else: ``_pull_now_graphed`` below, once the batch sits in the fixed buffers)
            fresh = self._ahead_key is not None and self._ahead_key.matches(src, dst, ts, neg_dst)
            if lookahead is None or len(lookahead) < 4:
                lookahead = (src, dst, ts, neg_dst)          # (no look-ahead: the rows fetched for "the next batch" are simply not used)
                self._ahead_key = None
            else:
                self._ahead_key = TensorsKey(lookahead[0], lookahead[1], lookahead[2], lookahead[3])
            torch._foreach_copy_([self.nxt[0], self.nxt[1], self.nxt[3]], [lookahead[0], lookahead[1], lookahead[3]])
            self.nxt[2].copy_(lookahead[2])
        torch._foreach_copy_([self.cur[0], self.cur[1], self.cur[3], self.cur[4]], [src, dst, eid, neg_dst])
        self.cur[2].copy_(ts)
        if pull and not fresh:
            pend = dl._take_pull(src, dst, ts, neg_dst)        # (fetched by a launch-by-launch iteration from ITS look-ahead: an event wait)
            if pend is not None:
                pend.wait()
            else:
                self._pull_now_graphed()
        dl.eng.__dict__.pop("_prefetched_group", None)
        dl._pending_pull = None
        if self.graph is not None and self.hyper != self._hyper():
            self.close()
        if self.graph is not None and pull and self._pull_cap is not None:
            # the request lists have grown towards the blocks the graph was captured with: capture again with larger ones BEFORE one overflows
            seen = dl._pull_seen_max()
            if seen is not None and seen * 1.1 > self._pull_cap:
                self.close()
        if self.graph is None:
            self._capture(batch_idx)
        else:
            self.graph.replay()
            self.replays += 1
            dl._ring.replay_tick()
            if pull and self._cnt_max is not None:
                dl._note_pull_counts(self._cnt_max)
        return self.out

    def _pull_now_graphed(self):
        """The rows this batch's gather reads, requested and fetched ON THE SPOT (the previous step's look-ahead did not name this batch: the
        first replayed step, an epoch boundary, a switch between training and evaluation) -- as a small captured graph of its own over the
        fixed batch buffers, replayed in front of the step's graph.  Once a step has been captured NO collective of this object is issued
        launch by launch any more: eager collectives behind a capture are what ended a process in round 4 (the process group's watchdog
        thread queried an event of an eager collective that had last been recorded in a capturing stream -- torch caches and re-uses the
        HIP events of its collective Work objects), and what ``bench.py`` had to route around."""
        from .model import _no_gc, new_graph, retire_graph
        dl = self.dl
        n_rows = 3 * (self.B // dl.W)
        cap = dl._pull_capacity_dev(n_rows * (dl.K + 1) + 1)      # (the size RowPullDev sizes its blocks from)
        if self.pull_graph is not None and self._pull_now_cap != cap:
            retire_graph(self.pull_graph)
            self.pull_graph = None
        if self.pull_graph is None:
            src, dst, ts, _, neg = self.cur
            quiesce_collectives(dl.device)      # (the pull stream joins this capture and may have carried launch-by-launch pulls)
            g = new_graph(self)
            with _no_gc(), self._capture_streams(), torch.cuda.graph(g, stream=nat.role_stream(dl.device, "capture"), capture_error_mode="thread_local"):
                p = dl._pull_now_dev((src, dst, neg), ts)
                p.wait()
                self._pull_now_keep = p          # (its tensors are the graph's: request blocks, served rows, the largest per-owner count)
            self.pull_graph, self._pull_now_cap = g, cap
        self.pull_graph.replay()
        self.pull_now_replays += 1
        if self._pull_now_keep is not None and self._pull_now_keep.cnt_max is not None:
            dl._note_pull_counts(self._pull_now_keep.cnt_max)

    @contextlib.contextmanager
    def _capture_streams(self):
        """For the duration of a capture the side branches that hold collectives -- update_pe's stream, the pull's -- run on streams of their
        own that no launch-by-launch iteration ever uses.  ``quiesce_collectives`` already keeps the NCCL watchdog from polling an event whose
        stream has started to capture (DESIGN.md section 10 iii); with the branches on capture-only streams there is no such event to poll,
        however late the watchdog's sweep comes on a loaded host."""
        dl = self.dl
        saved = (dl.eng._update_stream, dl._pull_stream)
        dl.eng._update_stream = nat.role_stream(dl.device, "update-captured")
        if saved[1] is not None:
            dl._pull_stream = nat.role_stream(dl.device, "pull-captured", priority=-1 if os.environ.get("LSTEP_PULL_PRIORITY", "1") == "1" else 0)
        try:
            yield
        finally:
            dl.eng._update_stream, dl._pull_stream = saved

    def _capture(self, batch_idx):
        from .model import _aux_stream, _no_gc, new_graph
        dl = self.dl
        eng, ring = dl.eng, dl._ring
        quiesce_collectives(dl.device)      # (the update / pull streams join this capture; the iterations before it put collectives on them)
        ring._advanced = [None, None]
        if ring.dev_start is None:
            ring.position_on_device()
        graph = new_graph(self)
        self.hyper = self._hyper()
        src, dst, ts, eid, neg = self.cur
        ahead = (self.nxt[0], self.nxt[1], self.nxt[2], self.nxt[3]) if dl.form == "pull" else None
        # thread_local: the process group's watchdog thread polls the events of collectives issued launch by launch BEFORE the capture
        # (hipEventQuery); in the default "global" mode that call from another thread is an error that kills the capture
        # ("operation not permitted when stream is capturing" out of ProcessGroupNCCL's watchdog -- seen once the bench's eager iterations
        # ran right in front of the capture).  The autograd thread's launches are captured in either mode.  (thread_local does NOT cover
        # the watchdog querying an event whose STREAM is now capturing: quiesce_collectives above.)
        with _no_gc(), self._capture_streams(), torch.cuda.graph(graph, stream=nat.role_stream(dl.device, "capture"), capture_error_mode="thread_local"):
            with eng.aux_streams():
                self.out = dl._train_iteration_dev(self.optimizer, batch_idx, src, dst, ts, eid, neg, None, ahead)
            main = torch.cuda.current_stream(dl.device)
            if eng.use_aux:
                main.wait_stream(_aux_stream(dl.device))
            main.wait_stream(eng._update_stream)
            if dl._pull_stream is not None and dl.form == "pull":
                main.wait_stream(dl._pull_stream)
        if dl._pending_pull is not None:
            self._pull_cap, self._cnt_max = dl._pending_pull.C, dl._pending_pull.cnt_max
        dl._pending_pull = None          # (the captured iteration's pull object belongs to the graph: replays fetch into the table directly)
        self.graph = graph
        graph.replay()
