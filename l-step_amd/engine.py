"""Device-resident driver of the L-STEP per-batch protocol (the fast harness around :class:`lstep_amd.model.LSTEP`).

It executes the same per-batch contract as the reference loops (``train_LSTEP_link_prediction.py:204-311``,
``evaluate_model_utils.py:38-142``; restated for reference-shaped tensors in ``lstep_amd/protocol.py``) but keeps all
state in HBM and never builds the reference's per-batch dense temporaries:

* the PE history is a slot-major ring ``[T+2, N+1, P]`` (one snapshot = one contiguous block; a spare slot receives the
  next snapshot, so the T-snapshot window the FFT filter and its backward read is never overwritten) instead of
  ``torch.cat`` + ``.cpu()`` of the whole ``[N+1, t, P]`` tensor every batch (``train:205,301,306``); a second spare
  slot lets the next batch's base copy (``train:229``) run on a copy stream underneath the backward pass;
* the "current PE" (``clone(last snapshot)`` with the FFT-filtered batch rows spliced in, ``train:229-230``) is
  materialised directly in the spare ring slot, where ``update_pe`` then turns it into the next snapshot in place;
* gradients w.r.t. the current PE exist only for the spliced rows: the gather backward scatters into ``[U, P]`` through an
  int32 ``slot_of`` map instead of allocating dense ``[N+1, P]`` gradients (three per batch in the reference);
* the three (train) / four (eval) ``combining_pe_raw_feat`` calls of a batch are one launch over ``3B`` / ``4B`` rows.

Edge streams live on the device (``EdgeStream``); a batch is a slice, no host round trip.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from . import _native as nat
from .model import LSTEP, MergeLayer, SplicedRows


@dataclass
class EdgeStream:
    """Chronological edge arrays on the device (src/dst/eid int64, ts float64)."""
    src: torch.Tensor
    dst: torch.Tensor
    ts: torch.Tensor
    eid: torch.Tensor

    @classmethod
    def from_numpy(cls, src, dst, ts, eid, device="cuda"):
        f = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)  # noqa: E731
        return cls(f(src, np.int64), f(dst, np.int64), f(ts, np.float64), f(eid, np.int64))

    def batch(self, lo: int, hi: int):
        return self.src[lo:hi], self.dst[lo:hi], self.ts[lo:hi], self.eid[lo:hi]


class HistoryRing:
    """Last-T PE snapshots, slot-major ``[T + 2, rows, P]``, with per-row change bits.

    Every snapshot of the reference is a clone of the one before it plus the rows its batch wrote (train:229,301): bit ``slot`` of row
    n says "node n's row in physical slot ``slot`` differs from the slot before", and the FFT filter reads one row per run of equal
    snapshots (``lstep_history_filter_runs_*``).  Whoever writes rows of the snapshot being built must call ``mark``.

    Two ways of keeping the slots:
    * **sparse** (the single-GPU engine's default): the current PE lives in ``table`` and is updated in place; only the rows the batch
      wrote (~21 % of the table on the c4 workload) reach the batch's slot -- ``update_pe``'s kernels write them there themselves
      (``building()``, the ``mirror`` of ``lstep_update_rows``), ``commit`` copies what is left (``lstep_copy_rows``) -- because the run
      kernels never read any other row of a slot; ``oldest`` holds the window's oldest snapshot, moved on by
      ``lstep_history_advance_oldest`` when the window slides (``apply_advance``: behind the pending backward pass, which still reads the
      old window).  ~0.5 GB of row traffic per batch instead of the 2.8 GB of a whole-table clone.
    * **clones** (``sparse=False``: the owner-sharded rings of ``lstep_amd.parallel``, ``LSTEP_CLONE_HISTORY=1``, and whenever there is
      no mask -- ``LSTEP_DENSE_HISTORY=1`` or more than 128 slots): every slot is a full snapshot; the snapshot being built lives in a
      spare slot, and a second spare slot lets the next clone be prefetched on a copy stream under the backward pass without touching
      the window the pending FFT-filter backward still reads."""

    def __init__(self, num_rows: int, pe_dim: int, num_fft_batches: int, device="cuda", sparse: bool = False):
        self.T = int(num_fft_batches)
        self.S = self.T + 2
        self.rows, self.P = int(num_rows), int(pe_dim)
        self.buf = torch.zeros((self.S, self.rows, self.P), dtype=torch.float32, device=device)
        self.start = 0   # physical slot of the oldest snapshot in the window
        self.len = 0     # snapshots in the window (<= T)
        self._copy_stream = nat.role_stream(device, "ring-copy")     # (a stream of our own, never one of PyTorch's pool: _native.role_stream)
        self._prefetched = None  # (slot index, event) of a base copy issued ahead of time
        self.words = (self.S + 31) // 32
        on = torch.device(device).type == "cuda" and self.words <= 4 and os.environ.get("LSTEP_DENSE_HISTORY") != "1"
        self.mask = torch.zeros((self.rows, self.words), dtype=torch.int32, device=device) if on else None
        self.sparse = bool(sparse) and self.mask is not None and os.environ.get("LSTEP_CLONE_HISTORY") != "1"
        if self.sparse:
            self.table = torch.zeros((self.rows, self.P), dtype=torch.float32, device=device)    # the current PE (= newest snapshot)
            self.oldest = torch.zeros((self.rows, self.P), dtype=torch.float32, device=device)   # the window's oldest snapshot
            self._row0 = torch.zeros(1, dtype=torch.int64, device=device)
            self._written, self._all_written = [], False   # ids marked for the snapshot being built
            self._advance = None
            self._advanced = [None, None]            # events of the last two ``apply_advance`` calls
            self.advance_stream = None               # stream ``apply_advance`` runs on (default: the ring's copy stream)
        else:
            self.oldest = None
        # Optional device-resident ring position (``position_on_device``): the kernels then read the slot they work on from this word
        # (``lstep_ring_ref_t``) instead of receiving it as a launch argument, so the launch sequence of an iteration does not change as
        # the ring rotates -- the precondition for replaying a captured iteration (``GraphedTrainStep``).
        self.dev_start = None
        self._dev_mirror = 0     # the value the host knows ``dev_start`` to hold (it trails ``start`` between a commit and the tick)
        self.generation = 0      # bumped whenever the window is replaced from outside (``load`` / ``adopt_full_slots``): captured iterations are stale then

    def position_on_device(self):
        """Keep the ring position on the device from now on (sparse rings whose window is full)."""
        assert self.sparse and self.len == self.T
        if self.dev_start is None:
            self.dev_start = torch.zeros(1, dtype=torch.int32, device=self.buf.device)
        self.dev_start.fill_(self.start)
        self._dev_mirror = self.start

    def _ref(self, slot: int):
        """``lstep_ring_ref_t*`` for physical slot ``slot`` (None while the position lives on the host)."""
        if self.dev_start is None:
            return None
        return ctypes.byref(nat.RingRef(self.dev_start.data_ptr(), (int(slot) - self._dev_mirror) % self.S, self.S, self.rows * self.P))

    def tick(self):
        """End of an iteration: the device-resident position follows the host's (one launch on the current stream)."""
        if self.dev_start is None:
            return
        assert (self._dev_mirror + 1) % self.S == self.start, "exactly one commit per iteration"
        if self._advanced[1] is not None:     # the advance of this iteration reads the position on another stream: it goes first
            torch.cuda.current_stream(self.buf.device).wait_event(self._advanced[1])
        early = self.__dict__.pop("_early_event", None)
        if early is not None:                 # (captured iterations: the slide applied at the start of the replay, ``early_advance``)
            torch.cuda.current_stream(self.buf.device).wait_event(early)
        with torch.cuda.device(self.buf.device):
            nat.check(nat.load_library().lstep_ring_tick(nat.ptr(self.dev_start), self.S, nat.current_stream()))
        self._dev_mirror = self.start

    def replay_tick(self):
        """A captured iteration was replayed: its kernels advanced the device position; the host mirrors follow.  The slide of its commit
        is pending (the next replay applies it first thing, ``early_advance``; a launch-by-launch iteration settles it at its start)."""
        self.start = (self.start + 1) % self.S
        self._dev_mirror = self.start
        if self.sparse:
            self._advance = self.start

    def begin_slot(self, slot: int = None, all_changed: bool = False):
        """Reset the change bits of the snapshot about to be built (on the current stream): nothing changed yet, or everything."""
        if self.mask is None:
            return
        if self.sparse and all_changed:
            self._all_written = True
        slot = (self.start + self.len) % self.S if slot is None else slot
        with torch.cuda.device(self.mask.device):
            nat.check(nat.load_library().lstep_history_slot_bits(nat.ptr(self.mask), self.words, self.rows, int(slot), int(all_changed),
                                                                 self._ref(slot), nat.current_stream()))

    def written(self, ids: torch.Tensor, mirrored: bool = False):
        """``update_pe``'s callback: rows ``ids`` were written (``mirrored``: into ``building()`` as well, nothing left to copy)."""
        self.mark(ids, copy=not mirrored)

    def building(self):
        """Sparse rings: the slot of the snapshot being built, for writers that can put their rows there directly (``lstep_update_rows``)."""
        return self.buf[(self.start + self.len) % self.S] if self.sparse else None

    def building_ref(self):
        """(ring base, ``lstep_ring_ref_t*`` of the slot being built) when the position lives on the device, else (None, None)."""
        if self.dev_start is None:
            return None, None
        return self.buf, self._ref((self.start + self.len) % self.S)

    def mark(self, ids: torch.Tensor, world: int = 1, rank: int = 0, copy: bool = True):
        """Rows ``ids`` (int64 node ids; ``world > 1``: only those owned by ``rank``, stored at row id // world) of the snapshot being built
        were written (``copy=False``: and the writer has already put them into ``building()``)."""
        if self.mask is None or ids.numel() == 0:
            return
        if self.sparse and copy:
            self._written.append(ids)
        slot = (self.start + self.len) % self.S
        with torch.cuda.device(self.mask.device):
            nat.check(nat.load_library().lstep_history_mark(nat.ptr(self.mask), self.words, self.rows, slot, nat.ptr(ids), ids.numel(),
                                                            int(world), int(rank), self._ref(slot), nat.current_stream()))

    def geom(self):
        """(node_stride, time_stride, slots, rot, t_len, P) for ``lstep_history_filter_*`` (element strides)."""
        return (self.P, self.rows * self.P, self.S, self.start, self.len, self.P)

    def window_ref(self):
        """``lstep_ring_ref_t*`` of the window's oldest slot (the filter kernels' rotation), or None."""
        return self._ref(self.start)

    def last(self) -> torch.Tensor:
        assert self.len > 0
        return self.table if self.sparse else self.buf[(self.start + self.len - 1) % self.S]

    def spare(self) -> torch.Tensor:
        return self.table if self.sparse else self.buf[(self.start + self.len) % self.S]

    def commit(self):
        """The snapshot being built is finished (``train:301`` append + ``train:224-225`` trim)."""
        if self.sparse:
            # whatever the writers have not put into the slot themselves (update_pe's fused kernels do) is copied now: row 0, and the
            # rows of writers without a mirror (library-GEMM update path)
            lib = nat.load_library()
            slot = (self.start + self.len) % self.S
            dst = self.buf[slot]
            with torch.cuda.device(self.buf.device):
                if self._all_written or self.len == 0:
                    dst.copy_(self.table)
                else:
                    ref = self._ref(slot)
                    for ids in self._written + [self._row0]:     # (row 0 is rewritten by every update_pe and always marked)
                        nat.check(lib.lstep_copy_rows(nat.ptr(self.buf if ref is not None else dst), nat.ptr(self.table), self.P, self.P,
                                                      nat.ptr(ids), ids.numel(), self.rows, ref, nat.current_stream()))
                if self.len == 0:
                    self.oldest.copy_(self.table)
            self._written, self._all_written = [], False
        if self.len < self.T:
            self.len += 1
        else:
            self.start = (self.start + 1) % self.S
            if self.sparse:
                assert self._advance is None, "HistoryRing.apply_advance() must follow every commit()"
                self._advance = self.start
        if self.sparse:
            self.begin_slot()      # the next snapshot starts with no row written

    def apply_advance(self):
        """Sparse rings: bring ``oldest`` to the window's new first snapshot after a ``commit`` that slid the window.  Call when the
        current stream has been given everything that still reads the OLD window (the FFT filter's backward pass): the update runs on
        the copy stream behind that point.  Readers of the NEW window need not wait for it -- they take the rows it moves from the
        first slot itself (``lstep_history_filter_runs_*``) -- only for the update before it (``wait_window``).
        Inside a captured iteration nothing is launched here: the slide stays pending and is applied at the START of the next replay
        (``early_advance``), beside the forward pass instead of between the backward pass and the optimiser step."""
        if not self.sparse or self._advance is None:
            return
        if torch.cuda.is_current_stream_capturing():
            return
        slot, self._advance = self._advance, None
        dev = self.buf.device
        here = torch.cuda.Event()
        here.record()
        side = self.advance_stream or self._copy_stream
        with torch.cuda.device(dev), torch.cuda.stream(side):
            side.wait_event(here)
            ref = self._ref(slot)
            nat.check(nat.load_library().lstep_history_advance_oldest(nat.ptr(self.oldest), nat.ptr(self.buf if ref is not None else self.buf[slot]),
                                                                      self.P, self.P, nat.ptr(self.mask), self.words, slot, self.rows, ref,
                                                                      nat.current_stream()))
            ev = torch.cuda.Event()
            ev.record()
        self._advanced = [self._advanced[1], ev]

    def early_advance(self):
        """Captured iterations: apply the slide the PREVIOUS iteration's commit left pending, as the first thing of this one.  It only
        writes rows whose change bit of the window's first slot is set, and every reader of this iteration's window takes exactly those
        rows from the slot itself, so it may run beside the whole forward and backward pass; the caller joins its stream before the
        capture ends, i.e. before the next replay's advance.  At capture time the launch-by-launch iteration before has already applied
        its slide: the same rows are copied once more (idempotent)."""
        assert self.sparse and torch.cuda.is_current_stream_capturing()
        slot = self._advance if self._advance is not None else self.start
        assert slot == self.start, "the pending slide is the window's first slot"
        self._advance = None
        dev = self.buf.device
        here = torch.cuda.Event()
        here.record()
        side = self.advance_stream or self._copy_stream
        with torch.cuda.device(dev), torch.cuda.stream(side):
            side.wait_event(here)
            ref = self._ref(slot)
            nat.check(nat.load_library().lstep_history_advance_oldest(nat.ptr(self.oldest), nat.ptr(self.buf if ref is not None else self.buf[slot]),
                                                                      self.P, self.P, nat.ptr(self.mask), self.words, slot, self.rows, ref,
                                                                      nat.current_stream()))
            # the slide reads the device-resident ring position: ``tick`` (which moves it) waits for this event, which also joins the slide's
            # stream into the capturing one whichever stream it is (the auxiliary stream, or the ring's copy stream with LSTEP_NO_AUX_STREAM=1)
            self._early_event = torch.cuda.Event()
            self._early_event.record()
        self._advanced = [None, None]

    def wait_window(self):
        """Make the current stream wait until the window can be read: ``oldest`` is at most one slide behind."""
        if not self.sparse:
            return
        st = torch.cuda.current_stream(self.buf.device)
        if self._advance is not None and self._advanced[1] is not None:     # a slide is still pending: the one before it must be done
            st.wait_event(self._advanced[1])
        elif self._advanced[0] is not None:
            st.wait_event(self._advanced[0])

    def prefetch_base(self):
        """Clone mode: start copying the newest snapshot into the next spare slot on the copy stream (call right after ``commit``)."""
        if self.sparse or self._copy_stream is None or self.len == 0 or os.environ.get("LSTEP_NO_PREFETCH") == "1":
            return
        slot = (self.start + self.len) % self.S
        main = torch.cuda.current_stream(self.buf.device)
        self._copy_stream.wait_stream(main)
        with torch.cuda.stream(self._copy_stream):
            self.buf[slot].copy_(self.last(), non_blocking=True)
            self.begin_slot(slot)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        self._prefetched = (slot, ev)

    def base_for_next(self) -> torch.Tensor:
        """The table the next snapshot is built in, holding the newest snapshot (``torch.clone(positional_encoding[:, -1, :])``,
        train:229).  Clone mode: the spare slot -- the prefetched copy if ``prefetch_base`` ran for it, otherwise copied now."""
        if self.sparse:
            return self.table
        slot = (self.start + self.len) % self.S
        cur = self.buf[slot]
        if self._prefetched is not None and self._prefetched[0] == slot:
            torch.cuda.current_stream(self.buf.device).wait_event(self._prefetched[1])
        else:
            cur.copy_(self.last())
            self.begin_slot(slot)
        self._prefetched = None
        return cur

    def load(self, history: torch.Tensor):
        """Adopt a reference-shaped history ``[N+1, t, P]`` (keeps the last T snapshots)."""
        t = history.shape[1]
        keep = min(t, self.T)
        self.start, self.len = 0, keep
        self._prefetched = None
        if keep:
            self.buf[:keep].copy_(history[:, t - keep:, :].permute(1, 0, 2))
        self.adopt_full_slots()

    def adopt_full_slots(self):
        """The window's slots were filled with FULL snapshots from outside (``load``, a synthetic pre-fill): derive the change bits from
        them and, for a sparse ring, the ``oldest`` / current tables."""
        self.recompute_mask()
        self._reposition()
        if self.sparse:
            self._written, self._all_written, self._advance, self._advanced = [], False, None, [None, None]
            if self.len:
                self.oldest.copy_(self.buf[self.start])
                self.table.copy_(self.buf[(self.start + self.len - 1) % self.S])
            self.begin_slot()

    def _reposition(self):
        """The window was replaced from outside (checkpoint reload, ``EarlyStopping.load_pe``-style best-PE reload, a pre-fill): the
        device-resident position follows the host's when the window is full; a shorter window goes back to host positions (the
        device word only ever serves full, rotating windows: ``tick`` moves it by exactly one slot per iteration).  Whoever replays
        captured iterations compares ``generation`` and captures again."""
        self.generation += 1
        if self.dev_start is not None:
            if self.sparse and self.len == self.T:
                self.dev_start.fill_(self.start)
                self._dev_mirror = self.start
            else:
                self.dev_start, self._dev_mirror = None, 0

    def recompute_mask(self):
        """Change bits of the whole window from the stored rows; only meaningful while every slot of the window holds a full snapshot
        (clone mode, or right after ``load`` / a pre-fill)."""
        if self.mask is None:
            return
        self.mask.zero_()
        for i in range(self.len):
            ph = (self.start + i) % self.S
            if i == 0:
                differs = torch.ones(self.rows, dtype=torch.bool, device=self.buf.device)
            else:
                differs = (self.buf[ph] != self.buf[(ph - 1) % self.S]).any(dim=1)
                differs[:1] = True
            bit = 1 << (ph % 32)
            self.mask[:, ph // 32] |= differs.to(torch.int32) * (bit - (1 << 32) if bit >= (1 << 31) else bit)

    def as_reference_tensor(self) -> torch.Tensor:
        """``[N+1, t, P]`` copy of the window, oldest first (tests / checkpoint parity with ``EarlyStopping.save_pe``)."""
        if not self.sparse:
            idx = [(self.start + i) % self.S for i in range(self.len)]
            return self.buf[idx].permute(1, 0, 2).contiguous()
        out = torch.empty((self.rows, self.len, self.P), dtype=torch.float32, device=self.buf.device)
        for i, snap in enumerate(self.snapshots()):
            out[:, i] = snap
        return out

    def snapshots(self):
        """Yield the window's snapshots ``[rows, P]`` one at a time, oldest first (a 1 M-node window is 69 GB as one tensor).  Clone
        rings yield views of their slots; a sparse ring rebuilds each snapshot from the one before it and the rows whose change bit
        of that slot is set (the yielded tensor is reused for the next one: copy what must be kept)."""
        idx = [(self.start + i) % self.S for i in range(self.len)]
        if not self.sparse:
            for ph in idx:
                yield self.buf[ph]
            return
        torch.cuda.synchronize(self.buf.device)
        cur = None
        for i, ph in enumerate(idx):
            hit = ((self.mask[:, ph // 32] >> (ph % 32)) & 1).bool().unsqueeze(1)
            # (i == 0: rows the first slot's batch wrote are in the slot; ``oldest`` may not have taken them over yet)
            cur = torch.where(hit, self.buf[ph], self.oldest if i == 0 else cur)
            yield cur


class BatchKey:
    """Identity of a batch's endpoint tensors for the look-ahead grouping: the same memory (address, length, layout), not written
    since (version counters are shared by all views of a buffer).  The key HOLDS the tensors it was made from: as long as it lives
    the caching allocator cannot hand their block to another batch's tensors, so an equal address means the same buffer -- without the
    references a freed look-ahead batch and a later batch of the same size (epoch boundary, train -> eval switch) could collide."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor):
        self.src, self.dst = src, dst
        self.sig = self._sig(src, dst)

    @staticmethod
    def _sig(src, dst):
        return tuple((t.data_ptr(), t.numel(), t.stride(), t.dtype, t._version) for t in (src, dst))

    def matches(self, src: torch.Tensor, dst: torch.Tensor) -> bool:
        return self.sig == self._sig(src, dst) and self.sig == self._sig(self.src, self.dst)


class _LookupRows(torch.autograd.Function):
    """``table[ids]`` whose gradient flows to the spliced rows only (their values already live in ``table``).

    Backward is one ``index_add_`` into ``[U + n, P]``: ids that are not spliced rows land in private dummy rows past U,
    which are dropped (no data-dependent shapes, no host sync, no hot duplicate index for the atomics)."""

    @staticmethod
    def forward(ctx, rows, table, slot_of, ids):
        pos = slot_of[ids].long()
        ctx.save_for_backward(pos)
        ctx.u = rows.shape[0]
        return table[ids]

    @staticmethod
    def backward(ctx, g):
        (pos,) = ctx.saved_tensors
        u = ctx.u
        n = pos.numel()
        acc = torch.zeros((u + n, g.shape[1]), dtype=g.dtype, device=g.device)
        dump = torch.arange(u, u + n, device=g.device)        # ids that are not spliced rows: one private dummy row each
        acc.index_add_(0, torch.where(pos >= 0, pos, dump), g)
        return acc[:u], None, None, None


def _lookup_rows(table: torch.Tensor, spliced: SplicedRows, ids: torch.Tensor) -> torch.Tensor:
    if spliced is None:
        return table[ids]
    return _LookupRows.apply(spliced.rows, table, spliced.slot_of, ids)


class _LinkLoss(torch.autograd.Function):
    """``lstep_link_loss``: link-prediction BCE + positional-encoding MSE terms of train:257-275 and the gradient of their weighted
    sum (the gradient is produced in forward; backward only scales it by the incoming gradient).  The kernel emits the gradient of the
    positional-encoding rows per occurrence; ``groups = (seg, order)`` -- the int32 grouping of cat[src, dst] by batch node the engine made
    for the batch-node set -- lets ``lstep_segment_rows_sum`` reduce them by spliced row in a fixed order (the negatives that happen to be
    batch nodes, and everything when no grouping is given, go through ``lstep_scatter_add_rows``)."""

    @staticmethod
    def forward(ctx, logits, rows, table, slot_of, ids, pe_weight, neg_weight, groups=None):
        ctx.set_materialize_grads(False)     # (no zero tensors for the outputs nobody differentiates: each would be a fill launch in backward)
        lib = nat.load_library()
        dev = logits.device
        n = ids.numel() // 3
        P = rows.shape[1]
        logits = logits.contiguous()
        predicts = torch.empty(2 * n, dtype=torch.float32, device=dev)
        d_logits = torch.empty(2 * n, dtype=torch.float32, device=dev)
        g_rows = torch.empty((3 * n, P), dtype=torch.float32, device=dev)
        neg_slot = torch.empty(n, dtype=torch.int32, device=dev)
        from .model import SPLICE_SMALL_HITS, SPLICE_SMALL_ROWS
        small = rows.shape[0] <= SPLICE_SMALL_ROWS and 3 * n <= SPLICE_SMALL_HITS and os.environ.get("LSTEP_NO_SMALL_SPLICE") != "1"
        d_rows = torch.empty_like(rows) if small else torch.zeros_like(rows)
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        ws = nat._workspace(dev, int(lib.lstep_link_loss_workspace(n)))
        rows_c = rows.detach().contiguous()
        with torch.cuda.device(dev):
            nat.check(lib.lstep_link_loss(nat.ptr(logits), nat.ptr(ids), n, nat.ptr(table), nat.ptr(rows_c), nat.ptr(slot_of), P,
                                          float(pe_weight), float(neg_weight), nat.ptr(predicts), nat.ptr(d_logits), nat.ptr(g_rows),
                                          nat.ptr(neg_slot), nat.ptr(losses), nat.ptr(ws), ws.numel(), nat.current_stream()))
            ld = int(d_rows.stride(0))
            if small:
                # small batches: the per-occurrence rows are reduced by spliced row in ONE scanning launch, in occurrence order -- no
                # grouping, no join, and none of the float atomics the negatives' scatter needs (lstep_spliced_grad_small writes every row)
                nat.check(lib.lstep_spliced_grad_small(None, 0, 1, None, P, nat.ptr(slot_of), nat.ptr(ids), 3 * n, nat.ptr(g_rows), P, P,
                                                       nat.ptr(d_rows), ld, rows.shape[0], nat.current_stream()))
            elif groups is not None and groups[0].numel() == 2 * n:
                seg, order = groups
                sws, sws_bytes = nat.segment_workspace(dev, 2 * n, P)
                nat.check(lib.lstep_segment_rows_sum(nat.ptr(g_rows), P, P, None, None, 0, nat.ptr(seg), nat.ptr(order), None, 2 * n,
                                                     nat.ptr(d_rows), ld, 1, None, nat.ptr(sws), sws_bytes, nat.current_stream()))
                g_neg = g_rows[2 * n:]
                nat.check(lib.lstep_scatter_add_rows(nat.ptr(d_rows), P, ld, nat.ptr(neg_slot), n, nat.ptr(g_neg), P, nat.current_stream()))
            else:
                slots = slot_of[ids].contiguous()
                nat.check(lib.lstep_scatter_add_rows(nat.ptr(d_rows), P, ld, nat.ptr(slots), 3 * n, nat.ptr(g_rows), P, nat.current_stream()))
        ctx.save_for_backward(d_logits, d_rows)
        lp, pe, loss = losses[0], losses[1], losses[2]
        ctx.mark_non_differentiable(lp, pe, predicts)
        return loss, lp, pe, predicts

    @staticmethod
    def backward(ctx, g_loss, g_lp, g_pe, g_pred):
        if g_loss is None:
            return (None,) * 8
        d_logits, d_rows = ctx.saved_tensors
        if _LinkLoss.unit_gradient:      # the engine's own ``loss.backward()``: the incoming gradient is the scalar 1 (two launches less)
            return d_logits, d_rows, None, None, None, None, None, None
        return g_loss * d_logits, g_loss * d_rows, None, None, None, None, None, None

    unit_gradient = False     # set by LstepEngine around its own backward call (a plain ``loss.backward()``)


def _backward_unit(loss):
    """``loss.backward()`` of the engine's own iteration: the seed gradient is 1, ``_LinkLoss.backward`` hands its saved gradients on as they are."""
    _LinkLoss.unit_gradient = True
    try:
        loss.backward()
    finally:
        _LinkLoss.unit_gradient = False


class GraphedTrainStep:
    """One steady-state training iteration (train_LSTEP_link_prediction.py:204-311: FFT splice, 3 x combine, predictor, losses,
    update_pe, snapshot append, backward, Adam) captured ONCE as a HIP graph and replayed per batch.

    What makes the iteration a fixed launch sequence: every data-dependent size stays on the device (``LstepEngine.device_counts``),
    the ring position is read on the device (``HistoryRing.position_on_device``), the gradient sort runs on a fixed capacity (overflow
    stays exact), and the batch arrives in fixed buffers (five small device copies per step).  The three streams of the eager engine
    become parallel branches of the graph.  Per step the host issues the input copies and one graph launch: ~0.1 ms instead of the
    ~2.3 ms it takes Python to issue ~180 launches one by one -- the difference between 0.09 and 0.4 M edges/s at the reference's own
    batch sizes (B = 200).  Losses and link probabilities are device tensors that the next replay overwrites."""

    def __init__(self, engine: "LstepEngine", optimizer, batch: int):
        dev = engine.device
        self.eng, self.optimizer, self.B = engine, optimizer, int(batch)
        i64 = lambda: torch.zeros(self.B, dtype=torch.int64, device=dev)  # noqa: E731
        self.src, self.dst, self.eid, self.neg = i64(), i64(), i64(), i64()
        self.ts = torch.zeros(self.B, dtype=torch.float64, device=dev)
        self.graph, self.out = None, None
        self.replays = 0          # graph launches that stood for an iteration, the capture's own first replay not counted
        self.hyper = None         # the optimiser's scalars as they were baked into the captured Adam launch

    def _hyper(self):
        o = self.optimizer
        return (o.lr, o.weight_decay, tuple(o.betas), o.eps)

    def close(self):
        """Give the captured graph back (destroyed at the next safe point); the next ``step`` captures again."""
        from .model import drain_dead_graphs, retire_graph
        if self.graph is not None:
            retire_graph(self.graph)
        self.graph, self.out = None, None
        drain_dead_graphs()

    def step(self, batch_idx: int, src, dst, ts, eid, neg_dst):
        torch._foreach_copy_([self.src, self.dst, self.eid, self.neg], [src, dst, eid, neg_dst])
        self.ts.copy_(ts)
        eng = self.eng
        eng.__dict__.pop("_prefetched_group", None)        # the captured iteration groups its own batch
        if self.graph is not None and self.hyper != self._hyper():
            # lr / betas / eps / weight decay are launch constants of the captured Adam kernel: a schedule or a load_state_dict that changes
            # them makes the captured iteration stale
            self.close()
        if self.graph is None:
            self._capture(batch_idx)
        else:
            self.graph.replay()
            self.replays += 1
            eng.ring.replay_tick()
        return self.out

    def _capture(self, batch_idx: int):
        from .model import _aux_stream
        eng, ring = self.eng, self.eng.ring
        torch.cuda.synchronize(eng.device)
        ring._advanced = [None, None]          # events of eager iterations must not be waited for inside the capture
        if ring.dev_start is None:
            ring.position_on_device()
        from .model import _no_gc, new_graph
        graph = new_graph(self)
        self.hyper = self._hyper()
        dot = os.environ.get("LSTEP_GRAPH_DOT")       # diagnostics: the captured graph's nodes and edges as a DOT file (tools/graph_dot_summary.py)
        if dot:
            graph.enable_debug_mode()
        # (a process that also runs collectives -- lstep_amd.parallel beside this engine, a caller's own -- has an NCCL watchdog thread
        # polling their events: thread_local keeps its queries legal during the capture, the quiesce keeps it from polling an event whose
        # stream is about to capture; both are free without a process group)
        nat.quiesce_collectives(eng.device)
        with _no_gc(), torch.cuda.graph(graph, stream=nat.role_stream(eng.device, "capture"), capture_error_mode="thread_local"):
            with eng.aux_streams():
                self.out = eng._train_iteration(self.optimizer, batch_idx, self.src, self.dst, self.ts, self.eid, self.neg, None, None)
            main = torch.cuda.current_stream(eng.device)
            if eng.use_aux:
                main.wait_stream(_aux_stream(eng.device))      # (the backward pass's side stream was joined by join_aux_stream already)
            main.wait_stream(eng._update_stream)
        if dot:
            graph.debug_dump(dot)
        self.graph = graph
        graph.replay()      # capturing records the launches without running them: this replay IS the iteration


class LstepEngine:
    def __init__(self, backbone: LSTEP, predictor: MergeLayer, num_neighbors: int, time_gap: int,
                 pe_weight: float = 0.5, neg_sample_weight: float = 0.3, make_ring: bool = True):
        self.backbone, self.predictor = backbone, predictor
        self.K, self.G = int(num_neighbors), int(time_gap)
        self.pe_weight, self.neg_sample_weight = pe_weight, neg_sample_weight
        dev = backbone.device
        self.device = dev
        rows = backbone.node_raw_features.shape[0]
        # make_ring=False: lstep_amd.parallel.DistributedLstep owns an owner-sharded ring instead
        self.ring = HistoryRing(rows, backbone.pe_dim, backbone.num_fft_batches, dev, sparse=True) if make_ring else None
        # update_pe on a side stream underneath the backward pass (LSTEP_NO_OVERLAP=1 runs the reference order serially)
        self.overlap_update = torch.device(dev).type == "cuda" and os.environ.get("LSTEP_NO_OVERLAP") != "1"
        self._update_stream = nat.role_stream(dev, "update")    # (a stream of our own, never one of PyTorch's pool: _native.role_stream)
        self.slot_of = torch.full((rows,), -1, dtype=torch.int32, device=dev)
        self.fused_loss = torch.device(dev).type == "cuda" and backbone.pe_dim % 4 == 0 and os.environ.get("LSTEP_TORCH_LOSS") != "1"
        # every data-dependent size (batch nodes, grouped neighbour slots, touched rows) stays on the device: no host synchronisation and
        # no second host thread anywhere in an iteration.  Needs the change-mask ring and the fused kernels; LSTEP_HOST_COUNTS=1 (A/B
        # switch) restores the host-sized path, which also serves every other configuration.
        self._want_device_counts = os.environ.get("LSTEP_HOST_COUNTS") != "1"
        # LSTEP_RING_ON_DEVICE=1: keep the ring position on the device in eager mode too (the graphed step always does)
        self._ring_on_device = os.environ.get("LSTEP_RING_ON_DEVICE") == "1"
        # Steady-state training iterations replayed as ONE captured HIP graph (``GraphedTrainStep``): LSTEP_STEP_GRAPH=1, or set the
        # attribute.  Off by default: the drop-in behaviour (any batch size, any optimiser, host-visible losses) needs no capture.
        self.use_step_graph = os.environ.get("LSTEP_STEP_GRAPH") == "1"
        self._graphed = {}                 # batch size -> GraphedTrainStep
        self._steady_eager_steps = 0       # eager training iterations run with a full window (they prime the capture)
        self._ring_generation = 0          # ``HistoryRing.generation`` the captured iterations belong to
        # the engine joins the auxiliary stream before every optimiser step, so INSIDE its training iteration (``aux_streams``) the model
        # may put its weight-gradient products there; outside of it every backward() is self-contained on the caller's stream
        self.use_aux = torch.device(dev).type == "cuda" and os.environ.get("LSTEP_NO_AUX_STREAM") != "1"
        backbone.aux_wgrad_stream = predictor.aux_wgrad_stream = False
        if self.ring is not None and self.ring.sparse and self.use_aux:
            from .model import _aux_stream
            self.ring.advance_stream = _aux_stream(dev)    # idle between the weight gradients and the next weight composition

    def close(self):
        """Release every captured graph of this engine and of its model now (explicit graph lifetime, ``model.new_graph``): call it when
        the engine is dropped while other models keep capturing, e.g. between test cases or when a trainer rebuilds its model.  The
        engine stays usable: it captures again after two steady-state eager iterations."""
        self.drop_captured_iterations()
        self.backbone.close()

    def drop_captured_iterations(self):
        for gs in self._graphed.values():
            gs.close()
        self._graphed = {}
        self._steady_eager_steps = 0

    @contextlib.contextmanager
    def aux_streams(self):
        """Scope in which the model may defer parameter-gradient work to the auxiliary stream (``LSTEP.join_aux_stream`` is called before
        the optimiser step inside it).  The flag is off everywhere else: a ``backward()`` of the drop-in methods, of
        ``compute_src_dst_node_temporal_embeddings`` or of user code leaves complete gradients on the caller's stream."""
        bb, pr = self.backbone, self.predictor
        bb.aux_wgrad_stream = pr.aux_wgrad_stream = self.use_aux
        try:
            yield
        finally:
            bb.aux_wgrad_stream = pr.aux_wgrad_stream = False

    # ---- shared pieces
    def _splice(self, batch_nodes: torch.Tensor, batch_idx: int, live: torch.Tensor = None):
        """FFT-filter the batch rows over the ring window and build the current PE in the spare slot (train:224-230).
        ``live``: ``batch_nodes`` is the capacity-sized list of ``batch_nodes_device`` and ``live`` its device-resident length."""
        ring = self.ring
        ring.wait_window()
        if ring.mask is not None and ring.len > 0:
            # the run kernel writes the filtered rows into the current table and numbers them in slot_of on its way out
            cur = ring.base_for_next()
            rows = self.backbone.filter_history(ring.buf, ring.geom(), batch_nodes, batch_idx, mask=ring.mask, oldest=ring.oldest,
                                                splice=(cur, self.slot_of), live=live, ring=ring.window_ref())
        else:
            assert live is None, "device-resident counts need a ring with a change mask that already holds a snapshot"
            rows = self.backbone.filter_history(ring.buf, ring.geom(), batch_nodes, batch_idx, mask=ring.mask, oldest=ring.oldest)
            cur = ring.base_for_next()
            cur.index_copy_(0, batch_nodes, rows.detach())
            self.slot_of[batch_nodes] = torch.arange(batch_nodes.numel(), dtype=torch.int32, device=self.device)
        # (the spliced rows are marked as changed by update_pe's phase 1: same node set)
        # the engine's gather rows are cat[src, dst, neg]: their first 2B rows are the entries batch_nodes_and_segments grouped
        return cur, SplicedRows(rows, self.slot_of, getattr(self, "_batch_groups", None))

    def _probabilities(self, a, b):
        return self.predictor(input_1=a, input_2=b).squeeze(dim=-1).sigmoid().clamp(0, 1)

    def _labels(self, n: int) -> torch.Tensor:
        """[1]*n + [0]*n (train:263), cached per batch size."""
        lab = getattr(self, "_label_cache", None)
        if lab is None or lab.numel() != 2 * n:
            lab = self._label_cache = torch.cat([torch.ones(n, device=self.device), torch.zeros(n, device=self.device)])
        return lab

    @property
    def device_counts(self) -> bool:
        bb = self.backbone
        return (self._want_device_counts and self.ring is not None and self.ring.sparse and bb._fused_tail_ok()
                and getattr(bb.neighbor_sampler, "sample_neighbor_strategy", "recent") == "recent"
                and not any(os.environ.get(v) == "1" for v in ("LSTEP_TORCH_UPDATE", "LSTEP_TORCH_ENTRIES")))

    def _group_batch(self, src, dst, wait, keys=None):
        rows = self.backbone.node_raw_features.shape[0]
        if keys is None:
            keys = torch.cat([src, dst]).to(torch.int32)
        _, order, seg, uniq, counts = nat.group_by_key(keys, max(1, int(rows).bit_length()), rows, wait=wait)
        return order, seg, uniq, counts

    def prepare_batch(self, src, dst, neg_dst, ts):
        """``lstep_batch_prepare``: (cat[src, dst, neg_dst], cat[ts, ts, ts], int32 cat[src, dst], float32(max ts)) in one launch instead
        of three concatenations, a cast, a reduction and another cast."""
        n = src.numel()
        dev = self.device
        ids3 = torch.empty(3 * n, dtype=torch.int64, device=dev)
        t3 = torch.empty(3 * n, dtype=torch.float64, device=dev)
        keys = torch.empty(2 * n, dtype=torch.int32, device=dev)
        now32 = torch.empty(1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(nat.load_library().lstep_batch_prepare(nat.ptr(src.contiguous()), nat.ptr(dst.contiguous()), nat.ptr(neg_dst.contiguous()),
                                                             nat.ptr(ts.contiguous()), n, nat.ptr(ids3), nat.ptr(t3), nat.ptr(keys), nat.ptr(now32),
                                                             nat.current_stream()))
        return ids3, t3, keys, now32

    def batch_nodes_device(self, src, dst, keys=None):
        """``batch_nodes_and_segments`` without any host round trip: (bn int64 [2 B] = the sorted unique endpoints followed by a dead tail
        of node 0, n_live int32 [1] = their number on the device, (order, seg) = the int32 grouping of cat[src, dst] by batch node)."""
        pre = self.__dict__.pop("_prefetched_group", None)
        if pre is not None and pre[0].matches(src, dst) and isinstance(pre[1][3], torch.Tensor):
            order, seg, uniq, summary = pre[1]
        else:
            order, seg, uniq, summary = self._group_batch(src, dst, wait=None, keys=keys)
        bn = torch.empty(uniq.numel(), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            nat.check(nat.load_library().lstep_widen_ids(nat.ptr(uniq), uniq.numel(), nat.ptr(summary), nat.ptr(bn), nat.current_stream()))
        self._batch_groups = (seg, order)
        return bn, summary[0:1], (order, seg)

    def batch_nodes_and_segments(self, src, dst):
        """One native group-by-key of cat[src, dst] (``lstep_group_by_key``) gives the sorted unique batch nodes
        (train:221-222) AND the per-node segments of update_pe phase 1 (entries grouped by receiving endpoint).
        If ``prefetch_batch_nodes`` was called for exactly these tensors, its result is picked up: the only host wait is for the
        prefetched counts, which were copied out long ago."""
        pre = self.__dict__.pop("_prefetched_group", None)
        if pre is not None and pre[0].matches(src, dst) and not isinstance(pre[1][3], torch.Tensor):
            order, seg, uniq, counts = pre[1]
            n_unique = counts.get()[0]
        else:
            order, seg, uniq, (n_unique, _, _) = self._group_batch(src, dst, wait=True)
        self._batch_groups = (seg, order)          # int32: the batch rows cat[src, dst] grouped by batch node = by spliced row
        return uniq[:n_unique].long(), (order, seg, None)     # int32 order / segment ids (update_pe's phase 1 consumes them as they are)

    def prefetch_batch_nodes(self, src, dst):
        """Group the NEXT batch's endpoints now (the edge stream is known ahead).  Its kernels queue up behind the current forward pass
        and its counts travel to the host asynchronously, so the next ``train_iteration`` starts without draining the GPU: the host
        can enqueue the next forward while the current backward is still running."""
        self._prefetched_group = (BatchKey(src, dst), self._group_batch(src, dst, wait=None if self.device_counts else False))

    # ---- train:204-311
    def train_iteration(self, optimizer, batch_idx: int, src, dst, ts, eid, neg_dst, initial_pe: torch.Tensor = None, lookahead=None):
        """``lookahead = (src, dst[, ts, neg_dst])`` of the next batch (optional): see ``prefetch_batch_nodes``.  The grouping made from them is used by
        the next call only if it receives the same (unmodified) tensors; otherwise it is recomputed."""
        from .model import drain_dead_graphs
        drain_dead_graphs()          # a safe point: nothing is capturing here (graphs of dropped models / closed engines die now)
        if self.ring is not None and self._ring_generation != self.ring.generation:
            # the history window was replaced from outside (``HistoryRing.load``): captured iterations belong to the old one
            self.drop_captured_iterations()
            self._ring_generation = self.ring.generation
        if self._graph_ready(optimizer, batch_idx):
            gs = self._graphed.get(src.numel())
            if gs is None or gs.optimizer is not optimizer:
                if gs is not None:
                    gs.close()
                gs = self._graphed[src.numel()] = GraphedTrainStep(self, optimizer, src.numel())
            return gs.step(batch_idx, src, dst, ts, eid, neg_dst)
        if self.ring is not None and self.ring.sparse and self.ring.len == self.ring.T and batch_idx > 0:
            self._steady_eager_steps += 1
        with self.aux_streams():
            return self._train_iteration(optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead)

    def _graph_ready(self, optimizer, batch_idx: int) -> bool:
        """A whole training iteration can be replayed as one graph once nothing in it depends on the host any more: sizes on the device
        (``device_counts``), a full window (no ``batch_idx`` mask in the FFT filter), the single-kernel Adam whose step counters live
        on the device, and two eager iterations of this shape behind us (they size the gradient sort and warm every lazy
        initialisation)."""
        from .optim import FusedAdam
        ring = self.ring
        return (self.use_step_graph and batch_idx > 0 and self.device_counts and ring.len == ring.T
                and isinstance(optimizer, FusedAdam) and self._steady_eager_steps >= 2 and self.overlap_update and self.fused_loss)

    def _train_iteration(self, optimizer, batch_idx, src, dst, ts, eid, neg_dst, initial_pe, lookahead):
        bb, ring = self.backbone, self.ring
        out, loss = None, None
        if ring.sparse:
            if torch.cuda.is_current_stream_capturing():
                ring.early_advance()         # the previous iteration's slide: beside this iteration instead of in front of its optimiser step
            else:
                ring.apply_advance()         # (a slide left pending by a replayed iteration; nothing reads the old window any more)
        if (self._ring_on_device and ring.dev_start is None and ring.sparse and ring.len == ring.T and ring._advance is None):
            ring.position_on_device()
        bb.prepare_step()
        on_device = self.device_counts and (batch_idx == 0 or ring.len > 0)
        prep = None
        if (on_device and batch_idx > 0 and src.dtype == dst.dtype == neg_dst.dtype == torch.int64 and ts.dtype == torch.float64
                and src.device == dst.device == neg_dst.device == ts.device and os.environ.get("LSTEP_NO_BATCH_PREPARE") != "1"):
            prep = self.prepare_batch(src, dst, neg_dst, ts)
        # LSTEP_SPLIT_FORWARD=1 (A/B, measured and left off: DESIGN.md appendix A): the gather stage's edge + node channels -- 70 % of its
        # bytes, and independent of the FFT splice -- launched on a side stream BEFORE the batch-node grouping and the history filter, the
        # PE channel behind the splice; two launches instead of one fused one.
        ahead = None
        if (prep is not None and os.environ.get("LSTEP_SPLIT_FORWARD") == "1" and bb._fused_tail_ok()
                and getattr(bb.neighbor_sampler, "sample_neighbor_strategy", "recent") == "recent"):
            main = torch.cuda.current_stream(self.device)
            fwd_side = nat.role_stream(self.device, "fwd-side")
            fwd_side.wait_stream(main)
            with torch.cuda.stream(fwd_side):
                x_edge, x_node, _, _, _ = bb._gather(None, prep[0], prep[1], self.K, self.G, nat.BRANCH_EDGE_NODE, wide=True, row_blocks=3)
                ev = torch.cuda.Event()
                ev.record()
            ahead = (x_edge, x_node, ev)
        if on_device:
            batch_nodes, n_live, presorted = self.batch_nodes_device(src, dst, keys=prep[2] if prep is not None else None)
        else:
            n_live = None
            batch_nodes, presorted = self.batch_nodes_and_segments(src, dst)
        if batch_idx == 0:
            cur = ring.spare()
            cur.copy_(initial_pe)
            ring.begin_slot(all_changed=True)
            spliced = None
        else:
            cur, spliced = self._splice(batch_nodes, batch_idx, live=n_live)
            n = src.numel()
            ids3, t3 = (prep[0], prep[1]) if prep is not None else (torch.cat([src, dst, neg_dst]), torch.cat([ts, ts, ts]))
            if ahead is not None:
                x_edge, x_node, ev = ahead
                main = torch.cuda.current_stream(self.device)
                main.wait_event(ev)
                for t_ in (x_edge, x_node):
                    t_.record_stream(main)
                _, _, x_pe, own, _ = bb._gather(cur, ids3, t3, self.K, self.G, nat.BRANCH_PE, spliced, wide=True, row_blocks=3)
                emb_p = bb._combined_tail(x_edge, x_node, x_pe, own, True)
            else:
                emb_p = bb.combining_pe_raw_feat(cur, ids3, t3, self.K, self.G, spliced=spliced, padded=True, row_blocks=3)
            emb = emb_p[:, :bb.feat_dim]
            pos_src = emb[:n]
            # both predictor calls of train:254-255 in one launch: rows [pos_src | pos_dst] and [pos_src | neg_dst] (neg_src = pos_src, train:245)
            if self.fused_loss:
                if self.predictor.fused_ok(emb_p):
                    logits = self.predictor.pair_logits(emb_p, n, (0, n, 0, 2 * n))
                else:
                    logits = self.predictor(input_1=torch.cat([pos_src, pos_src], dim=0), input_2=emb[n:]).squeeze(dim=-1)
                loss, lp_loss, pe_loss, predicts = _LinkLoss.apply(logits, spliced.rows, cur, spliced.slot_of, ids3, self.pe_weight,
                                                                   self.neg_sample_weight, getattr(spliced, "self_groups", None))
            else:   # the same terms with framework ops (LSTEP_TORCH_LOSS=1, the A/B switch)
                predicts = self._probabilities(torch.cat([pos_src, pos_src], dim=0), emb[n:])
                labels = self._labels(n)
                lp_loss = F.binary_cross_entropy(predicts, labels)
                e_all = _lookup_rows(cur, spliced, ids3)      # one gather / one scatter for the three PE lookups
                e_src = e_all[:n]
                pe_loss = F.mse_loss(e_src, e_all[n:2 * n]) - self.neg_sample_weight * F.mse_loss(e_src, e_all[2 * n:])
                loss = (1.0 - self.pe_weight) * lp_loss + self.pe_weight * pe_loss
            # ("embeddings": the [3 B, 176] rows of cat[src, dst, negative] the predictor read, columns >= 172 zero -- a view, for parity tests)
            out = {"lp_loss": lp_loss.detach(), "pe_loss": pe_loss.detach(), "loss": loss.detach(), "predicts": predicts.detach(),
                   "embeddings": emb_p.detach()}
        if lookahead is not None:
            self.prefetch_batch_nodes(*lookahead[:2])      # (a longer tuple also names the next batch's times / negatives: lstep_amd.parallel)

        def update_and_append():
            if on_device:
                base, ref = ring.building_ref()
                bb.update_pe_device(cur, batch_nodes, n_live, src, dst, ts, self.K, presorted, changed=ring.written,
                                    mirror=base if ref is not None else ring.building(), mirror_ring=ref,
                                    now32=prep[3] if prep is not None else None)
            else:
                bb.update_pe(pe=cur, node_ids=batch_nodes, edge_ids=eid, batch_src_node_ids=src, batch_dst_node_ids=dst,
                             node_interact_times=ts, current_time=ts.max(), num_neighbors=self.K, time_gap=self.G,
                             presorted=presorted, changed=ring.written, mirror=ring.building())
            if batch_idx == 0 and initial_pe is not None:
                initial_pe.copy_(cur)  # the reference mutates initial_positional_encoding in place at batch 0 (train:281,286)
            ring.commit()
            ring.prefetch_base()  # next iteration's clone of this snapshot runs on the copy stream

        if loss is None or not self.overlap_update:
            update_and_append()
            if loss is not None:
                optimizer.zero_grad()
                _backward_unit(loss)
                bb.join_aux_stream()
                optimizer.step()
                self.slot_of.index_fill_(0, batch_nodes, -1)   # (tensor[index] = scalar blocks the host until the GPU has drained)
            ring.apply_advance()      # (after the backward pass, which still reads the window as it was)
            ring.tick()
            return out

        # update_pe (forward-only, reads and writes only the current PE table) and the backward pass (never reads that table)
        # are independent: update_pe runs on the update stream, the backward pass on the main stream, both issued by this thread.
        main = torch.cuda.current_stream(self.device)
        side = self._update_stream
        side.wait_stream(main)
        if on_device:
            # nothing in update_pe waits for the GPU any more: its ~25 launches go out from THIS thread onto the side stream (0.1 ms of
            # host time), then the backward pass onto the main stream -- no second thread contending for the interpreter lock
            bwd_first = torch.cuda.is_current_stream_capturing() and os.environ.get("LSTEP_CAPTURE_BWD_FIRST") == "1"
            if not bwd_first:
                with torch.cuda.stream(side):
                    update_and_append()
            optimizer.zero_grad()
            _backward_unit(loss)
            if bwd_first:
                # (A/B switch, off: capturing the backward pass BEFORE update_pe -- same dependencies, other capture order -- in the hope that
                # the replayed backward pass would start right behind the loss kernel instead of ~0.4 ms later (profiles/r03_a_timeline.txt):
                # measured 3.67 against 3.51 ms at c4, 0.625 / 0.600 at B = 200, 1.35 / 1.26 at B = 4096.  DESIGN.md appendix A.)
                with torch.cuda.stream(side):
                    update_and_append()
            bb.join_aux_stream()
            ring.apply_advance()  # the backward pass is enqueued: the window's oldest snapshot may move on behind it
            main.wait_stream(side)
            optimizer.step()      # after update_pe has read its weights
            self.slot_of.index_fill_(0, batch_nodes, -1)   # (the dead tail is node 0, whose entry is -1 anyway)
            ring.tick()
            return out
        # Host-sized update_pe (RNG-defined strategies: numpy's generator, torch CPU ops, pageable copies both ways; LSTEP_HOST_COUNTS=1 and
        # non-default widths: a few host reads of data-dependent sizes): the backward pass is enqueued first and runs on the GPU underneath
        # update_pe, which THIS thread then issues onto the update stream.  There is no second host thread anywhere in this package: rounds
        # 2-4 drove this branch from one (it hid ~1 ms of host waits), and in round 4 that thread's launches landed on a stream another
        # thread was capturing a graph on -- PyTorch's round-robin stream pool had given the engine's update stream and torch.cuda.graph's
        # capture stream the same queue (the streams are dedicated now, _native.role_stream, and the thread is gone).
        try:
            optimizer.zero_grad()
            _backward_unit(loss)
            bb.join_aux_stream()
            with torch.cuda.stream(side):
                update_and_append()
        finally:
            ring.apply_advance()  # (no-op unless update_pe committed) the backward pass is enqueued: `oldest` may move on behind it
            main.wait_stream(side)
        optimizer.step()      # after update_pe has read its weights
        self.slot_of.index_fill_(0, batch_nodes, -1)   # (tensor[index] = scalar blocks the host until the GPU has drained)
        ring.tick()
        return out

    # ---- evaluate_model_utils.py:38-142 (call under torch.no_grad())
    def eval_iteration(self, batch_idx: int, src, dst, ts, eid, neg_src, neg_dst, lookahead=None):
        bb, ring = self.backbone, self.ring
        if ring.sparse:
            ring.apply_advance()             # (a slide left pending by a replayed training iteration)
        if (self._ring_on_device and ring.dev_start is None and ring.sparse and ring.len == ring.T and ring._advance is None):
            ring.position_on_device()
        on_device = self.device_counts and ring.len > 0
        if on_device:
            batch_nodes, n_live, presorted = self.batch_nodes_device(src, dst)
        else:
            n_live = None
            batch_nodes, presorted = self.batch_nodes_and_segments(src, dst)    # sorted unique, like the reference's torch.unique
        self._batch_groups = None
        cur, _ = self._splice(batch_nodes, batch_idx, live=n_live)
        self.slot_of.index_fill_(0, batch_nodes, -1)
        n = src.numel()
        # (the first 2 B rows are cat[src, dst], grouped by node by the batch-node grouping above: hub nodes' node channel, csrc/hub.hip)
        groups = (presorted[1], presorted[0]) if (presorted[0].dtype == torch.int32 and presorted[1] is not None) else None
        emb_p = bb.combining_pe_raw_feat(cur, torch.cat([src, dst, neg_src, neg_dst]), torch.cat([ts, ts, ts, ts]), self.K, self.G, padded=True,
                                         row_blocks=4, row_groups=groups)
        if self.predictor.fused_ok(emb_p):      # both predictor calls of evaluate_model_utils.py:100-101 in one launch, no concatenation
            predicts = self.predictor.pair_logits(emb_p, n, (0, n, 2 * n, 3 * n)).sigmoid().clamp(0, 1)
        else:
            emb = emb_p[:, :bb.feat_dim]
            predicts = torch.cat([self._probabilities(emb[:n], emb[n:2 * n]), self._probabilities(emb[2 * n:3 * n], emb[3 * n:])], dim=0)
        labels = self._labels(n)
        if lookahead is not None:
            self.prefetch_batch_nodes(*lookahead[:2])      # (a longer tuple also names the next batch's times / negatives: lstep_amd.parallel)
        if on_device:
            base, ref = ring.building_ref()
            bb.update_pe_device(cur, batch_nodes, n_live, src, dst, ts, self.K, presorted, changed=ring.written,
                                mirror=base if ref is not None else ring.building(), mirror_ring=ref)
        else:
            bb.update_pe(pe=cur, node_ids=batch_nodes, edge_ids=eid, batch_src_node_ids=src, batch_dst_node_ids=dst,
                         node_interact_times=ts, current_time=ts.max(), num_neighbors=self.K, time_gap=self.G, presorted=presorted,
                         changed=ring.written, mirror=ring.building())
        ring.commit()
        ring.apply_advance()
        ring.tick()
        return {"loss": F.binary_cross_entropy(predicts, labels), "predicts": predicts}
