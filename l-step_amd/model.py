"""MI355X-backed L-STEP backbone with the reference's method surface.

Drop-in for reference ``models.LSTEP.LSTEP`` (``models/LSTEP.py:28-340``) as it is used by
``train_LSTEP_link_prediction.py:127-142,198,228-295`` and ``evaluate_model_utils.py:28,61-129``:

* same constructor arguments, same parameter names / shapes / dtypes (reference checkpoints load with
  ``load_state_dict``), still an ``nn.Module`` that can sit in ``nn.Sequential(backbone, MergeLayer)``;
* same methods: ``set_neighbor_sampler``, ``fourier_transform_pe``, ``aggregated_node_embeddings``,
  ``compute_neighborhood_pe``, ``combining_pe_raw_feat``, ``update_pe`` (mutates ``pe`` in place, returns it);
* additionally ``compute_src_dst_node_temporal_embeddings`` (the DyGLib-style wrapper BASELINE.json names).

The sparse / irregular work (temporal search, row gathers, time encoding, segmented message sums, the history
stream of the FFT filter) runs in hand-written HIP kernels behind the C ABI of ``include/lstep_hip.h``; the small
dense projections stay ``torch.nn.functional.linear`` (rocBLAS/hipBLASLt fp32).  There is no CPU fallback.

Numerics: fp32 throughout, like the reference.  Two reassociations (both exact in real arithmetic, <= 1e-6 in fp32):
``edge_agg`` is applied before ``edge_mlp_1`` (no non-linearity between them, ``models/LSTEP.py:161-164``) and the FFT
filter is applied as its equivalent real ``[T, P]`` coefficient table (everything between the history gather and
``fft_agg`` is linear, ``models/LSTEP.py:116-135``).
"""
from __future__ import annotations

import contextlib
import ctypes
import gc
import math
import os
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as nat


def _round16(x: int) -> int:
    return (x + 15) // 16 * 16


def _pad2(w: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """Zero-pad a weight matrix to [rows, cols] (autograd slices the gradient back)."""
    return F.pad(w, (0, cols - w.shape[1], 0, rows - w.shape[0]))


def _pad1(b: torch.Tensor, n: int) -> torch.Tensor:
    return F.pad(b, (0, n - b.shape[0]))


def _addmm_relu(b, x, wt):
    """relu(x @ wt + b) with the relu in the GEMM epilogue when this torch build exposes it."""
    fused = getattr(torch, "_addmm_activation", None)
    return fused(b, x, wt) if fused is not None else torch.relu_(torch.addmm(b, x, wt))


class _Linear(torch.autograd.Function):
    """``x @ w.T + b`` whose weight gradient is a batched split-M GEMM.

    The weight gradient ``dY^T X`` reduces over M = 3 * batch rows into a small [N, K] output; hipBLASLt's fp32 kernels for
    that shape reach 30-60 TFLOP/s on MI355X, the same product as 32 row-chunk ``bmm`` + one ``sum`` reaches 65-105
    (tools/gemm_dw.py).  Forward and dX are the plain library GEMMs."""

    NATIVE_WGRAD = os.environ.get("LSTEP_TORCH_WGRAD", "0") != "1"   # A/B switch: 1 = library GEMMs for the weight gradient
    CHUNKS = int(os.environ.get("LSTEP_DW_CHUNKS", "16"))   # tuning knob: 16-32 row chunks are best (tools/gemm_dw.py; 16 in the full step)

    @staticmethod
    def forward(ctx, x, w, b, relu):
        ctx.has_bias = b is not None
        ctx.relu = bool(relu)
        if ctx.relu:  # relu in the GEMM epilogue (hipBLASLt), one launch instead of two
            y = _addmm_relu(b, x, w.t())
            ctx.save_for_backward(x, w, y)
            return y
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        if ctx.relu:
            x, w, y = ctx.saved_tensors
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)
        else:
            x, w = ctx.saved_tensors
            dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dy @ w
        if ctx.needs_input_grad[1]:
            m = x.shape[0]
            if _Linear.NATIVE_WGRAD and x.is_cuda and x.stride(1) == 1 and w.shape[0] >= 64:
                # hand-written fp32-MFMA kernel: dW and db in one pass over dy and x (lstep_linear_wgrad)
                dw, db = nat.linear_wgrad(dy, x, want_bias=ctx.has_bias and ctx.needs_input_grad[2])
                return dx, dw, db, None
            c = math.gcd(m, _Linear.CHUNKS)
            if c >= 4 and m // c >= 256 and x.is_contiguous():
                dw = torch.bmm(dy.view(c, m // c, -1).transpose(1, 2), x.view(c, m // c, -1)).sum(dim=0)
            else:
                dw = dy.t() @ x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=0)
        return dx, dw, db, None


def fast_linear(x, w, b=None, relu=False):
    """``relu(x @ w.T + b)`` if ``relu`` else ``x @ w.T + b`` (``relu`` needs a bias)."""
    return _Linear.apply(x, w, b, relu)


def _mm(a, b, out=None):
    """Weight-sized product: the native one-wave-per-tile kernel on the GPU (the library runs these as a single workgroup, 50-60 us
    each), torch.mm elsewhere."""
    if a.is_cuda and os.environ.get("LSTEP_TORCH_SMALL_MM") != "1":
        return nat.small_mm(a, b, out=out)
    if out is None:
        return a @ b
    out.copy_(a @ b)
    return out


def _native_compose(t) -> bool:
    """The weight composition's non-product work as one native launch per direction (csrc/compose.hip); LSTEP_TORCH_COMPOSE=1 (and
    every CPU tensor) takes the framework ops."""
    return t.is_cuda and os.environ.get("LSTEP_TORCH_COMPOSE") != "1"


def _tail_weights_forward(dims, W1, b1, aw, ab, W2, b2, Wn, bn, Wo, bo, Ws, bs, Wn1, bn1, Wn2, bn2):
    """Padded / pre-multiplied weights of the dense tail (see ``_TailWeights``).  Returns (the 8 operands, their 4 transposes for the
    backward kernel, (a_sum, M) for the hand-derived backward)."""
    Fd, C, P, CP, Ce, Fn, Cp, Pp = dims   # F, D+F, P, P+D and their 16-aligned paddings
    dev = W1.device
    sizes = [Ce * Ce, Ce, Pp * Cp, Pp, Pp * 2 * Pp, Pp, Fn * (Fn + Ce + Pp), Fn]
    if _native_compose(W1):
        # three products + ONE launch for everything else (csrc/compose.hip) instead of ~20 framework launches
        params = [t.detach().contiguous() for t in (W1, b1, aw, ab, W2, b2, Wn, bn, Wo, bo, Ws, bs, Wn1, bn1, Wn2, bn2)]
        W2c, Wnc, Woc = params[4], params[6], params[8]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        flat_t = torch.empty(sizes[0] + sizes[2] + sizes[4] + sizes[6], dtype=torch.float32, device=dev)
        a_sum = torch.empty(1, dtype=torch.float32, device=dev)
        M = torch.empty((Fd, C), dtype=torch.float32, device=dev)
        parts = torch.split(flat, sizes)
        Wall = parts[6].view(Fn, Fn + Ce + Pp)
        nat.small_mm(Woc[:, :Fd], Wnc[:, Fd:], out=M)
        nat.small_mm(Woc[:, :Fd], Wnc[:, :Fd], out=Wall[:Fd, :Fd])
        nat.small_mm(M, W2c, out=Wall[:Fd, Fn:Fn + C])
        lib = nat.load_library()
        with torch.cuda.device(dev):
            nat.check(lib.lstep_tail_weights_pack((ctypes.c_void_p * 16)(*[t.data_ptr() for t in params]), nat.ptr(M),
                                                  (ctypes.c_int32 * 9)(Fd, C, P, CP, Ce, Fn, Cp, Pp, aw.numel()), nat.ptr(flat), nat.ptr(flat_t),
                                                  nat.ptr(a_sum), nat.current_stream()))
        tparts = torch.split(flat_t, [sizes[0], sizes[2], sizes[4], sizes[6]])
        outs = (parts[0].view(Ce, Ce), parts[1], parts[2].view(Pp, Cp), parts[3], parts[4].view(Pp, 2 * Pp), parts[5], Wall, parts[7])
        transposed = (tparts[0].view(Ce, Ce), tparts[1].view(Cp, Pp), tparts[2].view(2 * Pp, Pp), tparts[3].view(Fn + Ce + Pp, Fn))
        return outs, transposed, (a_sum, M)
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    parts = torch.split(flat, sizes)
    W1p, b1p, Wn1p, bn1p = parts[0].view(Ce, Ce), parts[1], parts[2].view(Pp, Cp), parts[3]
    Wq, bq, Wall, constp = parts[4].view(Pp, 2 * Pp), parts[5], parts[6].view(Fn, Fn + Ce + Pp), parts[7]
    a_sum = aw.sum()
    W1p[:C, :C] = W1
    torch.addcmul(ab.expand(C), a_sum.expand(C), b1, out=b1p[:C])
    Wn1p[:P, :CP] = Wn1
    bn1p[:P] = bn1
    Wq[:P, :P] = Ws
    Wq[:P, Pp:Pp + P] = Wn2
    torch.add(bs, bn2, out=bq[:P])
    wo_a, wo_b = Wo[:, :Fd], Wo[:, Fd:]
    wn_a, wn_b = Wn[:, :Fd], Wn[:, Fd:]
    M = _mm(wo_a, wn_b)                                       # [F, C]
    _mm(wo_a, wn_a, out=Wall[:Fd, :Fd])
    _mm(M, W2, out=Wall[:Fd, Fn:Fn + C])
    Wall[:Fd, Fn + Ce:Fn + Ce + P] = wo_b
    constp[:Fd] = torch.addmv(torch.addmv(bo, M, b2), wo_a, bn)
    transposed = (W1p.t().contiguous(), Wn1p.t().contiguous(), Wq.t().contiguous(), Wall.t().contiguous())
    return (W1p, b1p, Wn1p, bn1p, Wq, bq, Wall, constp), transposed, (a_sum, M)


def _tail_weights_backward(dims, K, b1, W2, b2, Wn, bn, Wo, a_sum, M, gW1p, gb1p, gWn1p, gbn1p, gWq, gbq, gWall, gconst, dense=False):
    """Hand-derived chain rule of ``_tail_weights_forward``: gradients of the 16 parameters in its argument order."""
    Fd, C, P, CP, Ce, Fn, Cp, Pp = dims
    if _native_compose(W2):
        # one product, ONE launch for every slice copy / bias / rank-1 term (csrc/compose.hip), five products: seven launches instead of ~20
        dev = W2.device
        gin = [g.contiguous() for g in (gW1p, gb1p, gWn1p, gbn1p, gWq, gbq, gWall, gconst)]
        gWall_c = gin[6]
        W2c, Wnc, Woc = W2.detach().contiguous(), Wn.detach().contiguous(), Wo.detach().contiguous()
        shapes = [(C, C), (C,), (1, K), (1,), (C, C), (C,), (Fd, Fd + C), (Fd,), (Fd, Fd + P), (Fd,), (P, P), (P,), (P, CP), (P,), (P, P), (P,)]
        numels = [int(np.prod(sh)) for sh in shapes]
        flat = torch.empty(sum(numels), dtype=torch.float32, device=dev)
        (d_W1, d_b1, d_aw, d_ab, d_W2, d_b2, d_Wn, d_bn, d_Wo, d_bo, d_Ws, d_bs, d_Wn1, d_bn1, d_Wn2, d_bn2) = \
            [t.view(sh) for t, sh in zip(torch.split(flat, numels), shapes)]
        dA1, dA2 = gWall_c[:Fd, :Fd], gWall_c[:Fd, Fn:Fn + C]
        dM = nat.small_mm(dA2, W2c.t())
        fwd = [t.detach().contiguous() for t in (b1, b2, bn)] + [Woc, M.contiguous(), a_sum.reshape(1).contiguous()]
        outs14 = (d_W1, d_b1, d_aw, d_ab, d_b2, d_bn, d_Wo, d_bo, d_Ws, d_bs, d_Wn1, d_bn1, d_Wn2, d_bn2)
        lib = nat.load_library()
        with torch.cuda.device(dev):
            nat.check(lib.lstep_tail_weights_unpack((ctypes.c_void_p * 8)(*[t.data_ptr() for t in gin]), (ctypes.c_void_p * 6)(*[t.data_ptr() for t in fwd]),
                                                    (ctypes.c_void_p * 14)(*[t.data_ptr() for t in outs14]), nat.ptr(dM),
                                                    (ctypes.c_int32 * 9)(Fd, C, P, CP, Ce, Fn, Cp, Pp, K), nat.current_stream()))
        nat.small_mm(M.t(), dA2, out=d_W2)
        nat.small_mm(dA1, Wnc[:, :Fd].t(), out=d_Wo[:, :Fd], beta=1.0)
        nat.small_mm(dM, Wnc[:, Fd:].t(), out=d_Wo[:, :Fd], beta=1.0)
        nat.small_mm(Woc[:, :Fd].t(), dA1, out=d_Wn[:, :Fd])
        nat.small_mm(Woc[:, :Fd].t(), dM, out=d_Wn[:, Fd:])
        return (d_W1, d_b1, d_aw, d_ab, d_W2, d_b2, d_Wn, d_bn, d_Wo, d_bo, d_Ws, d_bs, d_Wn1, d_bn1, d_Wn2, d_bn2)
    db1e = gb1p[:C]
    d_b1 = a_sum * db1e
    d_aw = torch.dot(db1e, b1).expand(1, K)
    d_ab = db1e.sum().reshape(1)
    wo_a = Wo[:, :Fd]
    wn_a, wn_b = Wn[:, :Fd], Wn[:, Fd:]
    dA1, dA2, dWo_b, dc = gWall[:Fd, :Fd], gWall[:Fd, Fn:Fn + C], gWall[:Fd, Fn + Ce:Fn + Ce + P], gconst[:Fd]
    dM = torch.addr(_mm(dA2, W2.t()), dc, b2)                 # from A2 = M W2 and const = M b2 + ...
    d_W2 = _mm(M.t(), dA2)
    d_b2 = M.t() @ dc
    d_Wo = torch.empty_like(Wo)
    d_Wo[:, Fd:] = dWo_b
    dWo_a = d_Wo[:, :Fd]
    _mm(dA1, wn_a.t(), out=dWo_a)
    if dM.is_cuda and os.environ.get("LSTEP_TORCH_SMALL_MM") != "1":
        nat.small_mm(dM, wn_b.t(), out=dWo_a, beta=1.0)
    else:
        dWo_a += dM @ wn_b.t()
    dWo_a.addr_(dc, bn)
    d_Wn = torch.empty_like(Wn)
    _mm(wo_a.t(), dA1, out=d_Wn[:, :Fd])
    _mm(wo_a.t(), dM, out=d_Wn[:, Fd:])
    d_bn = wo_a.t() @ dc
    out = (gW1p[:C, :C], d_b1, d_aw, d_ab, d_W2, d_b2, d_Wn, d_bn, d_Wo, dc, gWq[:P, :P], gbq[:P], gWn1p[:P, :CP],
           gbn1p[:P], gWq[:P, Pp:Pp + P], gbq[:P])
    if dense:   # the graph path assigns these to .grad itself: dense, parameter-shaped, pairwise distinct storage
        seen, res = set(), []
        for t in out:
            if not t.is_contiguous() or t.data_ptr() in seen:
                t = t.clone(memory_format=torch.contiguous_format)
            seen.add(t.data_ptr())
            res.append(t)
        out = tuple(res)
    return out


def _tail_grad_shapes(dims):
    Fd, C, P, CP, Ce, Fn, Cp, Pp = dims
    return [(Ce, Ce), (Ce,), (Pp, Cp), (Pp,), (Pp, 2 * Pp), (Pp,), (Fn, Fn + Ce + Pp), (Fn,)]


class _TailWeights(torch.autograd.Function):
    """Padded / pre-multiplied weights of the dense tail in ~25 launches forward and ~20 backward.

    Built from the raw parameters every call (they change every optimiser step):
        W1p  [Ce, Ce]  = edge_mlp_1.weight                      b1p  = (sum a) * edge_mlp_1.bias + edge_agg.bias
        Wn1p [Pp, Cp]  = pe_neighbor_mlp_1.weight               bn1p = pe_neighbor_mlp_1.bias
        Wq   [Pp, 2Pp] = [self_update_neighbor_pe.weight | pe_neighbor_mlp_2.weight]     bq = sum of their biases
        Wall [Fn, Fn+Ce+Pp] = [Wo_a Wn_a | Wo_a Wn_b W2 | Wo_b]  const = Wo_a Wn_b b2 + Wo_a bn + bo
    (Wo = out_node_emb.weight = [Wo_a | Wo_b], Wn = node_mlp.weight = [Wn_a | Wn_b], W2/b2 = edge_mlp_2), all zero-padded
    to the 16-aligned widths, plus the four transposes the backward kernel reads.  The same thing written with F.pad / cat / matmul
    costs ~45 launches forward and ~60 in autograd's backward; here the backward is the hand-derived chain rule of the products above.
    ``_TailWeightsGraph`` replays both directions as one HIP graph launch each."""

    @staticmethod
    def forward(ctx, dims, W1, b1, aw, ab, W2, b2, Wn, bn, Wo, bo, Ws, bs, Wn1, bn1, Wn2, bn2):
        ctx.set_materialize_grads(False)     # (no zero tensors for the outputs nobody differentiates: each would be a fill launch in backward)
        outs, transposed, (a_sum, M) = _tail_weights_forward(dims, W1, b1, aw, ab, W2, b2, Wn, bn, Wo, bo, Ws, bs, Wn1, bn1, Wn2, bn2)
        ctx.dims = dims
        ctx.save_for_backward(b1, a_sum, W2, b2, Wn, bn, Wo, M)
        ctx.K = aw.numel()
        ctx.mark_non_differentiable(*transposed)
        return outs + transposed

    @staticmethod
    def backward(ctx, gW1p, gb1p, gWn1p, gbn1p, gWq, gbq, gWall, gconst, *_):
        b1, a_sum, W2, b2, Wn, bn, Wo, M = ctx.saved_tensors
        dev = b1.device
        grads = [g if g is not None else torch.zeros(shape, dtype=torch.float32, device=dev)
                 for g, shape in zip((gW1p, gb1p, gWn1p, gbn1p, gWq, gbq, gWall, gconst), _tail_grad_shapes(ctx.dims))]
        return (None,) + _tail_weights_backward(ctx.dims, ctx.K, b1, W2, b2, Wn, bn, Wo, a_sum, M, *grads)


class _TailWeightsGraph:
    """``_TailWeights`` as two HIP graphs (``torch.cuda.CUDAGraph``): the ~25 + ~20 tiny launches that turn the live parameters into
    the dense tail's operands, and the operand gradients back into parameter gradients, are functions of fixed-address tensors with
    fixed shapes, so each direction is captured once and replayed with one launch per step.  The operand gradients arrive in
    fixed buffers (``grad_buffers``: ``lstep_linear_wgrad`` writes straight into them); the parameter gradients are static tensors
    too and are assigned to ``param.grad`` directly (returned through autograd they would be cloned one by one)."""

    def __init__(self, dims, params):
        self.dims, self.params = dims, list(params)
        dev = self.params[0].device
        self.gin = [torch.zeros(shape, dtype=torch.float32, device=dev) for shape in _tail_grad_shapes(dims)]
        self.g_fwd = self.g_bwd = None
        self.ptrs = None
        self.live = 0       # forward results whose backward has not run yet: the static buffers serve ONE autograd node at a time
        self.aux = None     # stream the backward replay runs on (set per forward by LSTEP._combined_tail)
        self.prepared = None   # event of a forward replay issued ahead of time

    def _detached(self):
        return [p.detach() for p in self.params]

    def close(self):
        """Give the two captured graphs back (destroyed at the next safe point, ``drain_dead_graphs``); the object re-captures on its next use."""
        for g in (self.g_fwd, self.g_bwd):
            if g is not None:
                retire_graph(g)
        self.g_fwd = self.g_bwd = self.ptrs = self.prepared = None
        self.__dict__.pop("outs", None), self.__dict__.pop("transposed", None), self.__dict__.pop("saved", None), self.__dict__.pop("pgrads", None)
        drain_dead_graphs()

    def forward(self):
        if torch.cuda.is_current_stream_capturing():
            # the whole iteration is being captured (engine.GraphedTrainStep): no graph inside a graph, the ~25 launches are recorded
            # as they are; the fixed gradient buffers and the deferred backward keep working
            pre = self.__dict__.pop("_cap_prepared", None)
            if pre is not None:        # issued ahead of time on the auxiliary stream (``prepare``): a parallel branch of the graph
                outs, transposed, self._cap_saved, ready = pre
                torch.cuda.current_stream().wait_event(ready)
                return outs, transposed
            with torch.no_grad():
                outs, transposed, self._cap_saved = _tail_weights_forward(self.dims, *self._detached())
            return outs, transposed
        ptrs = tuple(p.data_ptr() for p in self.params)
        if self.g_fwd is None or ptrs != self.ptrs:     # first use, or a parameter's storage was replaced: (re)capture
            if self.g_bwd is not None:
                retire_graph(self.g_bwd)
            self.ptrs, self.g_bwd = ptrs, None
            with torch.no_grad():
                _tail_weights_forward(self.dims, *self._detached())   # warm-up outside the capture (library handles, workspaces)
                torch.cuda.current_stream().synchronize()
                if self.g_fwd is not None:
                    retire_graph(self.g_fwd)
                self.g_fwd = new_graph(self)
                with _no_gc(), torch.cuda.graph(self.g_fwd, stream=nat.role_stream(self.params[0].device, "capture"), capture_error_mode="thread_local"):
                    self.outs, self.transposed, self.saved = _tail_weights_forward(self.dims, *self._detached())
        if self.prepared is not None:        # replayed ahead of time on the auxiliary stream (LSTEP.prepare_step): just wait for it
            torch.cuda.current_stream().wait_event(self.prepared)
            self.prepared = None
        else:
            self.g_fwd.replay()
        return self.outs, self.transposed

    def prepare(self, aux):
        """Replay the forward graph on ``aux`` now (it only depends on the parameters); the next ``forward()`` waits for it."""
        if torch.cuda.is_current_stream_capturing():
            if self.live > 0:
                return
            aux.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(aux), torch.no_grad():
                outs, transposed, saved = _tail_weights_forward(self.dims, *self._detached())
                ready = torch.cuda.Event()
                ready.record()
            self._cap_prepared = (outs, transposed, saved, ready)
            return
        if self.g_fwd is None or tuple(p.data_ptr() for p in self.params) != self.ptrs or self.live > 0:
            return
        aux.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(aux):
            self.g_fwd.replay()
            self.prepared = torch.cuda.Event()
            self.prepared.record()

    def backward(self):
        """Operand gradients in ``self.gin`` -> the 16 parameter gradients (static tensors, valid until the next call)."""
        if torch.cuda.is_current_stream_capturing():
            d = self._detached()
            W1, b1, aw, ab, W2, b2, Wn, bn, Wo, bo = d[:10]
            a_sum, M = self._cap_saved
            with torch.no_grad():
                return _tail_weights_backward(self.dims, aw.numel(), b1, W2, b2, Wn, bn, Wo, a_sum, M, *self.gin, True)
        if self.g_bwd is None:
            d = self._detached()
            W1, b1, aw, ab, W2, b2, Wn, bn, Wo, bo = d[:10]
            a_sum, M = self.saved
            args = (self.dims, aw.numel(), b1, W2, b2, Wn, bn, Wo, a_sum, M, *self.gin, True)
            with torch.no_grad():
                keep = [g.clone() for g in self.gin]
                _tail_weights_backward(*args)
                torch.cuda.current_stream().synchronize()
                self.g_bwd = new_graph(self)
                with _no_gc(), torch.cuda.graph(self.g_bwd, stream=nat.role_stream(self.params[0].device, "capture"), capture_error_mode="thread_local"):
                    self.pgrads = _tail_weights_backward(*args)
                for g, k in zip(self.gin, keep):
                    g.copy_(k)
        self.g_bwd.replay()
        return self.pgrads


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a stream is capturing (second line of defence; the first is ``new_graph`` below: no
    ``torch.cuda.CUDAGraph`` of this package is ever destroyed by the collector or by a reference count reaching zero)."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


# ---- explicit lifetime of captured graphs.  Destroying a ``torch.cuda.CUDAGraph`` is a HIP call the runtime refuses while a stream of the
# calling thread is capturing, and its destructor turns the refusal into an abort.  Left to Python, the destruction happens wherever the
# last reference dies: a dead model's graphs used to die inside whatever code the cyclic collector interrupted -- once, inside another
# model's capture on the autograd thread (round 2, gpurun_out/t3.log).  So every graph this package captures is OWNED by the registry below;
# its user only borrows it.  A graph is destroyed (``reset()``) in ``drain_dead_graphs`` alone, which runs at the package's safe points
# (before a capture starts, at the start of an engine iteration, in ``close()``) and never while a stream is capturing.
_GRAPH_REGISTRY = []      # [weak reference to the owner, graph, retired?]


def new_graph(owner) -> "torch.cuda.CUDAGraph":
    """A ``CUDAGraph`` whose lifetime is explicit: it lives until ``retire_graph`` / the death of ``owner`` AND the next drain."""
    import weakref
    drain_dead_graphs()
    g = torch.cuda.CUDAGraph()
    _GRAPH_REGISTRY.append([weakref.ref(owner), g, False])
    return g


def retire_graph(g):
    """The owner is done with ``g`` (close(), re-capture): it is destroyed at the next safe point."""
    for entry in _GRAPH_REGISTRY:
        if entry[1] is g:
            entry[2] = True


def drain_dead_graphs() -> int:
    """Destroy the graphs that were retired or whose owner is gone -- unless a capture is under way.  Returns how many were destroyed."""
    if not _GRAPH_REGISTRY or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        return 0
    keep, dead = [], []
    for entry in _GRAPH_REGISTRY:
        (dead if (entry[2] or entry[0]() is None) else keep).append(entry)
    _GRAPH_REGISTRY[:] = keep
    for entry in dead:
        entry[1].reset()
    return len(dead)


def live_graph_count() -> int:
    return len(_GRAPH_REGISTRY)


class _LiveToken:
    """Marks a ``_TailWeightsGraph`` as in use from a forward until its backward has run, or until the autograd node is dropped."""

    def __init__(self, tw):
        self.tw = tw
        tw.live += 1

    def release(self):
        if self.tw is not None:
            self.tw.live -= 1
            self.tw = None

    __del__ = release


class _TailWeightsReplay(torch.autograd.Function):
    """Autograd face of ``_TailWeightsGraph``."""

    @staticmethod
    def forward(ctx, tw, *params):
        ctx.set_materialize_grads(False)     # (no zero tensors for the outputs nobody differentiates: each would be a fill launch in backward)
        outs, transposed = tw.forward()
        ctx.tw = tw
        ctx.token = _LiveToken(tw)
        res = tuple(t.detach() for t in outs + transposed)   # fresh tensor objects over the static storage
        ctx.mark_non_differentiable(*res[len(outs):])
        return res

    @staticmethod
    def backward(ctx, *grads):
        tw = ctx.tw
        aux = getattr(tw, "aux", None)
        if aux is not None:      # the operand gradients are produced on the auxiliary stream (postponed): queue up behind them
            def on_aux():
                with torch.cuda.stream(aux):
                    _TailWeightsReplay._backward(ctx, tw, grads, on_aux_stream=True)
            _defer(tw.params[0].device, on_aux)
            return (None,) * (1 + len(tw.params))
        return _TailWeightsReplay._backward(ctx, tw, grads)

    @staticmethod
    def _backward(ctx, tw, grads, on_aux_stream=False):
        for buf, g in zip(tw.gin, grads[:len(tw.gin)]):
            if g is None:
                buf.zero_()
            elif g.data_ptr() != buf.data_ptr():
                buf.copy_(g)
        for i, (p, g) in enumerate(zip(tw.params, tw.backward())):
            if i == 2 and on_aux_stream:
                # edge_agg.weight also receives a gradient through autograd (the gather stage, on the critical stream).  Autograd adds
                # into an existing .grad IN PLACE on its own stream: if .grad were this static buffer, which the auxiliary stream
                # fills later, that contribution would be overwritten.  The two parts meet in LSTEP.join_aux_stream instead.
                _PENDING_AUX_GRADS.setdefault(p.device, []).append((p, g))
                continue
            if p.grad is None:
                p.grad = g.detach()
            else:
                p.grad = p.grad + g      # out of place: an earlier contribution may be a broadcast view or shared with autograd
        ctx.token.release()
        return (None,) * (1 + len(tw.params))


_AUX_STREAMS = {}
_SIDE_STREAMS = {}        # device -> stream for HBM-bound parameter-gradient work that should not queue behind the matrix-core products
_PENDING_AUX_GRADS = {}   # device -> [(parameter, gradient part produced on the auxiliary stream)]: added in LSTEP.join_aux_stream


def _aux_stream(dev):
    """Second stream for the weight-gradient products (matrix-core bound, no consumer inside the backward pass): they run beside the
    HBM-bound rest of the backward (gather backward, gradient sorts, history filter backward) instead of in front of it."""
    dev = torch.device(dev)
    st = _AUX_STREAMS.get(dev)
    if st is None:
        st = _AUX_STREAMS[dev] = nat.role_stream(dev, "aux")
    return st


def _side_stream(dev):
    """Third stream of the backward pass: the edge-row re-gather behind d(edge_agg.weight) (HBM-bound, 0.3 ms at the bench shape, no
    consumer before the optimiser).  On the auxiliary stream it would wait for ~1 ms of weight-gradient products; beside them it fills
    the memory system the matrix-core kernels leave idle.  Joined by ``LSTEP.join_aux_stream``."""
    dev = torch.device(dev)
    st = _SIDE_STREAMS.get(dev)
    if st is None:
        st = _SIDE_STREAMS[dev] = nat.role_stream(dev, "side")
    return st


_AUX_PARAM_EVENT = {}   # device -> event recorded behind an optimiser step that ran on the auxiliary stream (see ``LstepEngine``)


def note_aux_param_update(dev, event):
    """The parameters whose gradients the auxiliary stream produces (dense tail, link predictor) were updated ON that stream: whoever
    reads them next on another stream waits for ``event`` first (``wait_aux_param_update``)."""
    _AUX_PARAM_EVENT[torch.device(dev)] = event


def wait_aux_param_update(dev):
    ev = _AUX_PARAM_EVENT.pop(torch.device(dev), None)
    if ev is not None:
        torch.cuda.current_stream(dev).wait_event(ev)


_PENDING_HEAD_WGRADS = {}   # device -> (items, results): the link predictor's two weight-gradient products waiting for the dense tail's batched launch
_DEFERRED = {}     # device -> callables: auxiliary-stream work whose launch is postponed until the critical kernels are out


def _defer(dev, fn, front: bool = False):
    q = _DEFERRED.setdefault(torch.device(dev), [])
    if front:       # (the products must precede the work that consumes them, which an earlier node of the backward pass may have queued)
        q.insert(0, fn)
    else:
        q.append(fn)


def _flush_deferred(dev=None):
    """Launch the postponed auxiliary-stream work (weight-gradient products, replayed weight-composition backward).  Called right
    after the gather backward kernel has been launched -- the host work of these launches (~0.3 ms) would otherwise sit between the
    dense tail's backward kernel and the gather backward kernel on the critical stream -- and, as a safety net, by ``join_aux_stream``."""
    for d in ([torch.device(dev)] if dev is not None else list(_DEFERRED)):
        q = _DEFERRED.get(d)
        while q:
            q.pop(0)()


class _FusedTail(torch.autograd.Function):
    """All dense layers after the gather stage as one launch per direction (``lstep_tail_fwd`` / ``lstep_tail_bwd``, fp32 matrix
    cores) plus four ``lstep_linear_wgrad`` products, instead of ~15 + ~40 library launches.  ``cat1`` = [x_node | . | .] and
    ``cat2`` = [own | . ] arrive from the gather stage with their first block filled; the kernels write h1 / q / p1 into the rest."""

    @staticmethod
    def forward(ctx, x_edge, x_pe, cat1, cat2, W1p, b1p, Wn1p, bn1p, Wq, bq, Wall, constp, w1t, wn1t, wqt, wallt, grad_buffers, aux):
        ctx.set_materialize_grads(False)     # (no zero tensors for the outputs nobody differentiates: each would be a fill launch in backward)
        lib = nat.load_library()
        m = x_edge.shape[0]
        out = torch.empty((m, Wall.shape[0]), dtype=torch.float32, device=x_edge.device)
        ws = [t.contiguous() for t in (W1p, b1p, Wn1p, bn1p, Wq, bq, Wall, constp)]
        with torch.cuda.device(x_edge.device):
            nat.check(lib.lstep_tail_fwd(nat.ptr(x_edge), x_edge.stride(0), nat.ptr(x_pe), x_pe.stride(0), nat.ptr(cat1), nat.ptr(cat2),
                                         nat.ptr(out), *[nat.ptr(t) for t in ws], m, nat.current_stream()))
        ctx.save_for_backward(x_edge, x_pe, cat1, cat2, w1t, wn1t, wqt, wallt)
        ctx.grad_buffers = grad_buffers     # optional fixed [dW, db] x 4 destinations (``_TailWeightsGraph.gin``)
        ctx.aux = aux                       # optional stream for the weight-gradient products (only with grad_buffers)
        return out

    @staticmethod
    def backward(ctx, g_out):
        if g_out is None:
            return (None,) * 18
        lib = nat.load_library()
        x_edge, x_pe, cat1, cat2, w1t, wn1t, wqt, wallt = ctx.saved_tensors
        m, dev = x_edge.shape[0], x_edge.device
        g_out = g_out.contiguous()
        Ce, Cp, Pp = w1t.shape[1], wn1t.shape[0], wqt.shape[1]
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        d_xe, d_xp, d_own = new(m, w1t.shape[0]), new(m, Cp), new(m, cat2.shape[1])
        d_h1, d_p1, d_z = new(m, Ce), new(m, wn1t.shape[1]), new(m, Pp)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_tail_bwd(nat.ptr(g_out), nat.ptr(cat1), nat.ptr(cat2), nat.ptr(w1t), nat.ptr(wn1t), nat.ptr(wqt), nat.ptr(wallt),
                                         nat.ptr(d_xe), nat.ptr(d_xp), nat.ptr(d_own), d_own.stride(0), nat.ptr(d_h1), nat.ptr(d_p1),
                                         nat.ptr(d_z), m, nat.current_stream()))
        gb = ctx.grad_buffers or [None] * 8

        def weight_gradients():
            # the four products in ONE (partial, reduce) launch pair (lstep_linear_wgrad_batch); a link predictor that ran its backward just
            # before (``_Head.backward``, engine mode) has left its two products here to ride along
            # (the predictor's two products ride along only when they were parked for THIS auxiliary stream: a tail whose products run on
            # the main stream -- no auxiliary stream, LSTEP_NO_GRAPH=1, a weight graph still in use -- leaves them to the predictor's own
            # deferred launch, which waits for the right events)
            pend = _PENDING_HEAD_WGRADS.get(dev)
            riders = None
            if pend is not None and ctx.aux is not None and pend[2] is not None and pend[2].cuda_stream == ctx.aux.cuda_stream:
                riders = _PENDING_HEAD_WGRADS.pop(dev)
            items = [(d_h1, x_edge[:, :w1t.shape[0]], True, (gb[0], gb[1])), (d_p1, x_pe[:, :Cp], True, (gb[2], gb[3])),
                     (d_z, cat2, True, (gb[4], gb[5])), (g_out, cat1, True, (gb[6], gb[7]))]
            res = nat.linear_wgrad_batch(items + (riders[0] if riders else []))
            if riders:
                riders[1].extend(res[4:])
            (gW1, gb1), (gWn1, gbn1), (gWq, gbq), (gWall, gconst) = res[:4]
            return gW1, gb1, gWn1, gbn1, gWq, gbq, gWall, gconst

        if ctx.aux is not None:
            # engine mode: the four products (and, right behind them, the replayed backward of the weight composition) go to the
            # auxiliary stream, and their launch is postponed (``_flush_deferred``); the destinations are the fixed buffers, so the
            # gradients can be handed to autograd now.  The caller joins the stream before the optimiser step (LSTEP.join_aux_stream).
            aux = ctx.aux
            ready = torch.cuda.Event()
            ready.record()

            def products():
                with torch.cuda.stream(aux):
                    aux.wait_event(ready)
                    weight_gradients()
                for t in (d_h1, d_p1, d_z, g_out, x_edge, x_pe, cat1, cat2):
                    t.record_stream(aux)

            if os.environ.get("LSTEP_WGRAD_LATE") != "1":
                # the four products themselves go out now (four launches, ~50 us of host time: they are the long pole of the auxiliary
                # stream); what consumes them (weight-composition backward, gradient assignment) is postponed
                products()
            else:
                # (A/B switch, off: submit the critical gather backward -> sort -> segment sums -> filter backward chain BEFORE the products;
                # in the PROFILED replay the gather backward starts 0.65 ms after its input is ready, profiles/r03_b_timeline.txt, but the
                # un-profiled step does not move: 3.33 against 3.35 ms.  DESIGN.md appendix A.)
                _defer(dev, products, front=True)
            grads = tuple(gb)
        else:
            grads = weight_gradients()
        return (d_xe, d_xp, None, d_own) + grads + (None,) * 6


class _Head(torch.autograd.Function):
    """Link predictor over the padded embeddings of a batch (``lstep_head_fwd`` / ``lstep_head_bwd``): fc1 -> relu -> fc2 for the
    positive and the negative pair of every edge, straight from the [src | dst | negative] row blocks (no ``torch.cat``), and a
    backward that returns the gradient of all three blocks in one [3 n, 176] tensor."""

    @staticmethod
    def forward(ctx, emb, fc1_w, fc1_b, fc2_w, fc2_b, n, layout, aux=None):
        ctx.set_materialize_grads(False)     # (no zero tensors for the outputs nobody differentiates: each would be a fill launch in backward)
        lib = nat.load_library()
        dev = emb.device
        half = fc1_w.shape[1] // 2                      # 172
        Hd = emb.shape[1]                               # 176
        # the padded operands of both directions in one launch (lstep_head_pack) instead of a fill, four slice copies and a transpose
        flat = torch.empty(2 * Hd * 2 * Hd + 2 * Hd, dtype=torch.float32, device=dev)
        wp, wt = flat[:Hd * 2 * Hd].view(Hd, 2 * Hd), flat[Hd * 2 * Hd:2 * Hd * 2 * Hd].view(2 * Hd, Hd)
        b1p, w2p = flat[2 * Hd * 2 * Hd:2 * Hd * 2 * Hd + Hd], flat[2 * Hd * 2 * Hd + Hd:]
        b2 = fc2_b.contiguous()
        with torch.cuda.device(dev):
            nat.check(lib.lstep_head_pack(nat.ptr(fc1_w.detach().contiguous()), nat.ptr(fc1_b.detach().contiguous()),
                                          nat.ptr(fc2_w.detach().contiguous()), fc1_w.shape[0], half, nat.ptr(wp), nat.ptr(wt), nat.ptr(b1p),
                                          nat.ptr(w2p), nat.current_stream()))
        h = torch.empty((2 * n, Hd), dtype=torch.float32, device=dev)
        logits = torch.empty(2 * n, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_head_fwd(nat.ptr(emb), n, *layout, nat.ptr(wp), nat.ptr(b1p), nat.ptr(w2p), nat.ptr(b2), nat.ptr(h),
                                         nat.ptr(logits), nat.current_stream()))
        ctx.save_for_backward(emb, wt, w2p, h)
        ctx.n, ctx.half, ctx.layout = n, half, tuple(layout)
        ctx.aux, ctx.params = aux, (fc1_w, fc1_b, fc2_w, fc2_b)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        if d_logits is None:
            return (None,) * 8
        lib = nat.load_library()
        emb, wt, w2p, h = ctx.saved_tensors
        n, half, Hd, dev = ctx.n, ctx.half, emb.shape[1], emb.device
        if ctx.layout != (0, n, 0, 2 * n):
            raise NotImplementedError("lstep_head_bwd implements the training layout (src | dst | negative dst)")
        d_logits = d_logits.contiguous()
        d_emb = torch.empty((3 * n, Hd), dtype=torch.float32, device=dev)
        d_h = torch.empty((2 * n, Hd), dtype=torch.float32, device=dev)
        d_hsum = torch.empty((n, Hd), dtype=torch.float32, device=dev)
        dw2_part = torch.empty(((n + 15) // 16, Hd), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_head_bwd(nat.ptr(d_logits), nat.ptr(h), n, nat.ptr(wt), nat.ptr(w2p), nat.ptr(d_emb), nat.ptr(d_h),
                                         nat.ptr(d_hsum), nat.ptr(dw2_part), nat.current_stream()))
        head_items = [(d_hsum, emb[:n], False, None), (d_h, emb[n:3 * n], True, None)]
        done = []        # filled by the dense tail's batched launch when it takes the two products along (engine mode)

        def parameter_gradients():
            (g_first, _), (g_second, g_b1) = done if done else nat.linear_wgrad_batch(head_items)
            g_fc1 = torch.cat([g_first[:half, :half], g_second[:half, :half]], dim=1)
            col = dw2_part.sum(dim=0)                        # [:172] d fc2.weight, [172] d fc2.bias (lstep_head_bwd)
            return g_fc1, g_b1[:half].contiguous(), col[:half].reshape(1, half), col[half:half + 1]

        aux = ctx.aux
        if aux is None:
            return (d_emb,) + parameter_gradients() + (None, None, None)
        # engine mode: the predictor's parameter gradients are produced on the auxiliary stream, later (``_flush_deferred``), and
        # assigned to .grad there
        ready = torch.cuda.Event()
        ready.record()
        params = ctx.params

        def on_aux():
            pend = _PENDING_HEAD_WGRADS.get(torch.device(dev))
            if pend is not None and pend[1] is done:      # (no dense tail took the products along: they are launched here)
                _PENDING_HEAD_WGRADS.pop(torch.device(dev))
            with torch.cuda.stream(aux):
                aux.wait_event(ready)
                for p, g in zip(params, parameter_gradients()):
                    p.grad = g if p.grad is None else p.grad + g
            for t in (d_h, d_hsum, emb, h, d_logits, dw2_part):
                t.record_stream(aux)

        if os.environ.get("LSTEP_WGRAD_NO_RIDE") != "1":
            # the dense tail's backward runs next on this stream and launches its four products on the auxiliary stream behind an event
            # recorded after ITS kernel, i.e. after this one too: the two products here join that launch (one graph node instead of three)
            _PENDING_HEAD_WGRADS[torch.device(dev)] = (head_items, done, aux)
        _defer(dev, on_aux)
        return d_emb, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ small modules
class TimeEncoder(nn.Module):
    """``cos(t * w + b)``; same parameters as reference ``models/modules.py:7-39``."""

    def __init__(self, time_dim: int, parameter_requires_grad: bool = True):
        super().__init__()
        self.time_dim = time_dim
        self.w = nn.Linear(1, time_dim)
        self.w.weight = nn.Parameter(torch.from_numpy(1 / 10 ** np.linspace(0, 9, time_dim, dtype=np.float32)).reshape(time_dim, -1))
        self.w.bias = nn.Parameter(torch.zeros(time_dim))
        if not parameter_requires_grad:
            self.w.weight.requires_grad = False
            self.w.bias.requires_grad = False

    def forward(self, timestamps: torch.Tensor, zero_mask: torch.Tensor = None):
        """timestamps float32 [...]; returns [..., time_dim] (HIP kernel ``lstep_time_encode``; forward only)."""
        if not timestamps.is_cuda:
            raise nat.LstepNativeError("TimeEncoder runs on the GPU only (no CPU fallback)")
        dt = timestamps.contiguous().float()
        out = torch.empty(dt.shape + (self.time_dim,), dtype=torch.float32, device=dt.device)
        zm = None if zero_mask is None else zero_mask.to(torch.uint8).contiguous()
        lib = nat.load_library()
        with torch.cuda.device(dt.device):
            nat.check(lib.lstep_time_encode(nat.ptr(dt), nat.ptr(zm), dt.numel(), nat.ptr(self.w.weight), nat.ptr(self.w.bias),
                                            self.time_dim, nat.ptr(out), nat.current_stream()))
        return out


class MergeLayer(nn.Module):
    """Link predictor ``fc2(relu(fc1(cat[a, b])))`` (reference ``models/modules.py:42-68``); dense, stays in torch."""

    def __init__(self, input_dim1: int, input_dim2: int, hidden_dim: int, output_dim: int):
        super().__init__()
        self.fc1 = nn.Linear(input_dim1 + input_dim2, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)
        self.act = nn.ReLU()

    def pair_logits(self, emb: torch.Tensor, n: int, layout):
        """fc2(relu(fc1(cat[a, b]))) for the positive and the negative pair of every edge, read straight from the row blocks of the
        padded embeddings ``emb`` [rows, 176] (``lstep_head_fwd``); ``layout`` = row offsets (pos_first, pos_second, neg_first,
        neg_second).  Returns the 2 n logits (positive pairs first)."""
        wait_aux_param_update(emb.device)
        aux = _aux_stream(emb.device) if (self.__dict__.get("aux_wgrad_stream", False) and torch.is_grad_enabled()) else None
        return _Head.apply(emb, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, int(n), tuple(int(x) for x in layout), aux)

    def fused_ok(self, emb: torch.Tensor) -> bool:
        return (emb.is_cuda and emb.shape[1] == 176 and tuple(self.fc1.weight.shape) == (172, 344) and tuple(self.fc2.weight.shape) == (1, 172)
                and os.environ.get("LSTEP_TORCH_HEAD") != "1")

    def forward(self, input_1: torch.Tensor, input_2: torch.Tensor):
        if input_1.is_cuda:
            wait_aux_param_update(input_1.device)
        return self.fc2(fast_linear(torch.cat([input_1, input_2], dim=1), self.fc1.weight, self.fc1.bias, relu=True))


# ------------------------------------------------------------------------------------------------ autograd glue
class SplicedRows:
    """Where the gradient of the current PE table lives during training.

    The table the loss sees is ``clone(last snapshot)`` with the FFT-filtered rows of the batch nodes written in
    (``train_LSTEP_link_prediction.py:229-230``): only those ``U`` rows carry gradient.  ``rows`` is that ``[U, P]``
    tensor (requires grad), ``slot_of`` the int32 ``[N+1]`` map node id -> row (or -1).  With it the backward of the
    gather stage accumulates into ``[U, P]`` instead of a dense ``[N+1, P]`` buffer.
    """

    def __init__(self, rows: torch.Tensor, slot_of: torch.Tensor, self_groups=None):
        self.rows = rows
        self.slot_of = slot_of
        # optional (ent_seg, ent_row) int32: the leading rows of the next gather call, already grouped by spliced row (see
        # ``_reduce_spliced_gradient``); only valid when those rows are exactly the entries the grouping was made from
        self.self_groups = self_groups


class _LiveCount:
    """Number of live entries seen by the last few ``_segment_reduce_rows`` calls of one call site, read back asynchronously."""

    def __init__(self):
        self.last, self.pending = None, None      # last known count; (pinned tensor, event, device tensor) on its way

    def poll(self):
        if self.pending is not None and self.pending[1].query():
            self.last = int(self.pending[0][0])
            self.pending = None
        return self.last

    def send(self, count_dev):
        if self.pending is None:
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(count_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending = (host, ev, count_dev)


def _segment_reduce_rows(mod, out: torch.Tensor, seg_of_entry: torch.Tensor, src_row_of, table: torch.Tensor, accumulate: bool, div: int = 0):
    """out[u] (+)= sum of table[src_row, :P] over the entries with seg_of_entry == u; entries with a negative segment are
    dropped.  ``seg_of_entry`` int32 [n]; ``src_row_of(order)`` maps original entry indices to table rows (``div`` > 0: it is
    ``order // div``).  ``out`` [U, P] must be zero where nothing has been accumulated yet.

    The live entries (typically ~5 %) are compacted and sorted by segment.  Their number is only known on the device; with ``div`` given
    the sort runs on a fixed capacity taken from earlier batches (twice the last count seen) and every consumer reads the count on the
    device, so the backward pass has no host round trip here; live entries beyond the capacity (a sudden jump) are added with atomics."""
    lib = nat.load_library()
    dev, P = table.device, mod.pe_dim
    n = seg_of_entry.numel()
    if n == 0:
        return
    keys = seg_of_entry.contiguous()
    key_bits = max(1, int(out.shape[0]).bit_length())
    track = mod.__dict__.setdefault("_live_counts", {}).setdefault((n, div), _LiveCount()) if div > 0 else None
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing:      # no read-back inside a captured iteration: the capacity comes from the eager iterations before it (overflow stays exact)
        if track is None or track.last is None:
            raise nat.LstepNativeError("a captured iteration needs one eager training iteration of the same shape first")
        last = track.last
    else:
        last = track.poll() if track is not None else None
    if last is not None and (capturing or os.environ.get("LSTEP_SYNC_LIVE_SORT") != "1"):
        capacity = min(n, max(8192, ((3 if capturing else 2) * last + 4095) // 4096 * 4096))
        sorted_keys, order, live_index, count = nat.sort_live_bounded(keys, key_bits, int(out.shape[0]), capacity)
        with torch.cuda.device(dev):
            ws, ws_bytes = nat.segment_workspace(dev, capacity, P)
            # (entry e reads table row order[e] // div: the division is the kernel's, not a framework launch)
            nat.check(lib.lstep_segment_rows_sum_live(nat.ptr(table), P, int(table.stride(0)), nat.ptr(sorted_keys), nat.ptr(order), max(div, 1),
                                                      capacity, nat.ptr(count), nat.ptr(out), P, 1 if accumulate else 0, nat.ptr(ws), ws_bytes,
                                                      nat.current_stream()))
            if capacity < n:
                nat.check(lib.lstep_scatter_add_overflow(nat.ptr(out), P, P, nat.ptr(keys), nat.ptr(live_index), nat.ptr(count), capacity,
                                                         max(div, 1), nat.ptr(table), int(table.stride(0)), nat.current_stream()))
        if not capturing:
            track.send(count)
        return
    sorted_keys, order, n_hit = nat.sort_live(keys, key_bits)
    if track is not None:
        track.last = n_hit
    if n_hit == 0:
        return
    # NOTE: every tensor whose address goes to the C ABI must stay referenced until the launch has been issued: a
    # temporary dies as soon as nat.ptr() returns and the caching allocator may hand its block to the next temporary.
    ent_seg = sorted_keys[:n_hit]
    ent_row = src_row_of(order[:n_hit])
    with torch.cuda.device(dev):
        ws, ws_bytes = nat.segment_workspace(dev, n_hit, P)
        nat.check(lib.lstep_segment_rows_sum(nat.ptr(table), P, int(table.stride(0)), None, None, 0, nat.ptr(ent_seg), nat.ptr(ent_row), None,
                                             n_hit, nat.ptr(out), P, 1 if accumulate else 0, None, nat.ptr(ws), ws_bytes, nat.current_stream()))


SPLICE_SMALL_ROWS, SPLICE_SMALL_HITS = 2048, 65536     # up to this many spliced rows / gather slots the gradient is reduced by one scanning launch


def _reduce_spliced_gradient(mod, num_rows: int, hits, g_pe, self_slot, g_self, self_groups=None):
    """Gradient of the spliced PE rows: every (row b, slot j) whose neighbour is spliced row u contributes g_pe[b, :P],
    every row b whose own node is spliced row u contributes g_self[b].  Grouped by u (``lstep_sort_live``) and reduced
    by ``lstep_segment_rows_sum`` into one buffer: no atomics on hot (hub) rows, deterministic summation order.
    ``self_groups = (ent_seg, ent_row)`` (int32): the caller already knows the grouping of the first ``len(ent_row)`` rows by spliced
    row (the engine's batch rows are cat[src, dst], grouped once per batch for the batch-node set): no second sort; the remaining
    rows (negative samples, rarely batch nodes) are added with float atomics (``lstep_scatter_add_rows``)."""
    lib = nat.load_library()
    K = hits.shape[1]
    P = mod.pe_dim
    slot_of, ids = self_slot if isinstance(self_slot, tuple) else (None, None)
    if (slot_of is not None and num_rows <= SPLICE_SMALL_ROWS and hits.numel() <= SPLICE_SMALL_HITS and (g_pe is not None or g_self is not None)
            and os.environ.get("LSTEP_NO_SMALL_SPLICE") != "1"):
        # the reference's own batch sizes: one launch that scans the hit list once per spliced row -- no compaction, no sort, no joins, no
        # atomics (lstep_spliced_grad_small); at B = 200 the nine launches it replaces were the critical chain of the captured step
        total = torch.empty((num_rows, P), dtype=torch.float32, device=hits.device)
        with torch.cuda.device(hits.device):
            nat.check(lib.lstep_spliced_grad_small(nat.ptr(hits), hits.numel() if g_pe is not None else 0, K, nat.ptr(g_pe),
                                                   int(g_pe.stride(0)) if g_pe is not None else P, nat.ptr(slot_of), nat.ptr(ids),
                                                   ids.numel() if g_self is not None else 0, nat.ptr(g_self),
                                                   int(g_self.stride(0)) if g_self is not None else P, P, nat.ptr(total), P, num_rows,
                                                   nat.current_stream()))
        return total
    if slot_of is not None:
        self_slot = slot_of[ids]
    total = torch.zeros((num_rows, P), dtype=torch.float32, device=hits.device)
    if g_pe is not None:
        _segment_reduce_rows(mod, total, hits.reshape(-1), lambda o: (o // K).contiguous(), g_pe, accumulate=False, div=K)
    if g_self is None:
        return total
    if self_groups is None:
        # (div = 1: entry e reads row e -- the form whose sort runs on a capacity taken from earlier batches, without a host round trip)
        _segment_reduce_rows(mod, total, self_slot.to(torch.int32), lambda o: o.contiguous(), g_self, accumulate=True, div=1)
        return total
    ent_seg, ent_row = self_groups
    n_known = ent_row.numel()
    with torch.cuda.device(hits.device):
        ws, ws_bytes = nat.segment_workspace(hits.device, n_known, P)
        nat.check(lib.lstep_segment_rows_sum(nat.ptr(g_self), P, int(g_self.stride(0)), None, None, 0, nat.ptr(ent_seg), nat.ptr(ent_row), None,
                                             n_known, nat.ptr(total), P, 1, None, nat.ptr(ws), ws_bytes, nat.current_stream()))
        rest = self_slot[n_known:].to(torch.int32).contiguous()
        if rest.numel():
            g_rest = g_self[n_known:]
            nat.check(lib.lstep_scatter_add_rows(nat.ptr(total), P, P, nat.ptr(rest), rest.numel(), nat.ptr(g_rest), int(g_self.stride(0)),
                                                 nat.current_stream()))
    return total


class _GatherAggregate(torch.autograd.Function):
    """lstep_gather_aggregate_fwd / _bwd.  Differentiable inputs: ``pe`` (dense table) OR ``rows`` (spliced rows), ``agg_w``.
    Outputs are row-padded to multiples of 16 floats (``mod.ld_*``; padding columns are zero) so every following GEMM has
    16-aligned K: hipBLASLt runs 176-wide fp32 GEMMs up to 2.5x faster than 172-wide ones (tools/gemm_shapes.py)."""

    @staticmethod
    def forward(ctx, pe, rows, agg_w, mod, ids, times, K, G, branches, slot_of, wide=False, self_groups=None, explicit=None, hub_groups=None):
        ctx.set_materialize_grads(False)     # (no zero tensors for the outputs nobody differentiates: each would be a fill launch in backward)
        lib = nat.load_library()
        dev = ids.device
        B = ids.numel()
        Fd, P, D = mod.feat_dim, mod.pe_dim, mod.time_dim
        en, pb = bool(branches & nat.BRANCH_EDGE_NODE), bool(branches & nat.BRANCH_PE)
        # wide: the node / self rows land in the first columns of the fused tail's concatenated operands
        # ([x_node | h1 | q] and [own | p1], lstep_tail_fwd), so no torch.cat ever copies them
        ld_node = mod.ld_node + mod.ld_edge + mod.ld_self if wide else mod.ld_node
        ld_self = 2 * mod.ld_self if wide else mod.ld_self
        out_edge = torch.empty((B, mod.ld_edge), dtype=torch.float32, device=dev) if en else None
        out_node = torch.empty((B, ld_node), dtype=torch.float32, device=dev) if en else None
        out_pe = torch.empty((B, mod.ld_pe), dtype=torch.float32, device=dev) if pb else None
        out_self = torch.empty((B, ld_self), dtype=torch.float32, device=dev) if pb else None
        count = torch.empty((B,), dtype=torch.int32, device=dev)
        pe_c = None
        if pb:
            pe_c = pe.detach()
            if pe_c.dtype != torch.float32 or not pe_c.is_contiguous():
                pe_c = pe_c.float().contiguous()
        aw = agg_w.detach().contiguous() if en else None
        tw, tb = mod.time_encoder.w.weight, mod.time_encoder.w.bias
        s = mod.neighbor_sampler
        sink = getattr(mod, "gather_event_sink", None)   # bench.py: HIP events on the launch stream right around the launch (roofline.achieved)
        if sink is not None and torch.cuda.is_current_stream_capturing():
            sink = None                                  # (timed events cannot be recorded into a graph)
        if sink is not None:
            if getattr(mod, "gather_event_idle", False):
                # bench.py's second timing: the launch with every other stream idle (the in-step launch runs beside the window slide,
                # history_advance_oldest, on the auxiliary stream: roofline.launch_ms vs roofline.launch_ms_idle)
                torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ws_flag = nat.WEIGHTED_SUM if (en and mod.weighted_sum) else 0
        # Hub nodes (round 5, csrc/hub.hip): on a graph with long adjacency rows, the batch rows of a node that occurs >= 16 times in this batch
        # -- known from the engine's grouping of cat[src, dst] by node (``self_groups``) -- get their node channel from prefix differences
        # over the union of their windows instead of time_gap row reads each; the gather kernel skips it for them.  LSTEP_NO_HUB_SUMS=1: off.
        # Measured (profiles/r05_hub_rows_ab.txt, c4-sized Zipf graphs): step 4.31 -> 2.91 ms (s = 1.2), 4.34 -> 2.87 (s = 1.5) with nodes of >= 4
        # occurrences served (>= 16: 3.10 / 2.93); the reference's small batches lose (Enron shape 0.485 -> 0.50-0.53 ms: a handful of
        # latency-bound work items and three more launches), so batches of fewer than 4096 grouped rows keep the gather kernel's own channel.
        hub = None
        groups = hub_groups if hub_groups is not None else self_groups
        if (explicit is None and en and groups is not None and not ws_flag and int(G) > 256 and getattr(s, "max_degree", 0) > 256
                and Fd <= 176 and 4096 <= groups[0].numel() <= B and groups[0].dtype == torch.int32 and os.environ.get("LSTEP_NO_HUB_SUMS") != "1"):
            seg32, order32 = groups
            n2 = seg32.numel()
            min_occ = max(2, int(os.environ.get("LSTEP_HUB_MIN_OCC", "4")))
            cap = int(lib.lstep_hub_capacity(n2, min_occ))
            served = torch.empty(B, dtype=torch.uint8, device=dev)
            seg_start = torch.empty(n2 + 1, dtype=torch.int32, device=dev)
            work = torch.empty((cap, 2), dtype=torch.int32, device=dev)
            nwork = torch.empty(1, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                nat.check(lib.lstep_hub_worklist(nat.ptr(seg32), nat.ptr(order32), n2, min_occ, nat.ptr(seg_start), nat.ptr(served), B, nat.ptr(work),
                                                 nat.ptr(nwork), cap, nat.current_stream()))
            hub = (served, order32, work, nwork, cap)
            mod.__dict__["_last_hub"] = (served, nwork)      # (bench.py: how many rows / work items the last launch served)
        with torch.cuda.device(dev):
            if explicit is None and hub is not None:
                nat.check(lib.lstep_gather_aggregate_fwd_skip(s.csr, nat.ptr(mod.node_raw_features), nat.ptr(mod.edge_raw_features), nat.ptr(pe_c),
                                                              Fd, P, nat.ptr(tw), nat.ptr(tb), D, nat.ptr(aw), nat.ptr(ids), nat.ptr(times), B,
                                                              int(K), int(G), int(branches) | ws_flag, nat.ptr(out_edge), nat.ptr(out_node),
                                                              nat.ptr(out_pe), nat.ptr(out_self), mod.ld_edge, ld_node, mod.ld_pe, ld_self,
                                                              nat.ptr(count), nat.ptr(hub[0]), nat.current_stream()))
                nat.check(lib.lstep_hub_node_sums(s.csr, nat.ptr(mod.node_raw_features), Fd, nat.ptr(ids), nat.ptr(times), int(G), nat.ptr(hub[1]),
                                                  nat.ptr(hub[2]), nat.ptr(hub[3]), hub[4], nat.ptr(out_node), ld_node, nat.current_stream()))
            elif explicit is None:
                nat.check(lib.lstep_gather_aggregate_fwd(s.csr, nat.ptr(mod.node_raw_features), nat.ptr(mod.edge_raw_features), nat.ptr(pe_c),
                                                         Fd, P, nat.ptr(tw), nat.ptr(tb), D, nat.ptr(aw), nat.ptr(ids), nat.ptr(times), B,
                                                         int(K), int(G), int(branches) | ws_flag, nat.ptr(out_edge), nat.ptr(out_node),
                                                         nat.ptr(out_pe), nat.ptr(out_self), mod.ld_edge, ld_node, mod.ld_pe, ld_self,
                                                         nat.ptr(count), nat.current_stream()))
            else:
                # explicit neighbourhoods (RNG-defined sampling): one launch per channel, each on its own draw
                l1, lg, l3 = explicit
                num_rows = int(mod.node_raw_features.shape[0])
                if en:
                    nat.check(lib.lstep_gather_explicit_fwd(nat.ptr(mod.node_raw_features), nat.ptr(mod.edge_raw_features), None, Fd, P, nat.ptr(tw),
                                                            nat.ptr(tb), D, nat.ptr(aw), nat.ptr(ids), nat.ptr(times), B, int(K), int(G),
                                                            nat.BRANCH_EDGE_NODE | ws_flag, nat.ptr(l1[0]), nat.ptr(l1[1]), nat.ptr(l1[2]),
                                                            nat.ptr(lg[0]), nat.ptr(lg[2]), num_rows, nat.ptr(out_edge), nat.ptr(out_node), None, None,
                                                            mod.ld_edge, ld_node, mod.ld_pe, ld_self, nat.current_stream()))
                if pb:
                    nat.check(lib.lstep_gather_explicit_fwd(None, None, nat.ptr(pe_c), Fd, P, nat.ptr(tw), nat.ptr(tb), D, None, nat.ptr(ids),
                                                            nat.ptr(times), B, int(K), int(G), nat.BRANCH_PE, nat.ptr(l3[0]), None, nat.ptr(l3[2]),
                                                            None, None, num_rows, None, None, nat.ptr(out_pe), nat.ptr(out_self), mod.ld_edge,
                                                            ld_node, mod.ld_pe, ld_self, nat.current_stream()))
                count.fill_(int(K))
        if sink is not None:
            e1.record()
            sink.append((e0, e1, count, int(branches)))      # (branches: which channels this launch computed -- bench.py pairs split launches by it)
        ctx.mod, ctx.sampler, ctx.K, ctx.branches, ctx.ld_self, ctx.self_groups = mod, s, int(K), int(branches), ld_self, self_groups
        ctx.explicit = explicit
        ctx.pe_shape = tuple(pe.shape) if pe is not None else None
        ctx.rows_shape = tuple(rows.shape) if rows is not None else None
        ctx.save_for_backward(ids, times, count, slot_of if slot_of is not None else torch.empty(0, device=dev))
        ctx.has_slot = slot_of is not None
        outs = tuple(o if o is not None else torch.empty(0, device=dev) for o in (out_edge, out_node, out_pe, out_self))
        ctx.mark_non_differentiable(outs[1], count)
        return outs + (count,)

    @staticmethod
    def backward(ctx, g_edge, g_node, g_pe, g_self, g_count):
        lib = nat.load_library()
        ids, times, count, slot_of = ctx.saved_tensors
        mod, K = ctx.mod, ctx.K
        dev = ids.device
        B = ids.numel()
        Fd, P, D = mod.feat_dim, mod.pe_dim, mod.time_dim
        en, pb = bool(ctx.branches & nat.BRANCH_EDGE_NODE), bool(ctx.branches & nat.BRANCH_PE)
        need_pe, need_rows, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        g_edge = g_edge.contiguous() if (en and need_w and g_edge is not None) else None
        want_pe = pb and (need_pe or need_rows)
        g_pe = g_pe.contiguous() if (want_pe and g_pe is not None) else None
        g_self = g_self.contiguous() if (want_pe and g_self is not None) else None
        slot_dot = torch.empty((B, K), dtype=torch.float32, device=dev) if g_edge is not None else None
        grad_rows = None
        hits = None
        use_slot = False
        atomic_rows = False
        if want_pe:
            if need_rows:
                use_slot = True
                # LSTEP_HITS_ATOMIC=1 (A/B, measured and left off: DESIGN.md appendix A): the spliced rows' gradient accumulated by the
                # backward kernel itself with float atomics instead of hit list -> compaction -> sort -> segment sums
                atomic_rows = os.environ.get("LSTEP_HITS_ATOMIC") == "1" and ctx.explicit is None
                if atomic_rows:
                    grad_rows = torch.zeros(ctx.rows_shape, dtype=torch.float32, device=dev)
                else:
                    hits = torch.empty((B, K), dtype=torch.int32, device=dev)   # sort-based reduction below, no atomics
            else:
                grad_rows = torch.zeros(ctx.pe_shape, dtype=torch.float32, device=dev)
        tw, tb = mod.time_encoder.w.weight, mod.time_encoder.w.bias
        g_w = None
        if ctx.explicit is not None and (g_edge is not None or grad_rows is not None or hits is not None):
            l1, lg, l3 = ctx.explicit
            num_rows = int(mod.node_raw_features.shape[0])
            with torch.cuda.device(dev):
                if g_edge is not None:       # edge channel on ITS draw
                    nat.check(lib.lstep_gather_explicit_bwd(nat.ptr(mod.edge_raw_features), Fd, P, nat.ptr(tw), nat.ptr(tb), D, nat.ptr(ids),
                                                            nat.ptr(times), B, K, nat.ptr(l1[0]), nat.ptr(l1[1]), nat.ptr(l1[2]), num_rows,
                                                            nat.ptr(g_edge), None, None, mod.ld_edge, mod.ld_pe, ctx.ld_self, None,
                                                            nat.ptr(slot_dot), None, None, nat.current_stream()))
                if grad_rows is not None or hits is not None:      # PE channel on its own draw
                    nat.check(lib.lstep_gather_explicit_bwd(None, Fd, P, nat.ptr(tw), nat.ptr(tb), D, nat.ptr(ids), nat.ptr(times), B, K,
                                                            nat.ptr(l3[0]), None, nat.ptr(l3[2]), num_rows, None, nat.ptr(g_pe), nat.ptr(g_self),
                                                            mod.ld_edge, mod.ld_pe, ctx.ld_self, nat.ptr(slot_of) if use_slot else None, None,
                                                            nat.ptr(grad_rows), nat.ptr(hits), nat.current_stream()))
        elif g_edge is not None or grad_rows is not None or hits is not None:
            # Two consumers with very different urgency share this kernel: the spliced-row HITS (a few index reads per row) head the
            # chain sort -> segment sums -> history-filter backward, while the slot dots re-read 688 * k bytes of edge rows per row
            # (0.3 ms at the bench shape) for ONE parameter gradient, d(edge_agg.weight), that nothing waits for before the optimiser.
            # LSTEP_SPLIT_GATHER_BWD=1 launches them apart (hits here, the slot dots on a side stream).  Off by default: measured 3.38
            # against 3.37 ms per step at c4 -- the backward phase is bound by the matrix-core kernels the three streams share, not by
            # this chain.
            aux_split = (g_edge is not None and hits is not None and mod.__dict__.get("aux_wgrad_stream", False)
                         and mod.edge_agg.weight.grad is None and os.environ.get("LSTEP_SPLIT_GATHER_BWD") == "1")
            with torch.cuda.device(dev):
                nat.check(lib.lstep_gather_aggregate_bwd(ctx.sampler.csr, nat.ptr(mod.edge_raw_features), Fd, P, nat.ptr(tw), nat.ptr(tb), D,
                                                         nat.ptr(ids), nat.ptr(times), nat.ptr(count), B, K, None if aux_split else nat.ptr(g_edge),
                                                         nat.ptr(g_pe), nat.ptr(g_self), mod.ld_edge, mod.ld_pe, ctx.ld_self,
                                                         nat.ptr(slot_of) if use_slot else None, None if aux_split else nat.ptr(slot_dot),
                                                         nat.ptr(grad_rows), nat.ptr(hits), nat.current_stream()))
            if aux_split:
                aux = _side_stream(dev)
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(aux), torch.cuda.device(dev):
                    aux.wait_event(ready)
                    nat.check(lib.lstep_gather_aggregate_bwd(ctx.sampler.csr, nat.ptr(mod.edge_raw_features), Fd, P, nat.ptr(tw), nat.ptr(tb), D,
                                                             nat.ptr(ids), nat.ptr(times), nat.ptr(count), B, K, nat.ptr(g_edge), None, None,
                                                             mod.ld_edge, mod.ld_pe, ctx.ld_self, None, nat.ptr(slot_dot), None, None,
                                                             nat.current_stream()))
                    g_w = slot_dot.sum(dim=0)
                for t_ in (g_edge, slot_dot, ids, times, count):
                    t_.record_stream(aux)
                g_w.record_stream(torch.cuda.current_stream(dev))
        g_w_done = g_w is not None
        if os.environ.get("LSTEP_WGRAD_LATE") != "1":
            _flush_deferred(dev)     # the critical kernel is out: now launch the postponed auxiliary-stream work
        if use_slot and not atomic_rows:
            grad_rows = _reduce_spliced_gradient(mod, ctx.rows_shape[0], hits, g_pe, (slot_of, ids), g_self, ctx.self_groups)
        if g_w is None and slot_dot is not None:
            g_w = slot_dot.sum(dim=0)
        g_table = None
        if grad_rows is not None and not use_slot:
            if g_pe is not None and ctx.explicit is None:  # padding slots all read row 0: one weighted column sum instead of a hot atomic row
                npad = (K - count.clamp(max=K)).to(torch.float32)      # (explicit lists carry their padding slots as id 0: already added)
                grad_rows[0] += npad @ g_pe[:, :P]
            g_table = grad_rows
        elif grad_rows is not None and g_pe is not None:
            # spliced mode: row 0 only has gradient if node 0 is itself a spliced row (never in the reference data)
            pass
        return (g_table, grad_rows if use_slot else None, g_w, None, None, None, None, None, None, None, None, None, None, None)


class _FftCoefficients(torch.autograd.Function):
    """Real [T, P] coefficient table of the FFT filter (see ``LSTEP.fft_coefficients``) with a hand-derived backward.

        A[f]    = sum_t E+[f, t] a[t] m[t]          E+[f, t] = e^{+2 pi i f t / T}
        c[f]    = m[f] A[f] / T
        coef    = Re(E- Q),  Q[f, p] = c[f] W[f, p]  E-[s, f] = e^{-2 pi i f s / T}

    For a real loss with G = d(loss)/d(coef):  gQ = E+ G,  gW = conj(c) gQ,  gc[f] = sum_p conj(W[f, p]) gQ[f, p],
    gA = (m / T) gc,  g(a m) = Re(E- gA)  (a is real),  ga = m * g(a m).  (PyTorch's convention: the gradient of a complex
    tensor z = x + iy is dL/dx + i dL/dy.)  ~10 launches instead of ~25 through complex autograd."""

    @staticmethod
    def forward(ctx, w, a_w, m, e_pos, e_neg_t, T):
        ctx.set_materialize_grads(False)     # (no zero tensors for the outputs nobody differentiates: each would be a fill launch in backward)
        ctx.native = bool(w.is_cuda and T <= 256 and os.environ.get("LSTEP_TORCH_FFTCOEF") != "1")
        if ctx.native:      # one kernel (lstep_fft_coef_fwd) instead of ~12 complex128 framework launches
            lib = nat.load_library()
            wr = torch.view_as_real(w.detach().contiguous())
            a32 = a_w.detach().reshape(-1).to(torch.float32).contiguous()
            coef = torch.empty((T, w.shape[1]), dtype=torch.float32, device=w.device)
            c = torch.empty((T, 2), dtype=torch.float64, device=w.device)
            with torch.cuda.device(w.device):
                nat.check(lib.lstep_fft_coef_fwd(nat.ptr(wr), nat.ptr(a32), nat.ptr(m), T, w.shape[1], nat.ptr(coef), nat.ptr(c), nat.current_stream()))
            ctx.save_for_backward(wr, c, m)
            ctx.T, ctx.a_shape = T, tuple(a_w.shape)
            return coef
        am = (a_w.reshape(-1).to(torch.float64) * m).to(torch.complex128)
        big_a = e_pos @ am
        c = big_a * (m / T)                                       # [f] complex128
        w128 = w.to(torch.complex128)
        coef = (e_neg_t @ (w128 * c.unsqueeze(1))).real.to(torch.float32)
        ctx.save_for_backward(w128, c, m, e_pos, e_neg_t)
        ctx.T = T
        ctx.a_shape = tuple(a_w.shape)
        return coef

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 6
        if ctx.native:
            lib = nat.load_library()
            wr, c, m = ctx.saved_tensors
            T, P = wr.shape[0], wr.shape[1]
            g = g.contiguous()
            g_w = torch.empty((T, P, 2), dtype=torch.float32, device=g.device)
            g_a = torch.empty(T, dtype=torch.float32, device=g.device)
            scratch = torch.empty((T, 2), dtype=torch.float64, device=g.device)
            with torch.cuda.device(g.device):
                nat.check(lib.lstep_fft_coef_bwd(nat.ptr(g), nat.ptr(wr), nat.ptr(c), nat.ptr(m), T, P, nat.ptr(g_w), nat.ptr(g_a), nat.ptr(scratch),
                                                 nat.current_stream()))
            return torch.view_as_complex(g_w), g_a.reshape(ctx.a_shape), None, None, None, None
        w128, c, m, e_pos, e_neg_t = ctx.saved_tensors
        gq = e_pos @ g.to(torch.complex128)                       # E-^H = E+
        g_w = (gq * c.conj().unsqueeze(1)).to(torch.complex64)
        g_c = (w128.conj() * gq).sum(dim=1)
        g_am = (e_neg_t @ (g_c * (m / ctx.T))).real               # E+^H = E-
        g_a = (g_am * m).to(torch.float32).reshape(ctx.a_shape)
        return g_w, g_a, None, None, None, None


class _HistoryFilter(torch.autograd.Function):
    """out[u] = sum_s coef[s] * hist[ids[u], s]  (lstep_history_filter_fwd / _bwd); gradient only w.r.t. ``coef``.
    With a change ``mask`` (int32 [rows, words], maintained by the device ring: which snapshots of a node differ from the one before)
    the ``*_runs_*`` kernels read one row per run of equal snapshots instead of one per snapshot."""

    @staticmethod
    def forward(ctx, coef, hist_base, geom, ids, mask=None, oldest=None, splice=None, live=None, ring=None):
        ctx.set_materialize_grads(False)     # (no zero tensors for the outputs nobody differentiates: each would be a fill launch in backward)
        lib = nat.load_library()
        node_stride, time_stride, slots, rot, t_len, P = geom
        U = ids.numel()
        if live is not None and mask is None:
            raise ValueError("a device-resident row count needs the change-mask kernels")
        # (with ``live``, ids is a capacity-sized list: rows past the live count are never written and never read)
        out = torch.empty((U, P), dtype=torch.float32, device=ids.device)
        cc = coef.detach().contiguous()
        with torch.cuda.device(ids.device):
            if mask is None:
                nat.check(lib.lstep_history_filter_fwd(nat.ptr(hist_base), node_stride, time_stride, slots, rot, t_len, P, nat.ptr(ids), U,
                                                       nat.ptr(cc), nat.ptr(out), nat.current_stream()))
            else:
                ws = nat._workspace(ids.device, int(lib.lstep_history_filter_runs_workspace(t_len, P)))
                nat.check(lib.lstep_history_filter_runs_fwd(nat.ptr(hist_base), node_stride, time_stride, slots, rot, t_len, P, nat.ptr(mask),
                                                            int(mask.shape[1]), nat.ptr(oldest), nat.ptr(ids), U, nat.ptr(cc), nat.ptr(ws),
                                                            nat.ptr(out), nat.ptr(splice[0]) if splice else None,
                                                            nat.ptr(splice[1]) if splice else None, nat.ptr(live), ring, nat.current_stream()))
        ctx.geom, ctx.coef_shape = geom, tuple(coef.shape)
        # NOT save_for_backward: the device ring appends its next snapshot (a slot outside this window) in place
        # before backward runs; the window itself (rows and mask bits) is guaranteed untouched by HistoryRing (engine.py).
        ctx.hist, ctx.mask, ctx.oldest, ctx.ring = hist_base, mask, oldest, ring    # (ring: lstep_ring_ref_t*, the window's rotation on the device)
        ctx.save_for_backward(ids)
        return out

    @staticmethod
    def backward(ctx, g_out):
        if g_out is None:
            return (None,) * 9
        lib = nat.load_library()
        (ids,) = ctx.saved_tensors
        hist_base, mask = ctx.hist, ctx.mask
        node_stride, time_stride, slots, rot, t_len, P = ctx.geom
        U = ids.numel()
        full = ctx.needs_input_grad[0] and U > 0 and t_len == ctx.coef_shape[0] and mask is not None      # every row is written below
        g_coef = (torch.empty if full else torch.zeros)(ctx.coef_shape, dtype=torch.float32, device=ids.device)
        if ctx.needs_input_grad[0] and t_len > 0 and U > 0:
            chunks = int(lib.lstep_history_filter_bwd_chunks(U) if mask is None else lib.lstep_history_filter_runs_bwd_chunks(U, t_len))
            partial = torch.empty((chunks, t_len, P), dtype=torch.float32, device=ids.device)
            g = g_out.contiguous()
            with torch.cuda.device(ids.device):
                if mask is None:
                    nat.check(lib.lstep_history_filter_bwd(nat.ptr(hist_base), node_stride, time_stride, slots, rot, t_len, P, nat.ptr(ids), U,
                                                           nat.ptr(g), nat.ptr(partial), nat.current_stream()))
                    g_coef[:t_len] = partial.sum(dim=0)
                else:
                    nat.check(lib.lstep_history_filter_runs_bwd(nat.ptr(hist_base), node_stride, time_stride, slots, rot, t_len, P,
                                                                nat.ptr(mask), int(mask.shape[1]), nat.ptr(ctx.oldest), nat.ptr(ids), U,
                                                                nat.ptr(g), nat.ptr(partial), ctx.ring, nat.current_stream()))
                    diff = partial.sum(dim=0)
                    nat.check(lib.lstep_history_filter_runs_finish(nat.ptr(diff), t_len, P, nat.ptr(g_coef), nat.current_stream()))
        _flush_deferred(ids.device)      # the critical chain is out: now the postponed parameter-gradient work (see ``_FusedTail.backward``)
        return g_coef, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ backbone
class LSTEP(nn.Module):
    def __init__(self, node_raw_features: np.ndarray, edge_raw_features: np.ndarray, neighbor_sampler, full_neighbor_sampler=None,
                 pe_dim: int = 172, num_neighbors: int = 20, time_feat_dim: int = 100, num_fft_batches: int = 100,
                 use_dropout=False, dropout: float = 0.1, weighted_sum=False, concat_pe=True, device: str = "cuda"):
        super().__init__()
        nat.load_library()  # fail loudly: no HIP library, no model
        # (use_dropout=True -- never set by the reference drivers, SURVEY.md appendix A.8 -- is served by plain framework tails behind the
        # same gather kernels: a dropout between edge_mlp_2 and node_mlp breaks the pre-multiplied tail; models/LSTEP.py:171-172)
        edge_feat_dim = edge_raw_features.shape[-1]
        node_feat_dim = node_raw_features.shape[-1]
        if edge_feat_dim != node_feat_dim:
            raise ValueError("node and edge feature widths must match (the reference pads both to 172)")
        self.num_fft_batches = num_fft_batches
        self.num_nodes = node_raw_features.shape[0]
        self.pe_dim, self.feat_dim, self.time_dim = pe_dim, node_feat_dim, time_feat_dim
        self.num_neighbors = num_neighbors
        # padded row strides of the gather-stage outputs / hidden activations (see _GatherAggregate)
        self.ld_edge, self.ld_node = _round16(time_feat_dim + node_feat_dim), _round16(node_feat_dim)
        self.ld_pe, self.ld_self = _round16(pe_dim + time_feat_dim), _round16(pe_dim)
        self.use_dropout, self.dropout, self.concat_pe, self.weighted_sum = use_dropout, dropout, concat_pe, weighted_sum
        self.device = torch.device(device)

        def table(x):
            if isinstance(x, torch.Tensor):
                return x.to(device=self.device, dtype=torch.float32).contiguous()
            return torch.from_numpy(np.asarray(x).astype(np.float32, copy=False)).to(self.device).contiguous()

        # plain attributes, not buffers: the reference keeps them out of state_dict too (models/LSTEP.py:45-46)
        self.node_raw_features = table(node_raw_features)
        self.edge_raw_features = table(edge_raw_features)
        # (a checked build -- LSTEP_LIB=.../liblstep_hip_checked.so -- learns the table heights its guards compare ids with; no-op otherwise)
        nat.set_debug_limits(self.node_raw_features.shape[0], self.edge_raw_features.shape[0])
        self.neighbor_sampler = neighbor_sampler
        self.full_neighbor_sampler = full_neighbor_sampler
        self.time_encoder = TimeEncoder(time_feat_dim, parameter_requires_grad=False)

        c = edge_feat_dim + time_feat_dim
        with warnings.catch_warnings():  # "Complex modules are a new feature": the reference builds the same complex Linear
            warnings.simplefilter("ignore", UserWarning)
            self.fft_filter = nn.Linear(pe_dim, num_fft_batches, bias=False).to(torch.complex64)
        self.fft_dropout = nn.Dropout(p=dropout)
        self.fft_agg = nn.Linear(num_fft_batches, 1, bias=False)
        self.edge_mlp_1 = nn.Linear(c, c)
        self.edge_agg = nn.Linear(num_neighbors, 1)
        self.edge_mlp_2 = nn.Linear(c, c)
        self.node_mlp = nn.Linear(c + node_feat_dim, node_feat_dim)
        self.self_update_pe = nn.Linear(pe_dim, pe_dim)
        self.pe_mlp_1 = nn.Linear(pe_dim + time_feat_dim, pe_dim)
        self.pe_mlp_2 = nn.Linear(pe_dim, pe_dim)
        self.self_update_neighbor_pe = nn.Linear(pe_dim, pe_dim)
        self.pe_neighbor_mlp_1 = nn.Linear(pe_dim + time_feat_dim, pe_dim)
        self.pe_neighbor_mlp_2 = nn.Linear(pe_dim, pe_dim)
        self.out_node_emb = nn.Linear(pe_dim + node_feat_dim, node_feat_dim)
        self.to(self.device)

    def close(self):
        """Release the captured weight-composition graphs now (they are re-captured on the next use).  Optional: a model that is simply
        dropped gives them back at the package's next safe point (``drain_dead_graphs``)."""
        tw = self.__dict__.pop("_tail_weight_graph", None)
        if tw is not None:
            tw.close()

    # ---- sampler handling (models/LSTEP.py:76-85)
    def set_neighbor_sampler(self, neighbor_sampler):
        self.neighbor_sampler = neighbor_sampler
        if self.neighbor_sampler.sample_neighbor_strategy in ["uniform", "time_interval_aware"]:
            assert self.neighbor_sampler.seed is not None
            self.neighbor_sampler.reset_random_state()

    # ---- helpers
    def _ids(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.int64).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(self.device)

    def _times(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float64).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def _check_rows(self, ids_np):
        if not isinstance(ids_np, torch.Tensor) and len(ids_np):
            lo, hi = int(np.min(ids_np)), int(np.max(ids_np))
            if lo < 0 or hi >= self.neighbor_sampler.num_rows or hi >= self.node_raw_features.shape[0]:
                raise IndexError(f"node id out of range [0, {self.node_raw_features.shape[0]})")

    def _draw_neighbourhoods(self, node_ids, node_interact_times, K, G, branches, row_blocks: int = 1):
        """RNG-defined sampling strategies ('uniform', 'time_interval_aware', utils/utils.py:175-198): the draws are defined by the call
        order of numpy's RandomState, so they are made by the host sampler in EXACTLY the reference's order -- per
        combining_pe_raw_feat call K slots (edge channel, models/LSTEP.py:147), time_gap slots (node channel, :177), K slots again (PE
        channel, :223) -- and handed to the explicit-neighbourhood kernels.  ``row_blocks``: the rows are that many equal blocks which
        the reference would have passed in separate calls (the engine merges src | dst | negatives into one launch)."""
        ids = node_ids.cpu().numpy() if isinstance(node_ids, torch.Tensor) else np.asarray(node_ids)
        ts = node_interact_times.cpu().numpy() if isinstance(node_interact_times, torch.Tensor) else np.asarray(node_interact_times)
        n = len(ids)
        if row_blocks < 1 or n % row_blocks:
            raise ValueError("row_blocks must divide the number of rows")
        step = n // row_blocks
        en, pb = bool(branches & nat.BRANCH_EDGE_NODE), bool(branches & nat.BRANCH_PE)
        sampler = self.neighbor_sampler
        if hasattr(sampler, "sample_random_into") and len(ids) == len(ts) and os.environ.get("LSTEP_RNG_PAGEABLE") != "1":
            # the draws land in pinned staging arrays, one [n, width] triple per channel, block by block in the reference's call order; one
            # asynchronous copy each (round 5: against np.zeros + concatenate + a pageable copy of ~0.6 GB per iteration at time_gap = 2000)
            chans = [(0, K, en), (1, G, en), (2, K, pb)]
            stage = [self._rng_stage(c, n, w) if on else None for c, w, on in chans]
            dev = [None if st is None else tuple(torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in st[0]) for st in stage]
            for b in range(row_blocks):
                sl = slice(b * step, (b + 1) * step)
                for (c, w, on), st, dv in zip(chans, stage, dev):
                    if on:
                        sampler.sample_random_into(ids[sl], ts[sl], w, tuple(a[sl] for a in st[1]))
                        for d, t in zip(dv, st[0]):       # (this block's copy runs under the next block's draws)
                            d[sl].copy_(t[sl], non_blocking=True)
            for st in stage:
                if st is not None:
                    st[2].record(torch.cuda.current_stream(self.device))
            return tuple(dev)
        draws = ([], [], [])
        for b in range(row_blocks):
            sl = slice(b * step, (b + 1) * step)
            if en:
                draws[0].append(sampler.get_historical_neighbors(ids[sl], ts[sl], K))
                draws[1].append(sampler.get_historical_neighbors(ids[sl], ts[sl], G))
            if pb:
                draws[2].append(sampler.get_historical_neighbors(ids[sl], ts[sl], K))

        def to_dev(parts):
            if not parts:
                return None
            cat = [np.concatenate([p[i] for p in parts], axis=0) for i in range(3)]
            return (torch.from_numpy(np.ascontiguousarray(cat[0], dtype=np.int64)).to(self.device),
                    torch.from_numpy(np.ascontiguousarray(cat[1], dtype=np.int64)).to(self.device),
                    torch.from_numpy(np.ascontiguousarray(cat[2], dtype=np.float32)).to(self.device))
        return tuple(to_dev(d) for d in draws)

    def _rng_stage(self, channel: int, rows: int, width: int):
        """Pinned host staging of one channel's draws: (three torch tensors, their numpy views, the event behind their last copy).  Two sets
        per (channel, shape), used alternately, so that filling the next call's draws does not wait for the previous call's copy."""
        pool = self.__dict__.setdefault("_rng_stages", {})
        key = (channel, rows, width)
        ring = pool.get(key)
        if ring is None:
            if len(pool) >= 12:                         # (shapes come and go with batch sizes: keep the staging bounded)
                pool.clear()
            ring = pool[key] = {"next": 0, "sets": []}
        if len(ring["sets"]) < 2:
            ts_ = (torch.empty((rows, width), dtype=torch.int64, pin_memory=True), torch.empty((rows, width), dtype=torch.int64, pin_memory=True),
                   torch.empty((rows, width), dtype=torch.float32, pin_memory=True))
            ring["sets"].append((ts_, tuple(t.numpy() for t in ts_), torch.cuda.Event()))
            return ring["sets"][-1]
        st = ring["sets"][ring["next"]]
        ring["next"] ^= 1
        st[2].synchronize()                             # (the copy that last read this set has finished)
        return st

    def _gather(self, pe, node_ids, node_interact_times, K, G, branches, spliced: SplicedRows = None, wide: bool = False, row_blocks: int = 1,
                row_groups=None):
        if K != self.num_neighbors and (branches & nat.BRANCH_EDGE_NODE):
            raise RuntimeError(f"edge_agg was built for num_neighbors={self.num_neighbors}, got {K} "
                               "(the reference fails the same way at models/LSTEP.py:164)")
        explicit = None
        if getattr(self.neighbor_sampler, "sample_neighbor_strategy", "recent") != "recent":
            if len(node_ids) != len(node_interact_times):
                raise ValueError("node_ids and node_interact_times must have the same length")
            explicit = self._draw_neighbourhoods(node_ids, node_interact_times, K, G, branches, row_blocks)
        self._check_rows(node_ids)
        ids, times = self._ids(node_ids), self._times(node_interact_times)
        if ids.numel() != times.numel():
            raise ValueError("node_ids and node_interact_times must have the same length")
        if branches & nat.BRANCH_PE:
            # the kernels index pe[neighbour id]: a short table would be an out-of-bounds read on the GPU
            if pe is None or pe.dim() != 2 or pe.shape[1] != self.pe_dim or pe.shape[0] < self.neighbor_sampler.num_rows:
                raise ValueError(f"pe must be [>= {self.neighbor_sampler.num_rows}, {self.pe_dim}] (one row per node id, row 0 = padding)")
            if not pe.is_cuda:
                raise nat.LstepNativeError("pe must live on the GPU (no CPU fallback)")
        rows = spliced.rows if spliced is not None else None
        slot_of = spliced.slot_of if spliced is not None else None
        if slot_of is not None and (slot_of.dtype != torch.int32 or slot_of.numel() < self.neighbor_sampler.num_rows):
            raise ValueError("slot_of must be an int32 map with one entry per node id")
        # (row_groups = (seg, order) int32: the leading rows grouped by node when no spliced-row grouping says so -- evaluation iterations: hub nodes)
        return _GatherAggregate.apply(pe, rows, self.edge_agg.weight.reshape(-1), self, ids, times, K, G, branches, slot_of, wide,
                                      getattr(spliced, "self_groups", None), explicit, row_groups)

    def _edge_node_tail(self, x_edge, x_node):
        """edge_mlp_1 -> edge_agg (reassociated) -> relu -> edge_mlp_2 ; node_mlp(cat[node, edge])  (models/LSTEP.py:161-170,219)."""
        a = self.edge_agg.weight.reshape(-1)
        h = F.linear(x_edge, self.edge_mlp_1.weight) + (a.sum() * self.edge_mlp_1.bias + self.edge_agg.bias)
        h = self.edge_mlp_2(torch.relu(h))
        if self.use_dropout:
            h = F.dropout(h, p=self.dropout)      # (the functional form with its default training=True, as models/LSTEP.py:171-172 calls it)
        return self.node_mlp(torch.cat([x_node, h], dim=-1))

    def _pe_tail(self, x_pe, own):
        """pe_neighbor_mlp_1/2, self_update_neighbor_pe, tanh, residual (models/LSTEP.py:240-247)."""
        wait_aux_param_update(x_pe.device)
        a = self.pe_neighbor_mlp_2(torch.relu(self.pe_neighbor_mlp_1(x_pe)))
        return own + torch.tanh(self.self_update_neighbor_pe(own) + a)

    # ---- A + N (models/LSTEP.py:139-220)
    def aggregated_node_embeddings(self, node_ids, node_interact_times, num_neighbors: int = 20, time_gap: int = 2000, testing=False):
        x_edge, x_node, _, _, _ = self._gather(None, node_ids, node_interact_times, num_neighbors, time_gap, nat.BRANCH_EDGE_NODE)
        return self._edge_node_tail(x_edge[:, :self.time_dim + self.feat_dim], x_node[:, :self.feat_dim])

    # ---- C (models/LSTEP.py:222-249)
    def compute_neighborhood_pe(self, pe, node_ids, node_interact_times, num_neighbors: int = 30, spliced: SplicedRows = None):
        _, _, x_pe, own, _ = self._gather(pe, node_ids, node_interact_times, num_neighbors, 1, nat.BRANCH_PE, spliced)
        return self._pe_tail(x_pe[:, :self.pe_dim + self.time_dim], own[:, :self.pe_dim])

    # ---- O (models/LSTEP.py:251-266): one fused gather launch serves A, N and C
    def combining_pe_raw_feat(self, pe, node_ids, node_interact_times, num_neighbors: int = 30, time_gap: int = 2000, testing=False,
                              spliced: SplicedRows = None, padded: bool = False, row_blocks: int = 1, row_groups=None):
        """``row_blocks`` (RNG-defined sampling only): see ``_draw_neighbourhoods``; ``row_groups``: see ``_gather``."""
        fused = self._fused_tail_ok()
        x_edge, x_node, x_pe, own, _ = self._gather(pe, node_ids, node_interact_times, num_neighbors, time_gap,
                                                    nat.BRANCH_EDGE_NODE | nat.BRANCH_PE, spliced, wide=fused, row_blocks=row_blocks,
                                                    row_groups=row_groups)
        if self.use_dropout:
            # the dropout of models/LSTEP.py:171-172 sits between edge_mlp_2 and node_mlp, inside the stretch the fused tail pre-multiplies:
            # the layers one by one (framework GEMMs) behind the same fused gather launch
            emb = self._edge_node_tail(x_edge[:, :self.time_dim + self.feat_dim], x_node[:, :self.feat_dim])
            q = self._pe_tail(x_pe[:, :self.pe_dim + self.time_dim], own[:, :self.pe_dim])
            out = self.out_node_emb(torch.cat([emb, q], dim=-1))
            return F.pad(out, (0, self.ld_node - self.feat_dim)) if padded else out
        out = self._combined_tail(x_edge, x_node, x_pe, own, fused)
        # padded: the [B, 176] rows the kernels work on (columns >= 172 are 0), for lstep_head_fwd; default: the reference's [B, 172]
        return out if padded else out[:, :self.feat_dim]

    def prepare_step(self):
        """Issue the parameter-only work of the next ``combining_pe_raw_feat`` (the replay of the dense tail's weight composition, ~25
        small kernels) on the auxiliary stream now, so it runs beside the history filter and the gather stage instead of between the
        gather stage and the dense tail.  Optional; call once per iteration, after the previous optimiser step."""
        tw = self.__dict__.get("_tail_weight_graph")
        if tw is not None and self.__dict__.get("aux_wgrad_stream", False) and torch.is_grad_enabled():
            tw.prepare(_aux_stream(tw.params[0].device))

    def join_aux_stream(self):
        """Make the current stream wait for the weight-gradient work that ``aux_wgrad_stream = True`` put on the auxiliary stream
        (call after ``backward()`` and before anything reads the parameter gradients)."""
        _flush_deferred()
        for dev, st in _SIDE_STREAMS.items():
            torch.cuda.current_stream(dev).wait_stream(st)
        for dev, st in _AUX_STREAMS.items():     # keyed by the tensors' device (always indexed, unlike a bare "cuda")
            torch.cuda.current_stream(dev).wait_stream(st)
            for p, g in _PENDING_AUX_GRADS.pop(dev, []):     # (see _TailWeightsReplay._backward)
                p.grad = g.detach().clone() if p.grad is None else p.grad + g
            _AUX_PARAM_EVENT.pop(dev, None)

    def _fused_tail_ok(self) -> bool:
        """The single-launch tail is compiled for the default widths (feature / PE dim 172, time dim 100); other shapes (and
        LSTEP_TORCH_TAIL=1, the A/B switch) take the library-GEMM tail.  Not with ``use_dropout``: see ``combining_pe_raw_feat``."""
        return (os.environ.get("LSTEP_TORCH_TAIL", "0") != "1" and not self.use_dropout
                and (self.ld_edge, self.ld_node, self.ld_pe, self.ld_self) == (272, 176, 272, 176))

    def _combined_tail(self, x_edge, x_node, x_pe, own, fused: bool = False):
        """All dense layers after the gather stage, with the purely linear stretches pre-multiplied.

        After the relu of the edge channel nothing non-linear touches the A/N branch any more
        (edge_mlp_2 -> node_mlp -> out_node_emb, models/LSTEP.py:170,219,264), so
            out = Wo_a (Wn_a x_node + Wn_b (W2 relu(h1) + b2) + bn) + Wo_b q + bo
                = [Wo_a Wn_a | Wo_a Wn_b W2 | Wo_b] . cat[x_node, relu(h1), q] + const
        is ONE [616 -> 172] GEMM instead of three (272->272, 444->172, 344->172); likewise
        self_update_neighbor_pe(own) + pe_neighbor_mlp_2(relu(p1)) is one [344 -> 172] GEMM.  The composed matrices
        are rebuilt from the live parameters every call (25 MFLOP), so autograd yields the gradients of the original
        parameters; state_dict is unchanged.  Exact in real arithmetic, <= 1e-6 in fp32 (golden-checked).
        All operands are zero-padded to 16-aligned widths (inputs by the gather kernel, weights here): the padding
        columns stay exactly 0 through relu / tanh / residual, so results are unchanged."""
        wait_aux_param_update(x_edge.device)     # (an optimiser step of these layers may still be running on the auxiliary stream)
        Fd, P, D = self.feat_dim, self.pe_dim, self.time_dim
        Ce, Fn, Cp, Pp = self.ld_edge, self.ld_node, self.ld_pe, self.ld_self   # 16-aligned widths (272, 176, 272, 176)
        dims = (Fd, D + Fd, P, P + D, Ce, Fn, Cp, Pp)
        params = (self.edge_mlp_1.weight, self.edge_mlp_1.bias, self.edge_agg.weight, self.edge_agg.bias,
                  self.edge_mlp_2.weight, self.edge_mlp_2.bias, self.node_mlp.weight, self.node_mlp.bias, self.out_node_emb.weight,
                  self.out_node_emb.bias, self.self_update_neighbor_pe.weight, self.self_update_neighbor_pe.bias,
                  self.pe_neighbor_mlp_1.weight, self.pe_neighbor_mlp_1.bias, self.pe_neighbor_mlp_2.weight, self.pe_neighbor_mlp_2.bias)
        tw = None
        if fused and x_edge.is_cuda and torch.is_grad_enabled() and os.environ.get("LSTEP_NO_GRAPH") != "1":
            tw = self.__dict__.get("_tail_weight_graph")
            if tw is None or tw.dims != dims or any(a is not b for a, b in zip(tw.params, params)):
                tw = self.__dict__["_tail_weight_graph"] = _TailWeightsGraph(dims, params)
            if tw.live > 0:       # a second call inside one autograd graph (reference-style loop: three calls per iteration): eager
                tw = None
        if tw is not None:
            W1p, b1p, Wn1p, bn1p, Wq, bq, Wall, constp, w1t, wn1t, wqt, wallt = _TailWeightsReplay.apply(tw, *params)
        else:
            W1p, b1p, Wn1p, bn1p, Wq, bq, Wall, constp, w1t, wn1t, wqt, wallt = _TailWeights.apply(dims, *params)
        if fused:   # x_node / own are the wide [x_node | h1 | q] / [own | p1] buffers of the gather stage
            aux = _aux_stream(x_edge.device) if (tw is not None and self.__dict__.get("aux_wgrad_stream", False)) else None
            if tw is not None:
                tw.aux = aux
            return _FusedTail.apply(x_edge, x_pe, x_node, own, W1p, b1p, Wn1p, bn1p, Wq, bq, Wall, constp, w1t, wn1t, wqt, wallt,
                                    tw.gin if tw is not None else None, aux)
        h1 = fast_linear(x_edge, W1p, b1p, relu=True)                                          # [B, Ce]
        p1 = fast_linear(x_pe, Wn1p, bn1p, relu=True)                                          # [B, Pp]
        q = own + torch.tanh(fast_linear(torch.cat([own, p1], dim=-1), Wq, bq))                # [B, Pp]
        return fast_linear(torch.cat([x_node, h1, q], dim=-1), Wall, constp)

    def compute_src_dst_node_temporal_embeddings(self, pe, src_node_ids, dst_node_ids, node_interact_times, num_neighbors: int = 20,
                                                 time_gap: int = 2000, spliced: SplicedRows = None):
        """DyGLib-style convenience wrapper (shape of reference ``models/GraphMixer.py:57-75``): both endpoints in ONE launch."""
        n = len(src_node_ids)
        if isinstance(src_node_ids, torch.Tensor):
            ids = torch.cat([src_node_ids, dst_node_ids])
            ts = torch.cat([node_interact_times, node_interact_times])
        else:
            ids = np.concatenate([src_node_ids, dst_node_ids])
            ts = np.concatenate([node_interact_times, node_interact_times])
        out = self.combining_pe_raw_feat(pe, ids, ts, num_neighbors, time_gap, spliced=spliced)
        return out[:n], out[n:]

    def _fourier_transform_pe_dropout(self, node_ids, pe, batch_idx):
        """``fourier_transform_pe(..., use_dropout=True)`` (models/LSTEP.py:131-133; no reference driver passes it): dropout on the filtered
        history and the history itself added back as a residual -- neither is linear in the filter, so the coefficient table does not
        apply: the transform written out with ``torch.fft`` (the mask of the short-history case as in ``fft_coefficients``)."""
        T, t_len = self.num_fft_batches, int(pe.shape[1])
        if t_len > T:
            raise RuntimeError(f"history holds {t_len} snapshots but num_fft_batches={T} (the reference's filter broadcast fails the same way)")
        x = pe[self._ids(node_ids)].to(torch.float32)                                   # [U, t, P]
        keep = None
        if t_len < T:
            x = F.pad(x, (0, 0, 0, T - t_len))
            keep = (torch.arange(T, device=x.device) < batch_idx).to(torch.float32)[None, :, None]
        masked = (lambda z: z * keep) if keep is not None else (lambda z: z)
        spectrum = masked(torch.fft.fft(x.to(torch.complex64), dim=1))
        filtered = masked(torch.fft.ifft(masked(self.fft_filter.weight.unsqueeze(0) * spectrum), dim=1)).real
        y = self.fft_dropout(filtered) + x
        return torch.einsum("utp,t->up", y, self.fft_agg.weight.reshape(-1))

    # ---- F (models/LSTEP.py:104-137)
    def fft_coefficients(self, t_len: int, batch_idx: int) -> torch.Tensor:
        """Real [T, P] table c with  fourier_transform_pe(x)[u, p] = sum_s c[s, p] * x[u, s, p].

        Reference pipeline: zero-pad to T, fft over time, (mask), * fft_filter.weight, (mask), ifft, (mask), real part,
        fft_agg over time; mask = 1 on indices < batch_idx and only when fewer than T snapshots are stored
        (models/LSTEP.py:108-113).  Everything is linear in x, so with m the mask, W the filter and a the fft_agg row,
            c[s, p] = Re( 1/T * sum_f m[f] W[f, p] e^{-2 pi i f s / T} * sum_t a[t] m[t] e^{+2 pi i f t / T} ).
        Built in complex128 from the live parameters by ``_FftCoefficients`` (hand-derived backward to fft_filter / fft_agg).
        """
        T = self.num_fft_batches
        dev = self.fft_agg.weight.device
        cache = getattr(self, "_dft_cache", None)
        if cache is None or cache[0] != (T, dev):
            k = torch.arange(T, device=dev, dtype=torch.float64)
            ang = (2.0 * math.pi / T) * torch.outer(k, k)
            e_pos = torch.polar(torch.ones_like(ang), ang)      # e^{+i 2 pi f t / T}, [f, t]
            self._dft_cache = cache = ((T, dev), k, e_pos, e_pos.conj().t().contiguous(), torch.ones(T, device=dev, dtype=torch.float64))
        _, k, e_pos, e_neg_t, ones = cache
        m = (k < batch_idx).to(torch.float64) if t_len < T else ones
        return _FftCoefficients.apply(self.fft_filter.weight, self.fft_agg.weight, m, e_pos, e_neg_t, T)

    def fourier_transform_pe(self, node_ids, pe, batch_idx, use_dropout=False, use_mixer=False):
        """``pe`` is the PE history ``[N+1, t, P]`` (any strides with unit last stride); returns ``[U, P]``."""
        if pe.dim() != 3 or pe.shape[2] != self.pe_dim:
            raise ValueError("pe history must be [N+1, t, P]")
        if use_dropout:
            return self._fourier_transform_pe_dropout(node_ids, pe, batch_idx)
        t_len = int(pe.shape[1])
        if t_len > self.num_fft_batches:
            raise RuntimeError(f"history holds {t_len} snapshots but num_fft_batches={self.num_fft_batches} "
                               "(the reference's filter broadcast fails the same way)")
        if t_len == 0:
            return torch.zeros((len(node_ids), self.pe_dim), dtype=torch.float32, device=self.device)
        hist = pe.detach()
        if hist.dtype != torch.float32 or hist.stride(2) != 1 or hist.stride(0) % 4 or hist.stride(1) % 4 or hist.data_ptr() % 16:
            hist = hist.float().contiguous()
        geom = (int(hist.stride(0)), int(hist.stride(1)), t_len, 0, t_len, self.pe_dim)
        return self.filter_history(hist, geom, self._ids(node_ids), batch_idx)

    def filter_history(self, hist_base: torch.Tensor, geom, ids: torch.Tensor, batch_idx: int, mask: torch.Tensor = None,
                       oldest: torch.Tensor = None, splice=None, live: torch.Tensor = None, ring=None):
        """Shared by the drop-in method above and the device ring of ``lstep_amd.engine`` (geom = strides/rotation; ``mask`` = the
        ring's change bits, ``oldest`` = its table of the window's oldest snapshot when the slots only hold changed rows, see
        ``HistoryRing``)."""
        if splice is not None:     # (table, slot_of): the run kernel also writes table[ids[u]] = out[u] and slot_of[ids[u]] = u
            table, slot_of = splice
            if (mask is None or table.dtype != torch.float32 or not table.is_contiguous() or table.shape[1] != geom[0]
                    or slot_of.dtype != torch.int32):
                raise ValueError("filter_history: the fused splice needs the change-mask path, a contiguous fp32 table and an int32 slot map")
        coef = self.fft_coefficients(geom[4], batch_idx)
        return _HistoryFilter.apply(coef, hist_base, geom, ids, mask, oldest, splice, live, ring)

    # ---- U1 + U2 (models/LSTEP.py:268-340).  Forward only: in the reference no gradient ever reaches these
    # parameters (the loss is taken before update_pe and the history is detached, train:233-275,306).
    def _segment_sum(self, pe, nseg, ent_seg, ent_row, ent_dt, exact: bool = False, live: torch.Tensor = None):
        """out[s] = sum over the entries of segment s of cat[pe[ent_row], time_feat(ent_dt)]  (lstep_segment_rows_sum).
        Library-GEMM consumers: rows are bucketed (``_bucket_rows``) and everything past the data is zero.  ``exact`` (the fused
        ``lstep_update_rows`` consumer, which reads exactly ``nseg`` rows): no bucket rows and NO memset -- every segment owns entries, so
        the kernel writes every row whole and zeroes only the rows it accumulates with atomics (356 MB less traffic in phase 2)."""
        lib = nat.load_library()
        P, D = self.pe_dim, self.time_dim
        exact = exact and self.ld_pe == P + D
        if exact:
            out = torch.empty((nseg, self.ld_pe), dtype=torch.float32, device=pe.device)
        else:
            out = torch.zeros((self._bucket_rows(nseg), self.ld_pe), dtype=torch.float32, device=pe.device)
        with torch.cuda.device(pe.device):
            ws, ws_bytes = nat.segment_workspace(pe.device, ent_row.numel(), P, D)
            nat.check(lib.lstep_segment_rows_sum(nat.ptr(pe), P, P, nat.ptr(self.time_encoder.w.weight), nat.ptr(self.time_encoder.w.bias), D,
                                                 nat.ptr(ent_seg), nat.ptr(ent_row), nat.ptr(ent_dt), ent_row.numel(), nat.ptr(out), self.ld_pe,
                                                 2 if exact else 0, nat.ptr(live), nat.ptr(ws), ws_bytes, nat.current_stream()))
        return out

    MLP_ROW_BLOCK = 65536   # hipBLASLt's fp32 rate for these skinny GEMMs swings 50-115 TFLOP/s with M; 65536-row blocks sit at ~100 (tools/gemm_m.py)

    def _update_mlp(self, agg):
        """pe_mlp_2(relu(pe_mlp_1(agg))) on row-padded operands -> [n, ld_self] (padding columns 0).
        ``agg`` may carry extra bucket rows (see ``_bucket_rows``); they are computed and ignored."""
        w1t, b1, w2t, b2 = self._update_weights()
        Pp = self.ld_self
        n, blk = agg.shape[0], self.MLP_ROW_BLOCK
        if n <= blk:
            return torch.addmm(b2, _addmm_relu(b1, agg, w1t), w2t)                     # relu runs in the GEMM epilogue
        out = torch.empty((n, Pp), dtype=torch.float32, device=agg.device)
        for i in range(0, n, blk):
            torch.addmm(b2, _addmm_relu(b1, agg[i:i + blk], w1t), w2t, out=out[i:i + blk])
        return out

    def _padded_cached(self, name: str, params, build):
        """Padded copies of forward-only weights (no gradient ever reaches the update_pe layers, so they only change when the user
        loads or edits them): rebuilt when any source tensor's version counter or storage changed."""
        key = tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.setdefault("_pad_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            hit = cache[name] = (key, build())
        return hit[1]

    def _update_weights(self):
        Cp, Pp = self.ld_pe, self.ld_self
        m1, m2 = self.pe_mlp_1, self.pe_mlp_2
        return self._padded_cached("update_mlp", (m1.weight, m1.bias, m2.weight, m2.bias), lambda: (
            _pad2(m1.weight.detach(), Pp, Cp).t(), _pad1(m1.bias.detach(), Pp), _pad2(m2.weight.detach(), Pp, Pp).t(), _pad1(m2.bias.detach(), Pp)))

    def _update_weights_pre(self):
        """Operands of ``lstep_update_rows_pre``: pe_mlp_1.weight split into its PE half (transposed, [172, 176]: the right factor of the
        batch-node product) and its time half ([176, 112] zero-padded), the biases and pe_mlp_2.weight [176, 176]."""
        P, D, Pp = self.pe_dim, self.time_dim, self.ld_self
        m1, m2 = self.pe_mlp_1, self.pe_mlp_2

        def build():
            w = m1.weight.detach()
            w1a_t = torch.zeros((P, Pp), dtype=torch.float32, device=w.device)
            w1a_t[:, :w.shape[0]] = w[:, :P].t()
            w1b = torch.zeros((Pp, 112), dtype=torch.float32, device=w.device)
            w1b[:w.shape[0], :D] = w[:, P:P + D]
            return (w1a_t, w1b, _pad1(m1.bias.detach(), Pp), _pad2(m2.weight.detach(), Pp, Pp).contiguous(), _pad1(m2.bias.detach(), Pp))
        return self._padded_cached("update_mlp_pre", (m1.weight, m1.bias, m2.weight, m2.bias), build)

    def _update_rows(self, pe, ids, agg, with_self: bool, mirror=None, live: torch.Tensor = None, ring=None, mirror_shard=(1, 0)):
        """``lstep_update_rows``: pe[ids] += tanh(pe_mlp_2(relu(pe_mlp_1(agg))) [+ self_update_pe(pe[ids])]) in place, one launch."""
        lib = nat.load_library()
        Pp = self.ld_self
        w1t, b1, w2t, b2 = self._update_weights()
        w1, w2 = self._padded_cached("update_mlp_rowmajor", (self.pe_mlp_1.weight, self.pe_mlp_2.weight),
                                     lambda: (w1t.t().contiguous(), w2t.t().contiguous()))
        ws = bs = None
        if with_self:
            su = self.self_update_pe
            ws, bs = self._padded_cached("self_update_sq", (su.weight, su.bias),
                                         lambda: (_pad2(su.weight.detach(), Pp, Pp).contiguous(), _pad1(su.bias.detach(), Pp)))
        ids = ids.contiguous()
        with torch.cuda.device(pe.device):
            nat.check(lib.lstep_update_rows(nat.ptr(agg), int(agg.stride(0)), nat.ptr(ids), ids.numel(), nat.ptr(w1), nat.ptr(b1), nat.ptr(w2),
                                            nat.ptr(b2), nat.ptr(ws), nat.ptr(bs), nat.ptr(pe), nat.ptr(mirror), self.pe_dim, nat.ptr(live),
                                            ring, int(mirror_shard[0]), int(mirror_shard[1]), nat.current_stream()))

    @classmethod
    def _bucket_rows(cls, n: int) -> int:
        """Row count rounded up to a coarse bucket: the number of updated rows changes every batch, and every new GEMM
        M costs ~150 us of hipBLASLt heuristic lookup on the host; a few repeating M values avoid that."""
        q = 1024 if n <= cls.MLP_ROW_BLOCK else cls.MLP_ROW_BLOCK
        return max(q, (n + q - 1) // q * q)

    def write_rows(self, pe, ids, rows):
        """In-place ``pe[ids] = rows`` (models/LSTEP.py:303,339)."""
        lib = nat.load_library()
        with torch.cuda.device(pe.device):
            rows = rows.contiguous()
            nat.check(lib.lstep_scatter_rows(nat.ptr(pe), self.pe_dim, nat.ptr(ids), ids.numel(), nat.ptr(rows), nat.current_stream()))

    def apply_residual_tanh(self, pe, ids, z):
        """In-place ``pe[ids] += tanh(z[:, :P])``: residual + tanh + row write of models/LSTEP.py:299-303 / :335-339 in one
        kernel; ``z`` may be row-padded (its row stride is passed on)."""
        lib = nat.load_library()
        z = z.contiguous()
        with torch.cuda.device(pe.device):
            nat.check(lib.lstep_residual_tanh_rows(nat.ptr(pe), self.pe_dim, nat.ptr(ids), ids.numel(), nat.ptr(z), int(z.shape[1]),
                                                   nat.current_stream()))

    @torch.no_grad()
    def update_pe_phase1(self, pe, bn, src, dst, t, now32: float, shard=None, presorted=None, fused: bool = False, owned_idx=None,
                         mirror=None):
        """U1 (LSTEP.py:277-303): every batch edge sends cat[pe[other endpoint], time_feat] to both endpoints.
        Returns (ids, z) with the new row = pe[ids] + tanh(z), WITHOUT writing (``fused=True``: writes the rows in place with
        ``lstep_update_rows`` and returns ids only); ``shard=(W, r)`` restricts the work to
        nodes with id % W == r; ``presorted=(order, inverse, counts)`` reuses the caller's stable sort of cat[src, dst]
        (the engine derives the batch-node set and the segments from one sort; then ``bn`` must be that node set);
        ``owned_idx`` (with ``shard``): the positions in ``bn`` of the nodes this rank owns, if the caller already has them -- the
        sharded update then needs no host synchronisation either."""
        if (fused and presorted is not None and (shard is None or owned_idx is not None) and isinstance(now32, torch.Tensor)
                and presorted[0].dtype == torch.int32 and os.environ.get("LSTEP_TORCH_ENTRIES") != "1"):
            # engine fast path: the grouping of cat[src, dst] by batch node is already there (int32 order / segment ids); one kernel
            # builds the message list, one sums the segments, one applies the MLP and writes the rows
            lib = nat.load_library()
            order32, seg32 = presorted[0], presorted[1]
            n2 = order32.numel()
            ent_row = torch.empty(n2, dtype=torch.int32, device=pe.device)
            ent_dt = torch.empty(n2, dtype=torch.float32, device=pe.device)
            with torch.cuda.device(pe.device):
                nat.check(lib.lstep_update_entries_p1(nat.ptr(order32), n2, nat.ptr(src), nat.ptr(dst), nat.ptr(t), nat.ptr(now32), src.numel(),
                                                      nat.ptr(ent_row), nat.ptr(ent_dt), nat.current_stream()))
            agg = self._segment_sum(pe, bn.numel(), seg32, ent_row, ent_dt, exact=True)
            if shard is None:
                self._update_rows(pe, bn, agg, with_self=True, mirror=mirror)
                return bn
            # sharded: the message sums of all batch nodes are cheap (2 rows per batch edge) and every rank has the inputs; only the
            # rows this rank owns go through the MLP and are written
            ids = bn[owned_idx]
            self._update_rows(pe, ids, agg[owned_idx], with_self=True)
            return ids
        # float32 scalar - float64 -> float64 -> .float(); now32 may be a 0-d float32 device tensor (no host round trip)
        dt1 = ((now32.to(torch.float64) if isinstance(now32, torch.Tensor) else now32) - t).to(torch.float32)
        if presorted is None:
            keys_s, order = torch.sort(torch.cat([src, dst]), stable=True)
            nodes, inverse, counts = torch.unique_consecutive(keys_s, return_inverse=True, return_counts=True)
            # only rows listed in bn are updated (LSTEP.py:292): map the receiving nodes onto bn, drop the others
            bn_sorted = torch.sort(bn).values
            pos = torch.searchsorted(bn_sorted, nodes).clamp(max=max(bn_sorted.numel() - 1, 0))
            listed = bn_sorted[pos] == nodes if bn_sorted.numel() else torch.zeros_like(nodes, dtype=torch.bool)
            ids = bn_sorted
            seg_of_node = torch.where(listed, pos, torch.full_like(pos, -1))
        else:
            order, inverse, counts = presorted
            ids = bn
            seg_of_node = torch.arange(bn.numel(), device=bn.device)
        ent_seg = seg_of_node[inverse]
        ent_row = torch.cat([dst, src])[order]
        ent_dt = torch.cat([dt1, dt1])[order]
        keep_e = ent_seg >= 0
        if shard is not None:
            mine = (ids % shard[0]) == shard[1]
            newpos = torch.cumsum(mine, 0) - 1
            keep_e = keep_e & mine[ent_seg.clamp(min=0)]
            ent_seg = newpos[ent_seg.clamp(min=0)]
            ids = ids[mine]
        if presorted is None or shard is not None:
            ent_seg, ent_row, ent_dt = ent_seg[keep_e], ent_row[keep_e], ent_dt[keep_e]
        # exact (no memset) only when every listed node is known to own entries: the engine's presorted batch-node set
        agg = self._segment_sum(pe, ids.numel(), ent_seg.to(torch.int32), ent_row.to(torch.int32), ent_dt.contiguous(),
                                exact=fused and presorted is not None)
        n = ids.numel()
        if fused:   # MLP + self term + tanh + residual + in-place row write in one launch
            self._update_rows(pe, ids, agg, with_self=True, mirror=mirror)
            return ids
        own = torch.zeros((agg.shape[0], self.pe_dim), dtype=torch.float32, device=pe.device)
        own[:n] = pe[ids]
        Pp = self.ld_self
        su = self.self_update_pe
        ws, bs = self._padded_cached("self_update", (su.weight, su.bias),
                                     lambda: (_pad2(su.weight.detach(), Pp, self.pe_dim), _pad1(su.bias.detach(), Pp)))
        z = F.linear(own, ws, bs) + self._update_mlp(agg)
        return ids, z[:n]

    @torch.no_grad()
    def update_pe_phase2(self, pe, bn, t, now32: float, num_neighbors: int, shard=None, fused: bool = False, mirror=None):
        """U2 (LSTEP.py:305-339): push the updated PE of each batch node to its K most recent neighbours.
        ``bn`` (U rows) is zipped with the B edge times: row i uses t[i]; rows >= min(U, B) stay padding.
        Sets pe[0] = 0 (LSTEP.py:317) before reading.  Returns (touched ids, z) with new row = pe[id] + tanh(z), WITHOUT
        writing (the self_update_pe term is dead code in the reference, :334-335)."""
        dev = pe.device
        P, D = self.pe_dim, self.time_dim
        U = bn.numel()
        if getattr(self.neighbor_sampler, "sample_neighbor_strategy", "recent") == "recent":
            nbr, _, nt = self.neighbor_sampler.sample_device(bn, t, num_neighbors)
        else:   # RNG-defined strategies draw on the host, in the reference's call order (one call per update_pe, models/LSTEP.py:306)
            a, _, c = self.neighbor_sampler.get_historical_neighbors(bn.cpu().numpy(), t.cpu().numpy(), num_neighbors)
            nbr = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)
            nt = torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32)).to(dev)
        key = nbr.reshape(-1)
        rows = pe.shape[0]
        pe[0].zero_()
        if fused and isinstance(now32, torch.Tensor) and os.environ.get("LSTEP_TORCH_ENTRIES") != "1":
            return self._phase2_native(pe, bn, nbr, nt, now32, num_neighbors, shard, mirror)
        real = key != 0
        if shard is not None:
            real = real & ((key % shard[0]) == shard[1])
        own_row0 = shard is None or shard[1] == 0                # row 0 belongs to shard 0
        # group the real entries by touched row: dropped entries get the sentinel key `rows` and sort to the end
        keys32 = torch.where(real, key, torch.full_like(key, rows)).to(torch.int32)
        _, order, seg, uniq, (_, n_real, nseg) = nat.group_by_key(keys32, max(1, int(rows + 1).bit_length()), rows)
        n_zero = (key.numel() - n_real) if shard is None else (int((key == 0).sum()) if own_row0 else 0)
        src_e = order[:n_real].long()
        inverse = seg[:n_real]
        touched = uniq[:nseg].long()
        ent_row = bn[src_e // num_neighbors].to(torch.int32)
        now_t = now32 if isinstance(now32, torch.Tensor) else torch.tensor(now32, dtype=torch.float32, device=dev)
        ent_dt = now_t - nt.reshape(-1)[src_e]                   # float32 - float32 (LSTEP.py:314)
        nseg = touched.numel()
        if own_row0 and n_zero > 0:
            # row 0 collects cat[pe[source], 0] from every padded slot: a weighted column sum instead of a hot segment.
            # It goes first (ids are sorted): segment 0 has no entries and its aggregate is filled in afterwards.
            touched = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), touched])
            agg2 = self._segment_sum(pe, nseg + 1, inverse + 1, ent_row, ent_dt, exact=fused)
            agg2[0].zero_()          # segment 0 (row 0) has no entries of its own: its aggregate is the padding sum below
            lib = nat.load_library()
            part = torch.empty((int(lib.lstep_padding_rows_sum_blocks(U)), P), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):   # sum over the rows of (their number of padded slots) * pe[source row]
                nat.check(lib.lstep_padding_rows_sum(nat.ptr(nbr), int(num_neighbors), nat.ptr(bn), U, nat.ptr(pe), P, int(pe.stride(0)),
                                                     nat.ptr(part), nat.current_stream()))
            agg2[0, :P] = part.sum(dim=0)
        else:
            agg2 = self._segment_sum(pe, nseg, inverse, ent_row, ent_dt, exact=fused)
        if fused:
            self._update_rows(pe, touched, agg2, with_self=False, mirror=mirror)
            return touched
        return touched, self._update_mlp(agg2)[:touched.numel()]

    def _phase2_native(self, pe, bn, nbr, nt, now32, num_neighbors, shard, mirror=None):
        """Phase 2 with the key / entry lists built by native kernels (lstep_update_keys_p2, lstep_update_entries_p2): ~10 launches
        instead of ~35 framework ones; same grouping, same order of summation."""
        lib = nat.load_library()
        dev, P, U = pe.device, self.pe_dim, bn.numel()
        rows = pe.shape[0]
        n = nbr.numel()
        world, rank = shard if shard is not None else (1, 0)
        own_row0 = rank == 0                                     # row 0 belongs to shard 0
        keys32 = torch.empty(n, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_update_keys_p2(nat.ptr(nbr), n, rows, world, rank, nat.ptr(keys32), nat.current_stream()))
        _, order, seg, uniq, (_, n_real, nseg) = nat.group_by_key(keys32, max(1, int(rows + 1).bit_length()), rows)
        n_zero = (n - n_real) if shard is None else (int((nbr == 0).sum()) if own_row0 else 0)
        shift = 1 if (own_row0 and n_zero > 0) else 0
        ent_row = torch.empty(n_real, dtype=torch.int32, device=dev)
        ent_dt = torch.empty(n_real, dtype=torch.float32, device=dev)
        ent_seg = torch.empty(n_real, dtype=torch.int32, device=dev)
        touched = torch.empty(nseg + shift, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_update_entries_p2(nat.ptr(order), nat.ptr(seg), n_real, nat.ptr(bn), nat.ptr(nt), nat.ptr(now32), int(num_neighbors),
                                                  shift, nat.ptr(uniq), nseg, nat.ptr(ent_row), nat.ptr(ent_dt), nat.ptr(ent_seg), nat.ptr(touched),
                                                  nat.current_stream()))
        agg2 = self._segment_sum(pe, nseg + shift, ent_seg, ent_row, ent_dt, exact=True)
        if shift:
            # row 0 collects cat[pe[source], 0] from every padded slot: segment 0 has no entries of its own, its aggregate is the sum
            # over the rows of (their number of padded slots) * pe[source row]
            agg2[0].zero_()
            part = torch.empty((int(lib.lstep_padding_rows_sum_blocks(U)), P), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                nat.check(lib.lstep_padding_rows_sum(nat.ptr(nbr), int(num_neighbors), nat.ptr(bn), U, nat.ptr(pe), P, int(pe.stride(0)),
                                                     nat.ptr(part), nat.current_stream()))
            agg2[0, :P] = part.sum(dim=0)
        self._update_rows(pe, touched, agg2, with_self=False, mirror=mirror)
        return touched

    @torch.no_grad()
    def update_pe_device(self, pe, bn, n_live, src, dst, t, num_neighbors: int, presorted, changed=None, mirror=None, mirror_ring=None,
                         mirror_shard=(1, 0), owner=None, owned_idx=None, after_phase1=None, now32=None, owned_live=None):
        """``update_pe`` for the engine, with every data-dependent size left on the device: no host synchronisation, no second host
        thread, a fixed launch sequence.

        ``bn`` int64 [2 B] is the capacity-sized batch-node list of ``lstep_widen_ids`` (sorted unique endpoints, dead tail = node 0),
        ``n_live`` int32 [1] their number on the device, ``presorted = (order, seg)`` the int32 grouping of cat[src, dst] by batch node.
        Dead rows are the padding node 0: it has no history, so the sampler returns all-padding neighbourhoods for them, their keys drop
        out of the grouping, their change-mask marks hit row 0 (always marked) and their contribution to row 0's padding sum is
        K * pe[0] = 0 (pe[0] is zeroed first, models/LSTEP.py:317).  What does depend on the exact count reads it on the device:
        ``lstep_update_rows`` (which rows to write), ``lstep_segment_rows_sum`` (how many grouped slots are real) and the decision
        whether row 0 takes part in phase 2 at all (``lstep_update_entries_p2_dev``).  ``mirror_ring`` (``lstep_ring_ref_t*``): ``mirror``
        is the base of the history ring and the slot that receives the new rows is picked on the device.

        OWNER-COMPUTES form (``lstep_amd.parallel``, one process per GPU): ``owner = (W, r)`` -- this rank computes only the rows with
        id % W == r.  Phase 1: the message sums of all batch nodes are formed (two rows per batch edge, every rank has the inputs), the MLP
        and the write run for the batch nodes at positions ``owned_idx`` of ``bn`` (sized on the host by the caller, or -- ``owned_live``, an
        int32 [1] device count -- a capacity-sized list whose dead tail points at position 0: ``lstep_owner_partition``);
        ``after_phase1(ids)`` then exchanges the new rows so that ``pe[bn]`` holds every batch node's phase-1 value on every rank
        (phase 2's messages carry them).  Phase 2: the sampled slots whose NEIGHBOUR this rank owns are grouped and summed, only those
        rows go through the MLP; row 0 belongs to rank 0.  Everything stays device-sized."""
        lib = nat.load_library()
        dev, P, K = pe.device, self.pe_dim, int(num_neighbors)
        cap = bn.numel()
        rows = pe.shape[0]
        world, rank = (int(owner[0]), int(owner[1])) if owner is not None else (1, 0)
        if now32 is None:      # torch.Tensor([current_time]) of models/LSTEP.py:277: float32-rounded (the engine hands it over: lstep_batch_prepare)
            now32 = t.max().to(torch.float32).reshape(1)
        order32, seg32 = presorted
        n2 = order32.numel()
        # ---- phase 1 (LSTEP.py:277-303)
        ent_row = torch.empty(n2, dtype=torch.int32, device=dev)
        ent_dt = torch.empty(n2, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_update_entries_p1(nat.ptr(order32), n2, nat.ptr(src), nat.ptr(dst), nat.ptr(t), nat.ptr(now32), src.numel(),
                                                  nat.ptr(ent_row), nat.ptr(ent_dt), nat.current_stream()))
        agg = self._segment_sum(pe, cap, seg32, ent_row, ent_dt, exact=True)
        if owner is None:
            self._update_rows(pe, bn, agg, with_self=True, mirror=mirror, live=n_live, ring=mirror_ring, mirror_shard=mirror_shard)
            if changed is not None:
                changed(bn, mirror is not None)       # (the dead tail marks row 0, which every update_pe rewrites anyway)
        else:
            ids1 = bn.index_select(0, owned_idx)
            if ids1.numel():
                self._update_rows(pe, ids1, agg.index_select(0, owned_idx), with_self=True, mirror=mirror, live=owned_live, ring=mirror_ring,
                                  mirror_shard=mirror_shard)
            if changed is not None:
                changed(ids1, mirror is not None)
            if after_phase1 is not None:
                after_phase1(ids1)
        # ---- phase 2 (LSTEP.py:305-339): row i of bn is zipped with the i-th EDGE time, rows >= min(U, B) stay padding
        nbr, _, nt = self.neighbor_sampler.sample_device(bn, t, K)
        n = nbr.numel()
        pe[0].zero_()
        keys32 = torch.empty(n, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.lstep_update_keys_p2(nat.ptr(nbr), n, rows, world, rank, nat.ptr(keys32), nat.current_stream()))
        _, order, seg, uniq, summary = nat.group_by_key(keys32, max(1, int(rows + 1).bit_length()), rows, wait=None)
        tcap = min(n, rows) + 1                                # row 0's reserved segment + at most one per slot / per table row
        ent_row = torch.empty(n, dtype=torch.int32, device=dev)
        ent_dt = torch.empty(n, dtype=torch.float32, device=dev)
        ent_seg = torch.empty(n, dtype=torch.int32, device=dev)
        touched = torch.empty(tcap, dtype=torch.int64, device=dev)
        counts = torch.empty(2, dtype=torch.int32, device=dev)    # {touched rows besides row 0, does row 0 take part}
        # pre-multiplied form (lstep_update_rows_pre): the messages' PE rows are the batch nodes' rows, so pe_mlp_1's PE half is applied to
        # those cap rows once and the segment sums run over the products
        premul = (os.environ.get("LSTEP_UPDATE_NO_PREMUL") != "1" and self.ld_pe == P + self.time_dim and self.ld_self == 176
                  and self.time_dim <= 112 and self.time_dim % 4 == 0)
        with torch.cuda.device(dev):
            # (owner-computes: the row-0 flag compares live_rows * K with the number of REAL slots; with the other owners' slots dropped
            # from the grouping it would always read "padding present" -- rank 0 takes the flag from a second, unfiltered count below)
            nat.check(lib.lstep_update_entries_p2_dev(nat.ptr(order), nat.ptr(seg), nat.ptr(summary), nat.ptr(n_live), n, tcap,
                                                      None if premul else nat.ptr(bn), nat.ptr(nt), nat.ptr(now32), K, nat.ptr(uniq),
                                                      nat.ptr(ent_row), nat.ptr(ent_dt), nat.ptr(ent_seg), nat.ptr(touched), nat.ptr(counts),
                                                      nat.current_stream()))
        row0_live = counts[1:2]
        if owner is not None:
            if rank != 0:
                row0_live = torch.zeros(1, dtype=torch.int32, device=dev)          # row 0 belongs to rank 0
            else:       # does any slot of a LIVE row hold the padding id?  (models/LSTEP.py:324: 0 in unique(neighbour ids))
                live_slots = (torch.arange(cap, device=dev) < n_live).unsqueeze(1) & (nbr == 0)
                row0_live = live_slots.any().to(torch.int32).reshape(1)
        if premul:
            D, Pp = self.time_dim, self.ld_self
            w1a_t, w1b, b1, w2, b2 = self._update_weights_pre()
            y = torch.mm(pe.index_select(0, bn), w1a_t)                       # [cap, 176]; dead rows are node 0: pe[0] = 0 -> 0
            agg2 = torch.empty((tcap, Pp + D), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                ws, ws_bytes = nat.segment_workspace(dev, n, Pp, D)
                nat.check(lib.lstep_segment_rows_sum(nat.ptr(y), Pp, Pp, nat.ptr(self.time_encoder.w.weight), nat.ptr(self.time_encoder.w.bias), D,
                                                     nat.ptr(ent_seg), nat.ptr(ent_row), nat.ptr(ent_dt), n, nat.ptr(agg2), Pp + D, 2,
                                                     nat.ptr(summary[1:2]), nat.ptr(ws), ws_bytes, nat.current_stream()))
            # row 0's aggregate: the block sums of (number of padded slots) x (row's product), added in block order; zero time part
            nblk = int(lib.lstep_padding_rows_sum_blocks(cap)) if rank == 0 else 0
            part = torch.empty((max(nblk, 1), Pp), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                if nblk:
                    nat.check(lib.lstep_padding_rows_sum(nat.ptr(nbr), K, None, cap, nat.ptr(y), Pp, Pp, nat.ptr(part), nat.current_stream()))
                nat.check(lib.lstep_padding_rows_finish(nat.ptr(part), nblk, Pp, nat.ptr(agg2), Pp + D, nat.current_stream()))
            for ids_, agg_, live_ in ((touched[1:], agg2[1:], counts[0:1]), (touched[:1], agg2[:1], row0_live)):
                with torch.cuda.device(dev):
                    nat.check(lib.lstep_update_rows_pre(nat.ptr(agg_), int(agg_.stride(0)), nat.ptr(ids_), ids_.numel(), nat.ptr(w1b), nat.ptr(b1),
                                                        nat.ptr(w2), nat.ptr(b2), nat.ptr(pe), nat.ptr(mirror), P, D, nat.ptr(live_), mirror_ring,
                                                        int(mirror_shard[0]), int(mirror_shard[1]), nat.current_stream()))
            if changed is not None:
                changed(touched, mirror is not None)
            return pe
        agg2 = self._segment_sum(pe, tcap, ent_seg, ent_row, ent_dt, exact=True, live=summary[1:2])
        # row 0 collects cat[pe[source], 0] from every padded slot: segment 0 has no entries of its own, its aggregate is the sum over
        # the rows of (their number of padded slots) * pe[source row]
        nblk = int(lib.lstep_padding_rows_sum_blocks(cap)) if rank == 0 else 0
        part = torch.empty((max(nblk, 1), P), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if nblk:
                nat.check(lib.lstep_padding_rows_sum(nat.ptr(nbr), K, nat.ptr(bn), cap, nat.ptr(pe), P, int(pe.stride(0)), nat.ptr(part),
                                                     nat.current_stream()))
            nat.check(lib.lstep_padding_rows_finish(nat.ptr(part), nblk, P, nat.ptr(agg2), int(agg2.shape[1]), nat.current_stream()))
        self._update_rows(pe, touched[1:], agg2[1:], with_self=False, mirror=mirror, live=counts[0:1], ring=mirror_ring, mirror_shard=mirror_shard)
        self._update_rows(pe, touched[:1], agg2[:1], with_self=False, mirror=mirror, live=row0_live, ring=mirror_ring, mirror_shard=mirror_shard)
        if changed is not None:
            changed(touched, mirror is not None)
        return pe

    @torch.no_grad()
    def update_pe(self, pe, node_ids, edge_ids, batch_src_node_ids, batch_dst_node_ids, node_interact_times, current_time,
                  num_neighbors: int = 30, time_gap: int = 2000, presorted=None, changed=None, mirror=None):
        """``changed`` (optional callable): receives the int64 ids of the rows each phase wrote (the device ring's change mask) and
        whether they were also written into ``mirror`` (optional second table, the batch's history slot: fused path only)."""
        if not (pe.is_cuda and pe.dtype == torch.float32 and pe.is_contiguous()):
            raise ValueError("update_pe needs a contiguous float32 GPU table (it is mutated in place)")
        if pe.dim() != 2 or pe.shape[1] != self.pe_dim or pe.shape[0] < self.neighbor_sampler.num_rows:
            raise ValueError(f"pe must be [>= {self.neighbor_sampler.num_rows}, {self.pe_dim}]")
        self._check_rows(node_ids)
        bn = self._ids(node_ids)
        src, dst = self._ids(batch_src_node_ids), self._ids(batch_dst_node_ids)
        t = self._times(node_interact_times)
        # torch.Tensor([current_time]) rounds to float32 first (LSTEP.py:277); a tensor current_time (e.g. ts.max()) stays on the device
        now32 = current_time.detach().to(device=pe.device, dtype=torch.float32).reshape(()) if isinstance(current_time, torch.Tensor) \
            else float(np.float32(current_time))
        if self._fused_tail_ok() and os.environ.get("LSTEP_TORCH_UPDATE") != "1":
            ids = self.update_pe_phase1(pe, bn, src, dst, t, now32, presorted=presorted, fused=True, mirror=mirror)
            if changed is not None:
                changed(ids, mirror is not None)
            ids = self.update_pe_phase2(pe, bn, t, now32, num_neighbors, fused=True, mirror=mirror)
            if changed is not None:
                changed(ids, mirror is not None)
            return pe
        ids, z = self.update_pe_phase1(pe, bn, src, dst, t, now32, presorted=presorted)
        self.apply_residual_tanh(pe, ids, z)
        if changed is not None:
            changed(ids, False)
        ids, z = self.update_pe_phase2(pe, bn, t, now32, num_neighbors)
        self.apply_residual_tanh(pe, ids, z)
        if changed is not None:
            changed(ids, False)
        return pe
