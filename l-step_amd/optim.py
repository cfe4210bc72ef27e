"""Single-kernel Adam for the L-STEP parameter set.

The reference optimises with plain ``torch.optim.Adam`` (``utils/utils.py:49-67``).  L-STEP has one complex parameter
(``fft_filter.weight``, complex64); Adam treats a complex tensor as its real view, so it enters through ``view_as_real`` (same storage),
its gradient likewise.  The update is ONE native launch (``lstep_adam_step``, csrc/adam.hip: the ~45 tensor pointers travel in the kernel
arguments, a workgroup takes 1024 elements): a few microseconds, where ``torch._fused_adam_`` takes 40 (a workgroup there loops over
64 K elements) and ``torch.optim.Adam.step`` adds ~0.3 ms of host bookkeeping.  Same arithmetic as ``torch._fused_adam_`` (no amsgrad,
L2-style weight decay), checked against ``torch.optim.Adam`` in tests/test_hip_parity.py.  ``LSTEP_TORCH_ADAM=1`` routes the step
through ``torch._fused_adam_`` for A/B.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _native as nat


class FusedAdam:
    def __init__(self, params, lr: float = 1e-4, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8):
        self._params = [p for p in params if p.requires_grad]
        if not self._params:
            raise ValueError("optimizer got an empty parameter list")
        if not all(p.is_cuda for p in self._params):
            raise ValueError("FusedAdam is a GPU kernel: parameters must live on the GPU")
        self.lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), (float(betas[0]), float(betas[1])), float(eps)
        dev = self._params[0].device
        self._real = [torch.view_as_real(p.data) if p.is_complex() else p.data for p in self._params]
        self._exp_avg = [torch.zeros_like(r) for r in self._real]
        self._exp_avg_sq = [torch.zeros_like(r) for r in self._real]
        self._steps = torch.zeros(len(self._params), dtype=torch.float32, device=dev)   # one counter per parameter, bumped in one launch
        self._step_views = [self._steps[i] for i in range(len(self._params))]
        self._active_key, self._active = None, None

    def zero_grad(self, set_to_none: bool = True):
        for p in self._params:
            p.grad = None

    def _lists(self, key):
        """Tensor lists of the parameters that currently have a gradient (the set is stable from step to step)."""
        if key != self._active_key:
            idx = [i for i, has in enumerate(key) if has]
            self._active_key = key
            self._active = (idx, [self._real[i] for i in idx], [self._exp_avg[i] for i in idx], [self._exp_avg_sq[i] for i in idx],
                            [self._step_views[i] for i in idx], torch.tensor([1.0 if has else 0.0 for has in key], device=self._steps.device))
        return self._active

    @torch.no_grad()
    def step(self):
        key = tuple(p.grad is not None for p in self._params)
        idx, params, exp_avg, exp_avg_sq, steps, bump = self._lists(key)
        if not idx:
            return
        grads = []
        for i in idx:
            g = self._params[i].grad
            grads.append(torch.view_as_real(g) if g.is_complex() else g)
        self._steps.add_(bump)   # parameters without a gradient keep their step count, like torch.optim.Adam
        if os.environ.get("LSTEP_TORCH_ADAM") == "1" or any(g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
            torch._fused_adam_(params, grads, exp_avg, exp_avg_sq, [], steps, amsgrad=False, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                               weight_decay=self.weight_decay, eps=self.eps, maximize=False, grad_scale=None, found_inf=None)
            return
        lib = nat.load_library()
        dev = self._steps.device
        cap = 48                  # LSTEP_ADAM_MAX_TENSORS
        with torch.cuda.device(dev):
            for lo in range(0, len(idx), cap):
                hi = min(len(idx), lo + cap)
                n = hi - lo
                arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts[lo:hi]])  # noqa: E731
                numel = (ctypes.c_int64 * n)(*[t.numel() for t in params[lo:hi]])
                nat.check(lib.lstep_adam_step(n, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), numel,
                                              self._steps.data_ptr(), (ctypes.c_int32 * n)(*idx[lo:hi]), self.lr, self.betas[0], self.betas[1], self.eps,
                                              self.weight_decay, nat.current_stream()))

    def state_dict(self):
        return {"lr": self.lr, "weight_decay": self.weight_decay, "betas": self.betas, "eps": self.eps, "steps": self._steps.clone(),
                "exp_avg": [t.clone() for t in self._exp_avg], "exp_avg_sq": [t.clone() for t in self._exp_avg_sq]}

    def load_state_dict(self, sd):
        self.lr, self.weight_decay, self.betas, self.eps = sd["lr"], sd["weight_decay"], tuple(sd["betas"]), sd["eps"]
        self._steps.copy_(sd["steps"])
        for dst, src in zip(self._exp_avg, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(self._exp_avg_sq, sd["exp_avg_sq"]):
            dst.copy_(src)
