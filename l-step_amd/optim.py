"""Single-kernel Adam for the L-STEP parameter set.

The reference optimises with plain ``torch.optim.Adam`` (``utils/utils.py:49-67``).  PyTorch's fused (one multi-tensor
kernel) implementation refuses complex parameters, and L-STEP has exactly one (``fft_filter.weight``, complex64).  Adam treats
a complex tensor as its real view, so this wrapper hands the optimiser ``view_as_real`` of that parameter (same storage) and
mirrors its gradient before each step: the same update rule, 1 launch instead of ~14 per step.
"""
from __future__ import annotations

import torch


class FusedAdam:
    """``torch.optim.Adam(..., fused=True)`` over the real parameters plus the real views of the complex ones."""

    def __init__(self, params, lr: float = 1e-4, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8):
        self._params = [p for p in params if p.requires_grad]
        self._complex = [(p, torch.view_as_real(p.data)) for p in self._params if p.is_complex()]
        real = [p for p in self._params if not p.is_complex()] + [v for _, v in self._complex]
        for _, v in self._complex:
            v.requires_grad_(False)
        self.optimizer = torch.optim.Adam(real, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps, fused=True)

    def zero_grad(self, set_to_none: bool = True):
        for p in self._params:
            p.grad = None
        for _, v in self._complex:
            v.grad = None

    def step(self):
        for p, v in self._complex:
            v.grad = None if p.grad is None else torch.view_as_real(p.grad)
        self.optimizer.step()

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd)
