"""``MaskedAdam`` behind the reference's ``model/adam.py`` surface (model/adam.py:167-221).

Contract kept: constructor ``MaskedAdam(params, lr, betas=(0.9, 0.99), eps)``, param-group keys
``lr / betas / eps / skip_zero_grad``, per-parameter state ``step / exp_avg / exp_avg_sq``,
``set_pervoxel_lr(count)``, and the rule that picks one of three fused update kernels per tensor
(per-voxel lr if a same-shape table is set, else masked if the group says ``skip_zero_grad``, else
dense).  The kernels are csrc/gridopt.hip (``fgs_adam_upd``), not the JIT-built ``adam_upd_cuda``.
The reference's unused ``Adam`` class (model/adam.py:16-161) is outside the hot path.
"""
from __future__ import annotations

import torch

from .ops import adam_upd_cuda


def _as_layout_of(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """`t` re-laid out with `like`'s strides (grids are stored channel-last; autograd may hand back another layout)."""
    if t.stride() == like.stride():
        return t
    return torch.empty_strided(like.shape, like.stride(), dtype=t.dtype, device=t.device).copy_(t)


class MaskedAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8):
        for ok, what in ((lr >= 0.0, f"Invalid learning rate: {lr}"),
                         (eps >= 0.0, f"Invalid epsilon value: {eps}"),
                         (0.0 <= betas[0] < 1.0, f"Invalid beta parameter at index 0: {betas[0]}"),
                         (0.0 <= betas[1] < 1.0, f"Invalid beta parameter at index 1: {betas[1]}")):
            if not ok:
                raise ValueError(what)
        self.per_lr = None
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def set_pervoxel_lr(self, count):
        self.per_lr = count.float() / count.max()

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            b1, b2 = group['betas']
            hyper = (b1, b2, group['lr'], group['eps'])
            masked = group['skip_zero_grad']
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self._state_of(p)
                st['step'] += 1
                g = _as_layout_of(p.grad, p)
                m, v, t = st['exp_avg'], st['exp_avg_sq'], st['step']
                if self.per_lr is not None and p.shape == self.per_lr.shape:
                    adam_upd_cuda.adam_upd_with_perlr(p, g, m, v, _as_layout_of(self.per_lr, p), t, *hyper)
                elif masked:
                    adam_upd_cuda.masked_adam_upd(p, g, m, v, t, *hyper)
                else:
                    adam_upd_cuda.adam_upd(p, g, m, v, t, *hyper)
