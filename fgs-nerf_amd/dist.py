"""Data-parallel ray sharding: one process per GPU, grids and MLPs replicated, one gradient sum per step.

The reference has no distributed code; this is the north_star's
multi-GPU extension.  Rank r renders rays [r*N/P, (r+1)*N/P) of the global batch with a loss that is a
mean over its local rays; averaging the gradients over ranks (sum all-reduce, then 1/P) gives exactly
the gradient of the global-batch mean loss.  TV and MaskedAdam then run identically on every rank, so
the replicas stay bit-identical without a parameter broadcast.

Collectives go through torch.distributed: backend "nccl" is RCCL over xGMI on ROCm, "gloo" is used by
the CPU tests.  The dense grid gradients are reduced in place as their own messages (they are single
contiguous tensors of 16 MB .. 1.7 GB: already at the large-message plateau of a direct
reduce-scatter + all-gather); the few hundred small MLP gradients are packed into one bucket.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def shard_rays(n_total: int, rank: int, world_size: int) -> slice:
    """Contiguous ray shard of rank `rank` (remainder rays go to the lowest ranks)."""
    base, rem = divmod(n_total, world_size)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


class GradAverager:
    """Averages `.grad` of the given parameters over the process group.

    Large tensors (numel >= big_numel) are reduced one message each, asynchronously; the rest are
    flattened into a single bucket.  ``masked_adam_upd`` keys on grad != 0: a sum keeps every voxel that
    any rank touched non-zero, so the masked update touches the union, as a single-GPU run on the
    concatenated batch would."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None,
                 big_numel: int = 1 << 20):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.big_numel = big_numel
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self._bucket = None

    @torch.no_grad()
    def average(self) -> None:
        if self.world_size == 1:
            return
        inv = 1.0 / self.world_size
        handles = []
        small = []
        for p in self.params:
            if p.grad is None:
                continue
            if p.grad.numel() >= self.big_numel:
                g = p.grad
                if not (g.is_contiguous() or g.is_contiguous(memory_format=torch.channels_last_3d)):
                    g = g.contiguous()
                    p.grad = g
                # reduce the dense storage as a flat view (layout-agnostic, no copy)
                flat = g.as_strided((g.numel(),), (1,))
                handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat))
            else:
                small.append(p.grad)
        if small:
            n = sum(g.numel() for g in small)
            if self._bucket is None or self._bucket.numel() != n or self._bucket.device != small[0].device:
                self._bucket = torch.empty(n, dtype=small[0].dtype, device=small[0].device)
            off = 0
            for g in small:
                self._bucket[off:off + g.numel()].copy_(g.reshape(-1))
                off += g.numel()
            dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM, group=self.group)
            self._bucket.mul_(inv)
            off = 0
            for g in small:
                g.copy_(self._bucket[off:off + g.numel()].view_as(g))
                off += g.numel()
        for h, flat in handles:
            h.wait()
            flat.mul_(inv)
