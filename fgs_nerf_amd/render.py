"""Differentiable building blocks of the render path, one per reference operator chain.

These are the stride-generic, operator-at-a-time forms (each a HIP kernel pair behind
``torch.autograd.Function``).  They serve arbitrary model configurations and are what the
reference-compatible modules (`grid.DenseGrid`, `nerf.nerf`, `dvgo.dvgo`) fall back on when a
configuration is outside the fused kernels of ``fused.py``.
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import ops
from .grid import trilerp


class Alphas2Weights(torch.autograd.Function):
    """model/nerf.py:1173-1189 / model/dvgo.py:390-406: ``apply(alpha, ray_id, N) -> (weights, alphainv_last)``."""

    @staticmethod
    def forward(ctx, alpha, ray_id, N):
        alpha = alpha.contiguous()
        weights, T, alphainv_last, i_start, i_end = ops.render_utils_cuda.alpha2weight(alpha, ray_id.contiguous(), N)
        if alpha.requires_grad:
            ctx.save_for_backward(alpha, weights, T, alphainv_last, i_start, i_end)
            ctx.n_rays = N
        return weights, alphainv_last

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_weights, grad_last):
        alpha, weights, T, alphainv_last, i_start, i_end = ctx.saved_tensors
        grad = ops.render_utils_cuda.alpha2weight_backward(
            alpha, weights, T, alphainv_last, i_start, i_end, ctx.n_rays,
            grad_weights.contiguous(), grad_last.contiguous())
        return grad, None, None


class _SdfTaps(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, pts, xyz_min, xyz_max, displace):
        feat, diff = ops.sdf_taps_fwd(grid, pts, xyz_min, xyz_max, displace)
        ctx.save_for_backward(pts, xyz_min, xyz_max)
        ctx.displace = tuple(displace)
        ctx.grid_shape = grid.shape
        ctx.mark_non_differentiable(diff)
        return feat, diff

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_feat, _grad_diff):
        pts, xyz_min, xyz_max = ctx.saved_tensors
        grad_grid = torch.zeros(ctx.grid_shape, dtype=torch.float32, device=grad_feat.device)
        ops.sdf_taps_bwd(grad_grid, pts, xyz_min, xyz_max, ctx.displace, grad_feat.contiguous())
        return grad_grid, None, None, None, None


def sample_sdfs(xyz, grid, xyz_min, xyz_max, voxel_size, displace_list: Sequence[float], use_grad_norm=False):
    """nerf.sample_sdfs (model/nerf.py:597-637): feat [M,6K], grad [M,3K] (zyx-major, displacement-minor)."""
    M = xyz.shape[:-1].numel()
    K = len(displace_list)
    feat, diff = _SdfTaps.apply(grid.contiguous(), xyz.reshape(-1, 3).contiguous(), xyz_min, xyz_max, tuple(displace_list))
    f = feat.view(M, 6, K)
    grad = (f[:, 1::2] - f[:, 0::2]) / diff.view(M, 3, K) / voxel_size
    if use_grad_norm:
        grad = grad / (grad.norm(dim=1, keepdim=True) + 1e-5)
    return feat, grad.reshape(M, 3 * K)


def grid_sampler(xyz, grid, xyz_min, xyz_max, voxel_size=None, sample_ret=True, sample_grad=False):
    """nerf.grid_sampler (model/nerf.py:639-672) for mode='bilinear', align_corners=True.
    sample_ret -> value(s) [.., C] squeezed; sample_grad -> + xyz-ordered gradient [M,3] and taps [M,6]."""
    shape = xyz.shape[:-1]
    pts = xyz.reshape(-1, 3)
    outs = []
    if sample_ret:
        ret = trilerp(grid, pts, xyz_min, xyz_max).reshape(*shape, grid.shape[1]).squeeze(-1)
        outs.append(ret)
    if sample_grad:
        feat, grad = sample_sdfs(pts, grid, xyz_min, xyz_max, voxel_size, [1.0], use_grad_norm=False)
        outs.append(torch.cat([grad[:, [2]], grad[:, [1]], grad[:, [0]]], dim=-1))
        outs.append(torch.cat([feat[:, 4:6], feat[:, 2:4], feat[:, 0:2]], dim=-1))
    return outs[0] if len(outs) == 1 else outs


def segment_sum(src: torch.Tensor, index: torch.Tensor, n: int) -> torch.Tensor:
    """torch_scatter.segment_coo(src, index, out=zeros([n,..]), reduce='sum') for sorted index
    (model/nerf.py:888-896)."""
    out = torch.zeros([n, *src.shape[1:]], dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def posenc(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """[x, sin(x f_i), cos(x f_i)], component-major / frequency-minor (model/nerf.py:838-839)."""
    emb = (x.unsqueeze(-1) * freqs).flatten(-2)
    return torch.cat([x, emb.sin(), emb.cos()], -1)


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """model/nerf.py:480-483."""
    eps = torch.tensor(torch.finfo(torch.float32).eps, device=x.device)
    return x / torch.sqrt(torch.maximum(torch.sum(x ** 2, dim=-1, keepdims=True), eps))
