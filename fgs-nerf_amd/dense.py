"""Per-iteration full-volume operators of the coarse stages as HIP stencil kernels with autograd.

``smooth3d(grid, taps)``      = ``nn.Conv3d(1, 1, k, padding=k//2, padding_mode='replicate')(grid)`` with the frozen
                                Gaussian taps of model/nerf.py:260-272 (the coarse stages smooth the SDF grid on
                                every forward, :791 / :969).
``sdf_gradient_volume(g, vs)`` = ``nerf.neus_sdf_gradient(mode='interpolate')`` (model/nerf.py:485-494): [1,3,X,Y,Z].
"""
from __future__ import annotations

import ctypes

import torch

from ._lib import call, ptr, stream


def _taps_c(taps: torch.Tensor):
    t = taps.detach().float().cpu().contiguous().reshape(-1)
    return (ctypes.c_float * t.numel())(*t.tolist())


def _check_grid(g: torch.Tensor):
    if not (g.is_cuda and g.dtype == torch.float32 and g.dim() == 5 and g.shape[0] == 1 and g.shape[1] == 1):
        raise RuntimeError("expected a float32 CUDA grid of shape [1,1,X,Y,Z]")
    return int(g.shape[2]), int(g.shape[3]), int(g.shape[4])


def _voxel_stride(t: torch.Tensor, X: int, Y: int, Z: int):
    """e if the spatial strides of a [1,C,X,Y,Z] tensor are (Y*Z*e, Z*e, e) -- dense (e = 1) or a channel slice of a
    voxel-interleaved buffer (e = 4) -- else None."""
    if t.dtype != torch.float32 or t.dim() != 5:
        return None
    e = t.stride(4)
    if e >= 1 and t.stride(3) == Z * e and t.stride(2) == Y * Z * e:
        return int(e)
    return None


class _Smooth3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, taps_c, k):
        X, Y, Z = _check_grid(grid)
        g = grid.contiguous()
        out = torch.empty_like(g)
        call("fgs_smooth3d_fwd", ptr(g), X, Y, Z, k, taps_c, ptr(out), stream())
        ctx.meta = (X, Y, Z, k, taps_c)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        X, Y, Z, k, taps_c = ctx.meta
        es = _voxel_stride(d_out, X, Y, Z)         # e.g. channel 0 of the interleaved [X,Y,Z,4] buffer: no copy
        if es is None:
            d_out, es = d_out.contiguous(), 1
        d_in = torch.empty(1, 1, X, Y, Z, dtype=torch.float32, device=d_out.device)
        scratch = torch.empty((X + k - 1) * (Y + k - 1) * (Z + k - 1), dtype=torch.float32, device=d_out.device)
        call("fgs_smooth3d_bwd", ptr(d_out), es, X, Y, Z, k, taps_c, ptr(scratch), ptr(d_in), stream())
        return d_in, None, None


def smooth3d(grid: torch.Tensor, taps: torch.Tensor, taps_c=None) -> torch.Tensor:
    """taps: [k,k,k] (or Conv3d weight [1,1,k,k,k]) normalised Gaussian weights."""
    k = int(taps.shape[-1])
    return _Smooth3d.apply(grid, taps_c if taps_c is not None else _taps_c(taps), k)


class _GradVol(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, voxel_size):
        X, Y, Z = _check_grid(grid)
        g = grid.contiguous()
        out = torch.empty(1, 3, X, Y, Z, dtype=torch.float32, device=g.device)
        call("fgs_sdf_gradvol_fwd", ptr(g), X, Y, Z, float(voxel_size), ptr(out), stream())
        ctx.meta = (X, Y, Z, float(voxel_size))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        X, Y, Z, vs = ctx.meta
        sv = _voxel_stride(d_out, X, Y, Z)
        if sv is None:
            d_out, sv = d_out.contiguous(), 1
        sc = d_out.stride(1)
        d_in = torch.empty(1, 1, X, Y, Z, dtype=torch.float32, device=d_out.device)
        call("fgs_sdf_gradvol_bwd", ptr(d_out), sc, sv, X, Y, Z, vs, ptr(d_in), 0, stream())
        return d_in, None


def sdf_gradient_volume(grid: torch.Tensor, voxel_size: float) -> torch.Tensor:
    return _GradVol.apply(grid, float(voxel_size))


class _SmoothTV(torch.autograd.Function):
    """weight * mean_masked((tv_smooth_conv(g).detach() - g)^2) over a [1,3,X,Y,Z] gradient volume: value and d/dg in one
    HIP pass per channel (include/fgs_hip.h fgs_smooth_tv_loss)."""

    @staticmethod
    def forward(ctx, grad3, taps_c, mask_u8, inv_count, weight):
        if not (grad3.is_cuda and grad3.dtype == torch.float32 and grad3.dim() == 5 and grad3.shape[:2] == (1, 3)):
            raise RuntimeError("expected a float32 CUDA gradient volume of shape [1,3,X,Y,Z]")
        g = grad3.contiguous()
        X, Y, Z = (int(v) for v in g.shape[2:])
        loss = torch.zeros((), dtype=torch.float32, device=g.device)
        d_g = torch.empty_like(g)
        call("fgs_smooth_tv_loss", ptr(g), X, Y, Z, taps_c, ptr(mask_u8), ptr(inv_count), float(weight), ptr(loss), ptr(d_g),
             stream())
        ctx.save_for_backward(d_g)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss):
        (d_g,) = ctx.saved_tensors
        return d_g * g_loss, None, None, None, None


def smooth_tv_loss(grad3: torch.Tensor, taps_c, mask_u8, inv_count: torch.Tensor, weight: float) -> torch.Tensor:
    return _SmoothTV.apply(grad3, taps_c, mask_u8, inv_count, float(weight))
