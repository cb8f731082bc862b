"""Per-iteration full-volume operators of the coarse stages as HIP stencil kernels with autograd.

``smooth3d(grid, taps)``      = ``nn.Conv3d(1, 1, k, padding=k//2, padding_mode='replicate')(grid)`` with the frozen
                                Gaussian taps of model/nerf.py:260-272 (the coarse stages smooth the SDF grid on
                                every forward, :791 / :969).
``sdf_gradient_volume(g, vs)`` = ``nerf.neus_sdf_gradient(mode)`` (model/nerf.py:485-508): [1,3,X,Y,Z]; 'interpolate' and 'raw'
                                are one stencil kernel each way, 'grad_conv' three ``smooth3d`` passes with the reference's
                                Sobel-like taps (:224-247).
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
    def forward(ctx, grid, voxel_size, pack, mode=0):
        X, Y, Z = _check_grid(grid)
        g = grid.contiguous()
        out = torch.empty(1, 3, X, Y, Z, dtype=torch.float32, device=g.device)
        vol4 = None
        if pack is not None:       # (pack_sdf [1,1,X,Y,Z], holder): also write the interleaved [X,Y,Z,4] copy
            pack_sdf, holder = pack
            _check_grid(pack_sdf)
            vol4 = torch.empty(X, Y, Z, 4, dtype=torch.float32, device=g.device)
            holder['vol4'] = vol4
            pack_sdf = pack_sdf.detach().contiguous()
        call("fgs_sdf_gradvol_fwd", ptr(g), X, Y, Z, float(voxel_size), int(mode), ptr(out),
             ptr(pack_sdf) if pack is not None else None, ptr(vol4), stream())
        ctx.meta = (X, Y, Z, float(voxel_size), int(mode))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        X, Y, Z, vs, mode = ctx.meta
        sv = _voxel_stride(d_out, X, Y, Z)
        if sv is None:
            d_out, sv = d_out.contiguous(), 1
        sc = d_out.stride(1)
        d_in = torch.empty(1, 1, X, Y, Z, dtype=torch.float32, device=d_out.device)
        call("fgs_sdf_gradvol_bwd", ptr(d_out), sc, sv, X, Y, Z, vs, mode, ptr(d_in), 0, stream())
        return d_in, None, None, None


GRAD_MODES = {'interpolate': 0, 'raw': 1}


def sdf_gradient_volume(grid: torch.Tensor, voxel_size: float, pack_sdf=None, holder=None, mode: str = 'interpolate',
                        grad_conv_weight=None) -> torch.Tensor:
    """[1,3,X,Y,Z] gradient volume of model/nerf.py:485-508.  'interpolate' (central difference, zero faces) and 'raw' (forward
    difference, zero last face): one stencil pass; with `pack_sdf` (the smoothed SDF grid) and a dict `holder`, the same pass
    also leaves the voxel-interleaved volume {pack_sdf, g_x, g_y, g_z} in holder['vol4'] for the coarse march (a plain
    by-product: no gradient flows through it).  'grad_conv': `grad_conv_weight` [3,1,3,3,3] (nerf.init_gradient_conv), one
    replicate-padded 3^3 convolution per component (smooth3d: forward and exact adjoint); no interleaved copy."""
    if mode == 'grad_conv':
        if grad_conv_weight is None:
            raise RuntimeError("sdf_gradient_volume(mode='grad_conv') needs the grad_conv weight")
        w = grad_conv_weight.detach()
        return torch.cat([smooth3d(grid, w[c, 0]) for c in range(3)], dim=1)
    pack = None if pack_sdf is None else (pack_sdf, holder)
    return _GradVol.apply(grid, float(voxel_size), pack, GRAD_MODES[mode])


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


class _GridTV(torch.autograd.Function):
    """`total_variation(v, mask)` of the reference (model/nerf.py:1212-1221; `per_axis_mean=True`: model/dvgo.py:420-428) as
    one HIP value pass and one HIP gradient pass (csrc/tvloss.hip) instead of ~20 dense torch kernels that each save a
    grid-sized tensor.  Denominators and the backward scale factors are device scalars: nothing is read by the host."""

    @staticmethod
    def forward(ctx, v, mask_u8, masked_count, per_axis_mean):
        if not (v.is_cuda and v.dtype == torch.float32 and v.dim() == 5 and v.shape[0] == 1):
            raise RuntimeError("expected a float32 CUDA grid of shape [1,C,X,Y,Z]")
        from .ops import grid_strides
        dims = grid_strides(v)                                   # C, X, Y, Z, sC, sX, sY, sZ
        sums = torch.zeros(7, dtype=torch.float64, device=v.device)
        call("fgs_tv_loss_value", ptr(v), ptr(mask_u8), *dims, ptr(sums), stream())
        S, V, n_pairs = sums[0:3], sums[3], sums[4:7]
        if per_axis_mean:                                        # (mean_x + mean_y + mean_z) / 3 over the valid pairs
            axis_w = 1.0 / (3.0 * n_pairs)
            loss = (S * axis_w).sum()
            w0 = torch.zeros((), dtype=torch.float64, device=v.device)
        else:                                                    # (S_x + S_y + S_z) / 3 / (mask.sum() or v.sum())
            den = masked_count.to(torch.float64) if mask_u8 is not None else V
            axis_w = (1.0 / (3.0 * den)).expand(3)
            loss = S.sum() / (3.0 * den)
            # without a mask the denominator is v.sum(): d/dv also carries -S / (3 V^2)
            w0 = torch.zeros((), dtype=torch.float64, device=v.device) if mask_u8 is not None else -S.sum() / (3.0 * den * den)
        ctx.save_for_backward(v, torch.cat([axis_w.reshape(3), w0.reshape(1)]))
        ctx.mask_u8, ctx.dims = mask_u8, dims
        return loss.to(torch.float32)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss):
        v, w = ctx.saved_tensors
        w = (w * g_loss.to(torch.float64)).to(torch.float32).contiguous()
        grad = torch.empty_strided(v.shape, v.stride(), dtype=torch.float32, device=v.device)
        call("fgs_tv_loss_grad", ptr(v), ptr(ctx.mask_u8), *ctx.dims, ptr(w), ptr(grad), 0, stream())
        return grad, None, None, None


def grid_tv_loss(v: torch.Tensor, mask=None, per_axis_mean: bool = False) -> torch.Tensor:
    """The reference's `total_variation(v, mask)` on a CUDA grid.  `mask`: bool [1,1,X,Y,Z] or [1,C,X,Y,Z] with identical
    channels (the reference builds the latter with `.repeat(1, C, 1, 1, 1)`: model/nerf.py:454) or None."""
    mask_u8 = count = None
    if mask is not None:
        m = mask
        if m.dim() != 5 or tuple(m.shape[2:]) != tuple(v.shape[2:]) or m.shape[1] not in (1, v.shape[1]):
            raise RuntimeError(f"mask shape {tuple(mask.shape)} does not fit grid {tuple(v.shape)}")
        reps = m.shape[1]
        m0 = m[0, 0].contiguous()
        mask_u8 = m0.view(torch.uint8) if m0.dtype == torch.bool else (m0 != 0).view(torch.uint8)
        count = m0.sum() * reps                                  # mask.sum() of the tensor the caller passed (device scalar)
    return _GridTV.apply(v, mask_u8, count, bool(per_axis_mean))

