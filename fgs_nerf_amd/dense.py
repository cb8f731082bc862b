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

from ._lib import call, lib, ptr, stream


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
    def forward(ctx, grid, voxel_size, pack, mode=0, side=None):
        # side: a dict shared with the result's consumers -- a term that would hand autograd a SECOND gradient of the volume
        # (the smooth-gradient TV term) leaves it there as side['extra'] and the backward pass below adds it on the fly
        ctx.side = side
        ctx.set_materialize_grads(False)
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
        extra = ctx.side.pop('extra', None) if ctx.side is not None else None
        if d_out is None:
            if extra is None:
                return None, None, None, None, None
            d_out, extra = extra, None
        sv = _voxel_stride(d_out, X, Y, Z)
        if sv is None:
            d_out, sv = d_out.contiguous(), 1
        sc = d_out.stride(1)
        d_in = torch.empty(1, 1, X, Y, Z, dtype=torch.float32, device=d_out.device)
        call("fgs_sdf_gradvol_bwd", ptr(d_out), sc, sv, X, Y, Z, vs, mode, ptr(d_in), 0, ptr(extra), stream())
        return d_in, None, None, None, None


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
    side = {}
    out = _GradVol.apply(grid, float(voxel_size), pack, GRAD_MODES[mode], side)
    out._fgs_side = side             # (read by smooth_tv_loss: see _GradVol.forward)
    return out


_TV_SCRATCH = {}          # (device index, kind) -> scratch of a TV value launch (first word: its arrival counter, left zero)
_TV_SCRATCH_RETIRED = []  # outgrown scratch buffers (a few KB each), kept alive: see _tv_scratch


def _tv_scratch(dev, kind: str, n: int, dtype) -> torch.Tensor:
    key = (dev.index, kind)
    t = _TV_SCRATCH.get(key)
    if t is None or t.numel() < n:
        if t is not None:
            _TV_SCRATCH_RETIRED.append(t)       # (a captured step may hold its address: never handed back to the allocator)
        t = _TV_SCRATCH[key] = torch.zeros(n, dtype=dtype, device=dev)
    return t


def _is_unit_seed(g_loss: torch.Tensor) -> bool:
    from .losses import UNIT_SEEDS
    return g_loss.data_ptr() in UNIT_SEEDS


class _SmoothTV(torch.autograd.Function):
    """weight * mean_masked((tv_smooth_conv(g).detach() - g)^2) (+ add_in) over a [1,3,X,Y,Z] gradient volume: value and d/dg in
    one HIP pass per channel (include/fgs_hip.h fgs_smooth_tv_loss); `add_in`: the loss so far (saves an addition launch)."""

    @staticmethod
    def forward(ctx, grad3, taps_c, mask_u8, inv_count, weight, add_in, side):
        if not (grad3.is_cuda and grad3.dtype == torch.float32 and grad3.dim() == 5 and grad3.shape[:2] == (1, 3)):
            raise RuntimeError("expected a float32 CUDA gradient volume of shape [1,3,X,Y,Z]")
        g = grad3.contiguous()
        X, Y, Z = (int(v) for v in g.shape[2:])
        loss = torch.empty((), dtype=torch.float32, device=g.device)
        d_g = torch.empty_like(g)
        need = int(lib().fgs_smooth_tv_scratch_floats(X, Y, Z))
        scratch = _tv_scratch(g.device, 'smooth', need, torch.float32)
        call("fgs_smooth_tv_loss", ptr(g), X, Y, Z, taps_c, ptr(mask_u8), ptr(inv_count), float(weight), ptr(add_in), ptr(scratch),
             scratch.numel(), ptr(loss), ptr(d_g), stream())
        ctx.save_for_backward(d_g)
        ctx.has_add, ctx.side = add_in is not None, side
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss):
        (d_g,) = ctx.saved_tensors
        # (a captured step passes a registered unit seed through the additions in front of this node: no grid-sized multiply)
        g = d_g if _is_unit_seed(g_loss) else d_g * g_loss
        if ctx.side is not None and 'extra' not in ctx.side:
            # the volume came from _GradVol: its backward pass (which autograd runs after this one) adds this gradient to the
            # march kernels' while it reads them -- handing it to autograd would cost a grid-sized addition with a strided operand
            ctx.side['extra'] = g
            g = None
        return g, None, None, None, None, (g_loss if ctx.has_add else None), None


def smooth_tv_loss(grad3: torch.Tensor, taps_c, mask_u8, inv_count: torch.Tensor, weight: float, add_in=None) -> torch.Tensor:
    side = getattr(grad3, '_fgs_side', None) if grad3.is_contiguous() else None
    return _SmoothTV.apply(grad3, taps_c, mask_u8, inv_count, float(weight), add_in, side)


class _GridTV(torch.autograd.Function):
    """scale * `total_variation(v, mask)` (+ add_in) of the reference (model/nerf.py:1212-1221; `per_axis_mean=True`:
    model/dvgo.py:420-428) as one HIP value pass and one HIP gradient pass (csrc/tvloss.hip) instead of ~20 dense torch kernels that
    each save a grid-sized tensor.  The scalar algebra between the seven sums, the loss and the backward factors runs in the value
    launch's last workgroup (double), the upstream gradient is read by the gradient launch: two launches in all."""

    @staticmethod
    def forward(ctx, v, mask_u8, masked_count, per_axis_mean, scale, add_in):
        if not (v.is_cuda and v.dtype == torch.float32 and v.dim() == 5 and v.shape[0] == 1):
            raise RuntimeError("expected a float32 CUDA grid of shape [1,C,X,Y,Z]")
        from .ops import grid_strides
        dims = grid_strides(v)                                   # C, X, Y, Z, sC, sX, sY, sZ
        loss = torch.empty((), dtype=torch.float32, device=v.device)
        w = torch.empty(4, dtype=torch.float32, device=v.device)
        scratch = _tv_scratch(v.device, 'grid', int(lib().fgs_tv_loss_scratch_doubles()), torch.float64)
        count = masked_count if (mask_u8 is not None and not per_axis_mean) else None
        call("fgs_tv_loss_value", ptr(v), ptr(mask_u8), *dims, ptr(count), int(bool(per_axis_mean)), float(scale), ptr(add_in),
             ptr(scratch), scratch.numel(), ptr(loss), ptr(w), None, stream())
        ctx.save_for_backward(v, w)
        ctx.mask_u8, ctx.dims, ctx.has_add = mask_u8, dims, add_in is not None
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss):
        v, w = ctx.saved_tensors
        up = None if _is_unit_seed(g_loss) else g_loss.to(torch.float32).contiguous()
        grad = torch.empty_strided(v.shape, v.stride(), dtype=torch.float32, device=v.device)
        call("fgs_tv_loss_grad", ptr(v), ptr(ctx.mask_u8), *ctx.dims, ptr(w), ptr(up), ptr(grad), 0, stream())
        return grad, None, None, None, None, (g_loss if ctx.has_add else None)


_MASK_COUNTS = []         # [(mask tensor, its _version, mask_u8 view, mask.sum() as a device int64)]: the count is two launches


def _mask_count(mask: torch.Tensor):
    for m, ver, u8, cnt in _MASK_COUNTS:
        if m is mask and ver == mask._version:
            return u8, cnt
    m0 = mask[0, 0].contiguous()
    u8 = m0.view(torch.uint8) if m0.dtype == torch.bool else (m0 != 0).view(torch.uint8)
    cnt = (m0.sum() * mask.shape[1]).to(torch.int64).reshape(1)          # mask.sum() of the tensor the caller passed (device scalar)
    _MASK_COUNTS.append((mask, mask._version, u8, cnt))
    del _MASK_COUNTS[:-4]
    return u8, cnt


def grid_tv_loss(v: torch.Tensor, mask=None, per_axis_mean: bool = False, scale: float = 1.0, add_in=None) -> torch.Tensor:
    """scale * the reference's `total_variation(v, mask)` (+ add_in) on a CUDA grid.  `mask`: bool [1,1,X,Y,Z] or [1,C,X,Y,Z] with
    identical channels (the reference builds the latter with `.repeat(1, C, 1, 1, 1)`: model/nerf.py:454) or None."""
    mask_u8 = count = None
    if mask is not None:
        if mask.dim() != 5 or tuple(mask.shape[2:]) != tuple(v.shape[2:]) or mask.shape[1] not in (1, v.shape[1]):
            raise RuntimeError(f"mask shape {tuple(mask.shape)} does not fit grid {tuple(v.shape)}")
        mask_u8, count = _mask_count(mask)
    return _GridTV.apply(v, mask_u8, count, bool(per_axis_mean), float(scale), add_in)

