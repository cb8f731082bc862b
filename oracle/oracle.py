"""CPU oracle for the FGS-NeRF voxel render/training hot path.

TEST INFRASTRUCTURE ONLY -- the product (fgs-nerf_amd/) never imports this module.
Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.

Two layers, both restating the reference (citations are paths under /root/reference):

* ``K``  -- ctypes view of oracle/liboracle.so, the C restatement of model/cuda/*.cu
  (see fgs_oracle.c; "parity unpinned": the reference holds no fixtures for it and its
  CUDA sources cannot be built in this image).
* torch-CPU functions restating the Python side of the path (model/grid.py,
  model/nerf.py, model/dvgo.py) with the same torch calls, in the same order, that the
  reference makes -- F.grid_sample, nn.functional.linear, sigmoid, index_add_ (the
  documented equivalent of torch_scatter.segment_coo(reduce='sum') for a sorted index;
  torch_scatter itself is not installed here).  Autograd through these gives the
  reference gradients.

Pinned parts: ray generation (tests/golden/rays_*.npz were produced by importing the
reference's own model/dvgo_ray.py, see oracle/make_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "fgs_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class _Kernels:
    """numpy front-end to the C restatement of model/cuda/*.cu."""

    def __init__(self):
        self._lib = None

    @property
    def lib(self):
        if self._lib is None:
            self._lib = ctypes.CDLL(build())
            self._lib.orc_sample_count.restype = ctypes.c_int64
            self._lib.orc_adam_step_size.restype = ctypes.c_float
        return self._lib

    # --- render_utils.cpp:170-184 surface -------------------------------------------
    def infer_t_minmax(self, rays_o, rays_d, xyz_min, xyz_max, near, far):
        rays_o, rays_d, xyz_min, xyz_max = map(_f32, (rays_o, rays_d, xyz_min, xyz_max))
        n = rays_o.shape[0]
        t_min, t_max = np.empty(n, np.float32), np.empty(n, np.float32)
        self.lib.orc_infer_t_minmax(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max),
                                    ctypes.c_float(near), ctypes.c_float(far), ctypes.c_int64(n),
                                    _p(t_min), _p(t_max))
        return t_min, t_max

    def infer_n_samples(self, rays_d, t_min, t_max, stepdist):
        rays_d, t_min, t_max = map(_f32, (rays_d, t_min, t_max))
        n = t_min.shape[0]
        out = np.empty(n, np.int64)
        self.lib.orc_infer_n_samples(_p(rays_d), _p(t_min), _p(t_max), ctypes.c_float(stepdist),
                                     ctypes.c_int64(n), _p(out))
        return out

    def infer_ray_start_dir(self, rays_o, rays_d, t_min):
        rays_o, rays_d, t_min = map(_f32, (rays_o, rays_d, t_min))
        n = rays_o.shape[0]
        s, d = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        self.lib.orc_infer_ray_start_dir(_p(rays_o), _p(rays_d), _p(t_min), ctypes.c_int64(n), _p(s), _p(d))
        return s, d

    def sample_pts_on_rays(self, rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
        rays_o, rays_d, xyz_min, xyz_max = map(_f32, (rays_o, rays_d, xyz_min, xyz_max))
        n = rays_o.shape[0]
        n_steps = np.empty(n, np.int64)
        t_min, t_max = np.empty(n, np.float32), np.empty(n, np.float32)
        tot = self.lib.orc_sample_count(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max),
                                        ctypes.c_float(near), ctypes.c_float(far), ctypes.c_float(stepdist),
                                        ctypes.c_int64(n), _p(n_steps), _p(t_min), _p(t_max))
        pts = np.empty((tot, 3), np.float32)
        mask = np.empty(tot, np.uint8)
        ray_id, step_id = np.empty(tot, np.int64), np.empty(tot, np.int64)
        self.lib.orc_sample_emit(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), ctypes.c_float(stepdist),
                                 ctypes.c_int64(n), _p(n_steps), _p(t_min), _p(pts), _p(mask), _p(ray_id), _p(step_id))
        return pts, mask.astype(bool), ray_id, step_id, n_steps, t_min, t_max

    def sample_ndc_pts_on_rays(self, rays_o, rays_d, xyz_min, xyz_max, n_samples):
        rays_o, rays_d, xyz_min, xyz_max = map(_f32, (rays_o, rays_d, xyz_min, xyz_max))
        n = rays_o.shape[0]
        pts = np.empty((n, n_samples, 3), np.float32)
        mask = np.empty((n, n_samples), np.uint8)
        self.lib.orc_sample_ndc_pts(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max),
                                    ctypes.c_int64(n_samples), ctypes.c_int64(n), _p(pts), _p(mask))
        return pts, mask.astype(bool)

    def sample_bg_pts_on_rays(self, rays_o, rays_d, t_max, bg_preserve, n_samples):
        rays_o, rays_d, t_max = map(_f32, (rays_o, rays_d, t_max))
        n = rays_o.shape[0]
        pts = np.empty((n, n_samples, 3), np.float32)
        self.lib.orc_sample_bg_pts(_p(rays_o), _p(rays_d), _p(t_max), ctypes.c_float(bg_preserve),
                                   ctypes.c_int64(n_samples), ctypes.c_int64(n), _p(pts))
        return pts

    def maskcache_lookup(self, world, xyz, scale, shift):
        world = np.ascontiguousarray(np.asarray(world, dtype=np.uint8))
        xyz, scale, shift = map(_f32, (xyz, scale, shift))
        n = xyz.shape[0]
        out = np.zeros(n, np.uint8)
        self.lib.orc_maskcache_lookup(_p(world), _p(xyz), _p(scale), _p(shift),
                                      ctypes.c_int(world.shape[0]), ctypes.c_int(world.shape[1]),
                                      ctypes.c_int(world.shape[2]), ctypes.c_int64(n), _p(out))
        return out.astype(bool)

    def raw2alpha(self, density, shift, interval):
        density = _f32(density)
        nonuni = None if np.isscalar(interval) else _f32(interval)
        e, a = np.empty_like(density), np.empty_like(density)
        self.lib.orc_raw2alpha(_p(density), ctypes.c_float(shift),
                               ctypes.c_float(0.0 if nonuni is not None else interval), _p(nonuni),
                               ctypes.c_int64(density.size), _p(e), _p(a))
        return e, a

    def raw2alpha_backward(self, exp_d, grad_back, interval):
        exp_d, grad_back = _f32(exp_d), _f32(grad_back)
        nonuni = None if np.isscalar(interval) else _f32(interval)
        g = np.empty_like(exp_d)
        self.lib.orc_raw2alpha_bwd(_p(exp_d), _p(grad_back),
                                   ctypes.c_float(0.0 if nonuni is not None else interval), _p(nonuni),
                                   ctypes.c_int64(exp_d.size), _p(g))
        return g

    def alpha2weight(self, alpha, ray_id, n_rays):
        alpha, ray_id = _f32(alpha), _i64(ray_id)
        m = alpha.shape[0]
        w, T = np.empty(m, np.float32), np.empty(m, np.float32)
        last = np.empty(n_rays, np.float32)
        i_s, i_e = np.empty(n_rays, np.int64), np.empty(n_rays, np.int64)
        self.lib.orc_alpha2weight_fwd(_p(alpha), _p(ray_id), ctypes.c_int64(m), ctypes.c_int64(n_rays),
                                      _p(w), _p(T), _p(last), _p(i_s), _p(i_e))
        return w, T, last, i_s, i_e

    def alpha2weight_backward(self, alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last):
        alpha, weight, T, alphainv_last, grad_weights, grad_last = map(
            _f32, (alpha, weight, T, alphainv_last, grad_weights, grad_last))
        i_start, i_end = _i64(i_start), _i64(i_end)
        g = np.empty_like(alpha)
        self.lib.orc_alpha2weight_bwd(_p(alpha), _p(weight), _p(T), _p(alphainv_last), _p(i_start), _p(i_end),
                                      ctypes.c_int64(alpha.shape[0]), ctypes.c_int64(n_rays),
                                      _p(grad_weights), _p(grad_last), _p(g))
        return g

    # --- total_variation.cpp:29-32 ---------------------------------------------------
    def total_variation_add_grad(self, param, grad, wx, wy, wz, dense_mode, mask=None):
        """In place on ``grad`` (float32 ndarray [1,C,X,Y,Z], C-contiguous)."""
        param = _f32(param)
        assert grad.dtype == np.float32 and grad.flags.c_contiguous and grad.shape == param.shape
        mk = None if mask is None else _f32(mask)
        self.lib.orc_tv_add_grad(_p(param), _p(grad), _p(mk), ctypes.c_float(wx), ctypes.c_float(wy),
                                 ctypes.c_float(wz), ctypes.c_int(bool(dense_mode)),
                                 ctypes.c_int64(param.shape[2]), ctypes.c_int64(param.shape[3]),
                                 ctypes.c_int64(param.shape[4]), ctypes.c_int64(param.size))

    # --- adam_upd.cpp:79-86 ----------------------------------------------------------
    def adam_upd(self, param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps, mode=0, perlr=None):
        """In place on param / exp_avg / exp_avg_sq (float32, C-contiguous).  mode: 0 adam_upd,
        1 masked_adam_upd, 2 adam_upd_with_perlr."""
        for a in (param, exp_avg, exp_avg_sq):
            assert a.dtype == np.float32 and a.flags.c_contiguous
        grad = _f32(grad)
        pl = None if perlr is None else _f32(perlr)
        self.lib.orc_adam_upd(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(pl), ctypes.c_int64(param.size),
                              ctypes.c_int(step), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(lr),
                              ctypes.c_float(eps), ctypes.c_int(mode))

    def adam_step_size(self, step, beta1, beta2, lr):
        return float(self.lib.orc_adam_step_size(ctypes.c_int(step), ctypes.c_float(beta1),
                                                 ctypes.c_float(beta2), ctypes.c_float(lr)))


K = _Kernels()


# =====================================================================================
# torch-CPU restatement of the Python side
# =====================================================================================

def grid_resolution(xyz_min: torch.Tensor, xyz_max: torch.Tensor, num_voxels: int):
    """model/nerf.py:298-303 `_set_grid_resolution` (fp32 tensor arithmetic, `.long()` truncation)."""
    voxel_size = ((xyz_max - xyz_min).prod() / num_voxels).pow(1 / 3)
    world_size = ((xyz_max - xyz_min) / voxel_size).long()
    return voxel_size, world_size


def ball_sdf(world_size: Sequence[int], radius: float = 1.0) -> torch.Tensor:
    """model/nerf.py:77-82: sdf = |p| - r on the np.mgrid[-1:1:G*1j]^3 lattice -> [1,1,X,Y,Z] float32."""
    x, y, z = np.mgrid[-1.0:1.0:world_size[0] * 1j, -1.0:1.0:world_size[1] * 1j, -1.0:1.0:world_size[2] * 1j]
    return torch.from_numpy((x ** 2 + y ** 2 + z ** 2) ** 0.5 - radius).float()[None, None, ...]


def ind_norm_of(xyz: torch.Tensor, xyz_min: torch.Tensor, xyz_max: torch.Tensor) -> torch.Tensor:
    """model/grid.py:55 / model/nerf.py:604,654: world xyz -> grid_sample coords (zyx order, [-1,1])."""
    return ((xyz - xyz_min) / (xyz_max - xyz_min)).flip((-1,)) * 2 - 1


def dense_grid_forward(grid: torch.Tensor, xyz: torch.Tensor, xyz_min, xyz_max) -> torch.Tensor:
    """model/grid.py:49-59 DenseGrid.forward (importance=None branch)."""
    channels = grid.shape[1]
    shape = xyz.shape[:-1]
    pts = xyz.reshape(1, 1, 1, -1, 3)
    out = F.grid_sample(grid, ind_norm_of(pts, xyz_min, xyz_max), mode='bilinear', align_corners=True)
    out = out.reshape(channels, -1).T.reshape(*shape, channels)
    if channels == 1:
        out = out.squeeze(-1)
    return out


def sample_sdfs(xyz, grid, xyz_min, xyz_max, voxel_size, displace_list, use_grad_norm=False):
    """model/nerf.py:597-637: 6*K axis taps (index space, clamped) and their finite differences.

    Returns feat [M, 6K] (layout ((axis_zyx*2 + sign) * K + k)) and grad [M, 3K] (axis_zyx * K + k)."""
    M = xyz.shape[:-1].numel()
    pts = xyz.reshape(1, 1, 1, -1, 3)
    gs = grid.shape[-3:]
    size_zyx = torch.tensor([gs[2], gs[1], gs[0]])
    ind = ((ind_norm_of(pts, xyz_min, xyz_max) + 1) / 2) * (size_zyx - 1)
    offset = torch.tensor([[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1]])
    displace = torch.tensor(list(displace_list))
    offset = offset[:, None, :] * displace[None, :, None]
    all_ind = (ind.unsqueeze(-2) + offset.view(-1, 3)).view(1, 1, 1, -1, 3)
    all_ind = torch.stack([all_ind[..., c].clamp(min=0, max=size_zyx[c] - 1) for c in range(3)], -1)
    all_ind_norm = (all_ind / (size_zyx - 1)) * 2 - 1
    feat = F.grid_sample(grid, all_ind_norm, mode='bilinear', align_corners=True)
    Kd = len(displace_list)
    all_ind = all_ind.view(1, 1, 1, -1, 6, Kd, 3)
    diff = (all_ind[:, :, :, :, 1::2] - all_ind[:, :, :, :, 0::2]).max(dim=-1)[0]
    feat_ = feat.view(1, 1, 1, -1, 6, Kd)
    grad = (feat_[:, :, :, :, 1::2] - feat_[:, :, :, :, 0::2]) / diff / voxel_size
    feat = feat.view(M, 6, Kd)
    grad = grad.view(M, 3, Kd)
    if use_grad_norm:
        grad = grad / (grad.norm(dim=1, keepdim=True) + 1e-5)
    return feat.reshape(M, 6 * Kd), grad.reshape(M, 3 * Kd)


def grid_sampler_ret_grad(xyz, grid, xyz_min, xyz_max, voxel_size):
    """model/nerf.py:639-672 with sample_ret=True, sample_grad=True: value, xyz-ordered gradient
    [M,3] and xyz-ordered taps [M,6]."""
    val = dense_grid_forward(grid, xyz, xyz_min, xyz_max)
    feat, grad = sample_sdfs(xyz, grid, xyz_min, xyz_max, voxel_size, [1.0], use_grad_norm=False)
    feat = torch.cat([feat[:, 4:6], feat[:, 2:4], feat[:, 0:2]], dim=-1)
    grad = torch.cat([grad[:, [2]], grad[:, [1]], grad[:, [0]]], dim=-1)
    return val, grad, feat


def grad_conv_weight(voxel_size, sigma: float = 0) -> torch.Tensor:
    """model/nerf.py:224-247: the [3,1,3,3,3] weight of `grad_conv` (Sobel-like taps over the 3^3 neighbourhood, sign and zero
    planes per component, scaled by 1 / (plane sum * 2 * voxel_size))."""
    kernel = np.asarray([[[1, 2, 1], [2, 4, 2], [1, 2, 1]], [[2, 4, 2], [4, 8, 4], [2, 4, 2]], [[1, 2, 1], [2, 4, 2], [1, 2, 1]]])
    distance = np.zeros((3, 3, 3))
    for i in range(3):
        for j in range(3):
            for k in range(3):
                distance[i, j, k] = ((i - 1) ** 2 + (j - 1) ** 2 + (k - 1) ** 2 - 1)
    kernel0 = kernel * np.exp(-distance * sigma)
    kernel1 = kernel0 / (kernel0[0].sum() * 2 * float(voxel_size))
    weight = torch.from_numpy(np.concatenate([kernel1[None] for _ in range(3)])).float()
    weight[0, 1, :, :] *= 0
    weight[0, 0, :, :] *= -1
    weight[1, :, 1, :] *= 0
    weight[1, :, 0, :] *= -1
    weight[2, :, :, 1] *= 0
    weight[2, :, :, 0] *= -1
    return weight.unsqueeze(1).float()


def neus_sdf_gradient(sdf: torch.Tensor, voxel_size, mode: str = 'interpolate', conv_weight=None) -> torch.Tensor:
    """model/nerf.py:485-508: 'interpolate' (interior central difference, zero faces), 'raw' (forward difference, zero last
    face), 'grad_conv' (Conv3d(1, 3, 3, padding=1, padding_mode='replicate') with `grad_conv_weight`, zero bias)."""
    if mode == 'grad_conv':
        w = (conv_weight if conv_weight is not None else grad_conv_weight(voxel_size)).to(sdf.dtype)
        return F.conv3d(F.pad(sdf, (1,) * 6, mode='replicate'), w, bias=torch.zeros(3, dtype=sdf.dtype))
    g = torch.zeros([1, 3, *sdf.shape[-3:]], dtype=sdf.dtype)
    if mode == 'raw':
        g[:, 0, :-1, :, :] = (sdf[:, 0, 1:, :, :] - sdf[:, 0, :-1, :, :]) / voxel_size
        g[:, 1, :, :-1, :] = (sdf[:, 0, :, 1:, :] - sdf[:, 0, :, :-1, :]) / voxel_size
        g[:, 2, :, :, :-1] = (sdf[:, 0, :, :, 1:] - sdf[:, 0, :, :, :-1]) / voxel_size
        return g
    assert mode == 'interpolate', mode
    g[:, 0, 1:-1, :, :] = (sdf[:, 0, 2:, :, :] - sdf[:, 0, :-2, :, :]) / 2 / voxel_size
    g[:, 1, :, 1:-1, :] = (sdf[:, 0, :, 2:, :] - sdf[:, 0, :, :-2, :]) / 2 / voxel_size
    g[:, 2, :, :, 1:-1] = (sdf[:, 0, :, :, 2:] - sdf[:, 0, :, :, :-2]) / 2 / voxel_size
    return g


def gaussian_kernel3d(ksize: int, sigma: float) -> torch.Tensor:
    """model/nerf.py:260-268: normalised exp(-(r^2)/(2 sigma^2)) taps, [k,k,k] float32."""
    ax = np.arange(-(ksize // 2), ksize // 2 + 1, 1)
    xx, yy, zz = np.meshgrid(ax, ax, ax)
    k = np.exp(-(xx ** 2 + yy ** 2 + zz ** 2) / (2 * sigma ** 2))
    k = torch.from_numpy(k).float()
    return k / k.sum()


def total_variation(v: torch.Tensor, mask=None, variant: str = "nerf") -> torch.Tensor:
    """total_variation(v, mask): variant "nerf" = model/nerf.py:1212-1221 ((tv2 + tv3 + tv4).sum() / 3 / mask.sum(), or
    / v.sum() without a mask), variant "dvgo" = model/dvgo.py:420-428 (per-axis means / 3).  Any float dtype (the parity
    tests run it in float64)."""
    tv = [v.diff(dim=d).abs() for d in (2, 3, 4)]
    if mask is not None:
        tv = [t[mask.narrow(d, 0, t.shape[d]) & mask.narrow(d, 1, t.shape[d])] for d, t in zip((2, 3, 4), tv)]
    if variant == "dvgo":
        return (tv[0].mean() + tv[1].mean() + tv[2].mean()) / 3
    total = tv[0].sum() + tv[1].sum() + tv[2].sum()
    return total / 3 / (mask.sum() if mask is not None else v.sum())


def tv_smooth_kernel(sigma: float = 0) -> torch.Tensor:
    """model/nerf.py:226-236,250-252: the 3^3 binomial taps of `tv_smooth_conv` (kernel0 / kernel0.sum()), [3,3,3] float32."""
    kernel = np.asarray([[[1, 2, 1], [2, 4, 2], [1, 2, 1]], [[2, 4, 2], [4, 8, 4], [2, 4, 2]], [[1, 2, 1], [2, 4, 2], [1, 2, 1]]])
    distance = np.zeros((3, 3, 3))
    for i in range(3):
        for j in range(3):
            for k in range(3):
                distance[i, j, k] = ((i - 1) ** 2 + (j - 1) ** 2 + (k - 1) ** 2 - 1)
    kernel0 = kernel * np.exp(-distance * sigma)
    return torch.from_numpy(kernel0 / kernel0.sum()).float()


def smooth_conv(grid: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """model/nerf.py:267-272: Conv3d(1,1,k, padding=k//2, padding_mode='replicate'), zero bias."""
    k = kernel.shape[0]
    padded = F.pad(grid, (k // 2,) * 6, mode='replicate')
    return F.conv3d(padded, kernel[None, None].to(grid.dtype), bias=torch.zeros(1, dtype=grid.dtype))


def s_val_schedule(global_step, s_ratio, s_start, step_start=0) -> float:
    """model/nerf.py:514."""
    return 1. / (global_step + s_ratio / s_start - step_start) * s_ratio


def neus_alpha_from_sdf_scatter(viewdirs, ray_id, dist, sdf, gradients, s_val: float, s_param=None):
    """model/nerf.py:510-544 (is_train, use_mid, cos_anneal_ratio=1).  `s_param`: the learnable s_val parameter of s_learn
    (:516-517; a [1] tensor that may require grad) -- otherwise the scheduled value (:514-515)."""
    if s_param is None:
        s_param = torch.ones(1) * s_val
    dirs = viewdirs[ray_id]
    inv_s = torch.ones(1) / s_param
    true_cos = (dirs * gradients).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - 1.0) + F.relu(-true_cos) * 1.0)
    sdf = sdf.unsqueeze(-1)
    est_next = sdf + iter_cos * dist.reshape(-1, 1) * 0.5
    est_prev = sdf - iter_cos * dist.reshape(-1, 1) * 0.5
    prev_cdf = torch.sigmoid(est_prev * inv_s.reshape(-1, 1))
    next_cdf = torch.sigmoid(est_next * inv_s.reshape(-1, 1))
    p = prev_cdf - next_cdf
    c = prev_cdf
    return ((p + 1e-5) / (c + 1e-5)).clip(0.0, 1.0).squeeze(-1)


class Alphas2Weights(torch.autograd.Function):
    """model/nerf.py:1173-1189 over the C restatement of alpha2weight(_backward)."""

    @staticmethod
    def forward(ctx, alpha, ray_id, N):
        w, T, last, i_s, i_e = K.alpha2weight(alpha.detach().numpy(), ray_id.numpy(), N)
        w, T, last = map(torch.from_numpy, (w, T, last))
        i_s, i_e = torch.from_numpy(i_s), torch.from_numpy(i_e)
        if alpha.requires_grad:
            ctx.save_for_backward(alpha, w, T, last, i_s, i_e)
            ctx.n_rays = N
        return w, last

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_weights, grad_last):
        alpha, w, T, last, i_s, i_e = ctx.saved_tensors
        g = K.alpha2weight_backward(alpha.detach().numpy(), w.numpy(), T.numpy(), last.numpy(), i_s.numpy(), i_e.numpy(),
                                    ctx.n_rays, grad_weights.contiguous().numpy(), grad_last.contiguous().numpy())
        return torch.from_numpy(g), None, None


class Alphas2WeightsF64(torch.autograd.Function):
    """The same recurrences as alpha2weight / alpha2weight_backward (render_utils_kernel.cu:576-605, 653-677) carried out
    in float64, for the fp64 error yardstick of the parity tests.  The per-ray stopping point (the first sample after
    which T < 1e-3) is a DECISION, taken from the float32 run (`i_end`), so that both precisions weight the same samples."""

    @staticmethod
    def forward(ctx, alpha, ray_id, N, i_end):
        a = alpha.detach().numpy().astype(np.float64)
        rid = ray_id.numpy()
        M = a.shape[0]
        w, T, last = np.zeros(M), np.ones(M), np.ones(N)
        i_start = np.searchsorted(rid, np.arange(N), side='left')
        i_end = np.asarray(i_end, dtype=np.int64)
        for r in range(N):
            s, e = int(i_start[r]), int(i_end[r])
            if e <= s:
                continue
            cp = np.cumprod(1.0 - a[s:e])
            T[s + 1:e] = cp[:-1]
            w[s:e] = T[s:e] * a[s:e]
            last[r] = cp[-1]
        w_t, last_t = torch.from_numpy(w), torch.from_numpy(last)
        ctx.save_for_backward(alpha.detach().double(), w_t, torch.from_numpy(T), last_t, torch.from_numpy(i_start),
                              torch.from_numpy(i_end))
        ctx.n_rays = N
        return w_t, last_t

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_weights, grad_last):
        alpha, w, T, last, i_start, i_end = (t.numpy() for t in ctx.saved_tensors)
        gw, gl = grad_weights.numpy().astype(np.float64), grad_last.numpy().astype(np.float64)
        g = np.zeros_like(alpha)
        for r in range(ctx.n_rays):
            s, e = int(i_start[r]), int(i_end[r])
            if e <= s:
                continue
            gww = gw[s:e] * w[s:e]
            # back_cum seen by sample i = grad_last*alphainv_last + sum_{j>i} gw_j w_j   (:660-676)
            tail = np.concatenate([np.cumsum(gww[::-1])[::-1][1:], [0.0]])
            back = gl[r] * last[r] + tail
            g[s:e] = gw[s:e] * T[s:e] - back / ((1.0 - alpha[s:e]) + 1e-10)
        return torch.from_numpy(g), None, None, None


def alphas2weights(alpha, ray_id, N, force_end=None):
    """(weights, alphainv_last, i_end).  float32: the C restatement; float64 (needs `force_end` from a float32 run): the
    same recurrences in double."""
    if alpha.dtype == torch.float64:
        assert force_end is not None, "the float64 yardstick takes its stopping points from a float32 run"
        w, last = Alphas2WeightsF64.apply(alpha, ray_id, N, force_end)
        return w, last, np.asarray(force_end)
    w, last = Alphas2Weights.apply(alpha, ray_id, N)
    _, _, _, _, i_e = K.alpha2weight(alpha.detach().numpy(), ray_id.numpy(), N)
    return w, last, i_e


def segment_sum(src: torch.Tensor, index: torch.Tensor, n: int) -> torch.Tensor:
    """torch_scatter.segment_coo(src, index, out=zeros([n, ...]), reduce='sum') for sorted `index`
    (model/nerf.py:888-896): a plain index_add_ into zeros."""
    out = torch.zeros([n, *src.shape[1:]], dtype=src.dtype)
    return out.index_add_(0, index, src)


def posenc(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """model/nerf.py:838-839: [x, sin(x*f_i), cos(x*f_i)] with (component-major, frequency-minor) order."""
    emb = (x.unsqueeze(-1) * freqs).flatten(-2)
    return torch.cat([x, emb.sin(), emb.cos()], -1)


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """model/nerf.py:480-483."""
    eps = torch.tensor(torch.finfo(torch.float32).eps)
    return x / torch.sqrt(torch.maximum(torch.sum(x ** 2, dim=-1, keepdims=True), eps))


def mlp_apply(layers: List[Tuple[torch.Tensor, torch.Tensor]], x: torch.Tensor, relu_masks=None, stats=None) -> torch.Tensor:
    """model/nerf.py:125-142: Linear(+ReLU) stack, no activation after the last Linear.
    `relu_masks` (one bool tensor per hidden layer, from ANOTHER float32 evaluation of the same network on the same inputs):
    the sign decisions of the ReLUs are replayed instead of re-taken -- like forward_fine's `decisions` -- so that the two
    evaluations differentiate the same piecewise-linear function; `stats['relu_flips']` counts the units whose own sign
    differs from the replayed one (pre-activations within rounding of zero)."""
    for i, (W, b) in enumerate(layers):
        x = F.linear(x, W, b)
        if i + 1 < len(layers):
            if relu_masks is not None:
                m = relu_masks[i].to(x.device)
                if stats is not None:
                    stats['relu_flips'] = stats.get('relu_flips', 0) + int(((x > 0) != m).sum())
                    stats['relu_units'] = stats.get('relu_units', 0) + m.numel()
                x = x * m.to(x.dtype)
            else:
                x = F.relu(x)
    return x


def mask_cache_forward(mc: Dict, xyz: torch.Tensor) -> torch.Tensor:
    """model/nerf.py:1202-1209: trilinear sample of the max-pooled sdf_mask >= thres."""
    shape = xyz.shape[:-1]
    pts = xyz.reshape(1, 1, 1, -1, 3)
    v = F.grid_sample(mc['sdf_mask'], ind_norm_of(pts, mc['xyz_min'], mc['xyz_max']), align_corners=True)
    return v.reshape(*shape) >= mc['thres']


def make_mask_cache(sdf_mask_grid: torch.Tensor, xyz_min, xyz_max, thres: float, ks: int = 3) -> Dict:
    """model/nerf.py:1193-1200."""
    return dict(sdf_mask=F.max_pool3d(sdf_mask_grid, kernel_size=ks, padding=ks // 2, stride=1),
                xyz_min=xyz_min, xyz_max=xyz_max, thres=thres)


def sample_ray(P: Dict, rays_o, rays_d, near, stepsize):
    """model/nerf.py:674-698 (`far` forced to 1e9, compaction by ~mask_outbbox)."""
    stepdist = float(stepsize * P['voxel_size'])
    pts, mask_out, ray_id, step_id, n_steps, t_min, t_max = K.sample_pts_on_rays(
        rays_o.numpy(), rays_d.numpy(), P['xyz_min'].numpy(), P['xyz_max'].numpy(), near, 1e9, stepdist)
    inb = ~mask_out
    return (torch.from_numpy(pts[inb]), torch.from_numpy(ray_id[inb]), torch.from_numpy(step_id[inb]),
            torch.from_numpy(mask_out), int(pts.shape[0]))


def params_f64(P: Dict) -> Dict:
    """Copy of an oracle parameter dict with the grids and MLP tensors in float64 (fresh leaves); scalars that the
    reference rounds to float32 before use (voxel_size, bbox, frequencies) stay float32 VALUES, promoted on use."""
    Q = dict(P)
    Q['sdf'], Q['k0'] = P['sdf'].detach().double(), P['k0'].detach().double()
    for net in ('rgbnet', 'refnet'):
        if P.get(net) is not None:
            Q[net] = [(W.detach().double(), b.detach().double()) for W, b in P[net]]
    if P.get('smooth_kernel') is not None:
        Q['smooth_kernel'] = P['smooth_kernel'].double()
    return Q


def sample_ray_ori(P: Dict, grid_shape, rays_o, rays_d, near, far, stepsize):
    """model/nerf.py:734-758 (is_train=False): padded [..., N_samples, 3] sampling by the slab test in torch; used by the
    mask-cache ray pre-filter (model/nerf_ray.py:230-231).  `grid_shape` = sdf.grid.shape[2:]."""
    N_samples = int(np.linalg.norm(np.array(list(grid_shape)) + 1) / stepsize) + 1
    vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
    rate_a = (P['xyz_max'] - rays_o) / vec
    rate_b = (P['xyz_min'] - rays_o) / vec
    t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=near, max=far)
    t_max = torch.maximum(rate_a, rate_b).amin(-1).clamp(min=near, max=far)
    mask_outbbox = (t_max <= t_min)
    rng = torch.arange(N_samples)[None].float()
    step = stepsize * P['voxel_size'] * rng
    interpx = (t_min[..., None] + step / rays_d.norm(dim=-1, keepdim=True))
    rays_pts = rays_o[..., None, :] + rays_d[..., None, :] * interpx[..., None]
    mask_outbbox = mask_outbbox[..., None] | ((P['xyz_min'] > rays_pts) | (rays_pts > P['xyz_max'])).any(dim=-1)
    return rays_pts, mask_outbbox, step


class _Seams:
    """Stage seams of a `forward_fine(..., staged=True)` run: between the four segments of the path -- A march (sampling, SDF
    lookups, NeuS alpha, Alphas2Weights, the two threshold compactions), B per-survivor features (k0 lookup, hierarchical taps,
    encodings, normal, reflection), C the two MLPs, D sigmoid + compositing -- every tensor that crosses over is cut
    (`detach().requires_grad_()`), so that the backward pass can be run ONE SEGMENT AT A TIME with a chosen upstream gradient.
    `backward(loss)` does that with the oracle's own gradients and returns, per seam, exactly what the corresponding HIP
    backward stage receives and must produce (tests/test_stagewise_bwd_gpu.py: k_composite_bwd, k_head_bwd + the data-gradient
    chain + k_gemm + k_mlp_wgrad, k_feat_*_bwd, k_march_fine_bwd + the sdf scatter)."""

    def __init__(self):
        self.cuts = {}        # name -> (output of the producing segment, the leaf the consuming segment reads)

    def cut(self, name, t):
        leaf = t.detach().requires_grad_(True)
        self.cuts[name] = (t, leaf)
        return leaf

    def backward(self, loss, P):
        """Segment-wise backward pass.  Returns {seam name: gradient} for every cut, plus 'sdf_march' / 'sdf_taps' (the two
        parts of sdf.grad: segment A and the hierarchical taps of segment B), 'k0', 'rgbnet' / 'refnet' ([(dW, db), ...])."""
        c = self.cuts
        g = {}

        def up(name):
            return c[name][1].grad

        def out(name):
            return c[name][0]

        # E: the loss reads leaves of D / B / A outputs
        loss.backward()
        for k in ('rgb_marched', 'sigmoid_rgb', 'alphainv_last_loss', 'raw_rgb', 'normal_loss'):
            g[k] = up(k)
        zero = torch.zeros_like

        def bw(outs, grads):
            pairs = [(o, gr) for o, gr in zip(outs, grads) if gr is not None and o.requires_grad]
            if pairs:
                torch.autograd.backward([o for o, _ in pairs], [gr for _, gr in pairs])

        # D: sigmoid + compositing
        bw([out('rgb_marched'), out('sigmoid_rgb'), out('raw_rgb')], [g['rgb_marched'], g['sigmoid_rgb'], g['raw_rgb']])
        g['logit'], g['weights'] = up('logit'), up('weights')
        # C: the MLPs
        bw([out('logit')], [g['logit']])
        g['X0'], g['reflect_emb'] = up('X0'), up('reflect_emb')
        g['rgbnet'] = [(W.grad.clone(), b.grad.clone()) for W, b in P['rgbnet']]
        g['refnet'] = [(W.grad.clone(), b.grad.clone()) for W, b in P['refnet']]
        # B: features
        bw([out('X0'), out('reflect_emb'), out('normal_loss')], [g['X0'], g['reflect_emb'], g['normal_loss']])
        g['sdf_s'], g['gradient_s'] = up('sdf_s'), up('gradient_s')
        g['k0'] = P['k0'].grad.clone()
        g['sdf_taps'] = up('sdf_grid_B') if up('sdf_grid_B') is not None else zero(P['sdf'])
        # A: march
        if P['sdf'].grad is not None:
            P['sdf'].grad = None
        bw([out('weights'), out('alphainv_last_loss'), out('sdf_s'), out('gradient_s')],
           [g['weights'], g['alphainv_last_loss'], g['sdf_s'], g['gradient_s']])
        g['sdf_march'] = P['sdf'].grad.clone() if P['sdf'].grad is not None else zero(P['sdf'])
        return g


def forward_fine(P: Dict, rays_o, rays_d, viewdirs, global_step, near, stepsize, bg,
                 render_depth=False, render_grad=False, decisions: Optional[Dict] = None, staged: bool = False,
                 relu_masks: Optional[Dict] = None) -> Dict:
    """model/nerf.py:776-941.  P holds: xyz_min, xyz_max, voxel_size (0-d fp32 tensor), sdf [1,1,X,Y,Z],
    k0 [1,C,X,Y,Z], rgbnet / refnet (lists of (W,b)), posfreq, viewfreq, reffreq, fast_color_thres,
    s_ratio, s_start, grad_feat_displace (sorted tuple), use_grad_norm, center_sdf, optional
    mask_cache, optional smooth_kernel.
    `staged`: the same statements in the same order, with every tensor that crosses a stage seam cut into a fresh leaf
    (class _Seams; the result dict then carries 'seams'); values are identical, gradients are taken segment by segment.
    `relu_masks` = {'rgbnet': [...], 'refnet': [...]}: ReLU sign decisions replayed from another evaluation (mlp_apply); the
    result dict's 'relu_stats' then says how many units that concerned."""
    # `decisions` (the 'decisions' entry of an earlier float32 run on the same inputs): every discrete choice of the
    # path -- mask-cache skip, alpha > thres, the T < 1e-3 stopping point, weights > thres -- is replayed instead of
    # re-taken, so that a float64 run (P from params_f64) differentiates the SAME sample set: the error yardstick of
    # the full-size parity tests.  Without it the function is the reference path, decisions included.
    dec, rec = decisions, {}
    seams = _Seams() if staged else None
    cut = seams.cut if staged else (lambda name, t: t)
    N = len(rays_o)
    dt = P['sdf'].dtype
    xyz_min, xyz_max, voxel_size = P['xyz_min'], P['xyz_max'], P['voxel_size']
    # ---------------------------------------------------------------------------------------------------- A: march
    ray_pts, ray_id, step_id, mask_outbbox, m_total = sample_ray(P, rays_o, rays_d, near, stepsize)
    ray_pts, viewdirs = ray_pts.to(dt), viewdirs.to(dt)
    n_inbbox = int(ray_pts.shape[0])
    if P.get('mask_cache') is not None:
        m = dec['mc'] if dec else mask_cache_forward(P['mask_cache'], ray_pts)
        rec['mc'] = m
        ray_pts, ray_id, step_id = ray_pts[m], ray_id[m], step_id[m]
        mask_outbbox[~mask_outbbox] |= ~m
    sdf_grid = smooth_conv(P['sdf'], P['smooth_kernel']) if P.get('smooth_kernel') is not None else P['sdf']
    sdf, gradient, _ = grid_sampler_ret_grad(ray_pts, sdf_grid, xyz_min, xyz_max, voxel_size)
    dist = stepsize * voxel_size
    s_val = float(P['s_param']) if P.get('s_param') is not None else s_val_schedule(global_step, P['s_ratio'], P['s_start'])
    alpha = neus_alpha_from_sdf_scatter(viewdirs, ray_id, dist, sdf, gradient, s_val, P.get('s_param'))
    mask = None
    viewdirs_pts = viewdirs[ray_id]
    thres = P['fast_color_thres']
    if thres > 0:
        mask = dec['alpha_mask'] if dec else alpha > thres
        rec['alpha_mask'] = mask
        alpha, ray_id, viewdirs_pts, ray_pts = alpha[mask], ray_id[mask], viewdirs_pts[mask], ray_pts[mask]
        step_id, gradient, sdf = step_id[mask], gradient[mask], sdf[mask]
    weights, alphainv_last, rec['i_end'] = alphas2weights(alpha, ray_id, N, dec['i_end'] if dec else None)
    if thres > 0:
        mask = dec['weight_mask'] if dec else weights > thres
        rec['weight_mask'] = mask
        weights, alpha, ray_pts, viewdirs_pts = weights[mask], alpha[mask], ray_pts[mask], viewdirs_pts[mask]
        ray_id, step_id, gradient, sdf = ray_id[mask], step_id[mask], gradient[mask], sdf[mask]
    weights, alphainv_last = cut('weights', weights), cut('alphainv_last_loss', alphainv_last)
    sdf, gradient = cut('sdf_s', sdf), cut('gradient_s', gradient)
    sdf_grid_taps = cut('sdf_grid_B', sdf_grid)
    # ---------------------------------------------------------------------------------------------------- B: features
    normal = l2_normalize(gradient / (gradient.norm(dim=-1, keepdim=True) + 1e-7))
    rays_xyz = (ray_pts - xyz_min) / (xyz_max - xyz_min)
    xyz_emb = posenc(rays_xyz, P['posfreq'])
    k0 = dense_grid_forward(P['k0'], ray_pts, xyz_min, xyz_max)
    hier = []
    if P.get('center_sdf', True):
        hier.append(sdf[:, None])
    disp = P.get('grad_feat_displace', ())
    if len(disp) > 0:
        all_feat, all_grad = sample_sdfs(ray_pts, sdf_grid_taps, xyz_min, xyz_max, voxel_size, sorted(disp),
                                         use_grad_norm=P.get('use_grad_norm', True))
        hier += [all_feat, all_grad]
    viewdirs_emb = posenc(viewdirs, P['viewfreq'])[ray_id]
    rgb_feat = torch.cat([k0, xyz_emb, viewdirs_emb, *hier, gradient], dim=-1)
    reflect_r = viewdirs_pts - 2. * torch.sum(viewdirs_pts * normal, dim=-1, keepdim=True) * normal
    reflect_emb = posenc(reflect_r, P['reffreq'])
    rgb_feat, reflect_emb = cut('X0', rgb_feat), cut('reflect_emb', reflect_emb)
    normal = cut('normal_loss', normal)
    # ---------------------------------------------------------------------------------------------------- C: the MLPs
    relu_stats = {}
    rgb_feat = mlp_apply(P['rgbnet'], rgb_feat, relu_masks['rgbnet'] if relu_masks else None, relu_stats)
    ref_feat = torch.cat([rgb_feat, reflect_emb], dim=-1)
    logit = cut('logit', mlp_apply(P['refnet'], ref_feat, relu_masks['refnet'] if relu_masks else None, relu_stats))
    # ---------------------------------------------------------------------------------------------------- D: compositing
    rgb = torch.sigmoid(logit)
    sig_rgb = torch.sigmoid(rgb)
    rgb_marched = segment_sum(weights.unsqueeze(-1) * rgb, ray_id, N)
    cum_weights = segment_sum(weights.unsqueeze(-1), ray_id, N)
    sigmoid_rgb = segment_sum(weights.unsqueeze(-1) * sig_rgb, ray_id, N)
    rgb_marched = (rgb_marched + (1 - cum_weights) * bg).clamp(0, 1)
    sigmoid_rgb = (sigmoid_rgb + (1 - cum_weights) * bg).clamp(0, 1)
    normal_marched = segment_sum(weights.unsqueeze(-1) * normal, ray_id, N) if render_grad else None
    rgb_marched, sigmoid_rgb, rgb = cut('rgb_marched', rgb_marched), cut('sigmoid_rgb', sigmoid_rgb), cut('raw_rgb', rgb)
    depth = disp_map = None
    if render_depth:
        with torch.no_grad():
            depth = segment_sum(weights * step_id * dist, ray_id, N)
            disp_map = 1 / depth
    return {
        'alphainv_cum': alphainv_last, 'weights': weights, 'ray_id': ray_id, 'viewdirs': viewdirs[ray_id],
        'rgb_marched': rgb_marched, 'sigmoid_rgb': sigmoid_rgb, 'normal_marched': normal_marched,
        'normal': normal, 'raw_alpha': alpha, 'raw_rgb': rgb, 'depth': depth, 'disp': disp_map, 'mask': mask,
        'mask_outbbox': mask_outbbox, 'gradient': gradient, 's_val': s_val,
        # bookkeeping for the bench/tests (not in the reference dict)
        'step_id': step_id, 'n_total': m_total, 'n_inbbox': n_inbbox, 'sdf': sdf, 'decisions': rec, 'seams': seams,
        'relu_stats': relu_stats,
    }


def forward_coarse(P: Dict, rays_o, rays_d, viewdirs, global_step, near, stepsize, bg, stage='coarse',
                   render_depth=True, render_grad=False, decisions: Optional[Dict] = None,
                   relu_masks: Optional[Dict] = None) -> Dict:
    """model/nerf.py:943-1075.  Extra keys in P: inc_mask (dict(mask, scale, shift) or None).  `decisions`, `relu_masks`
    ({'refnet': [...]}): see forward_fine."""
    dec, rec = decisions, {}
    N = len(rays_o)
    dt = P['sdf'].dtype
    xyz_min, xyz_max, voxel_size = P['xyz_min'], P['xyz_max'], P['voxel_size']
    ray_pts, ray_id, step_id, mask_outbbox, m_total = sample_ray(P, rays_o, rays_d, near, stepsize)
    n_inbbox = int(ray_pts.shape[0])
    ray_pts32 = ray_pts
    ray_pts, viewdirs = ray_pts.to(dt), viewdirs.to(dt)
    viewdirs_pts = viewdirs[ray_id]
    if stage == 'coarse' and P.get('mask_cache') is not None:
        m = dec['mc'] if dec else mask_cache_forward(P['mask_cache'], ray_pts)
        rec['mc'] = m
        ray_pts, ray_id, viewdirs_pts, step_id = ray_pts[m], ray_id[m], viewdirs_pts[m], step_id[m]
        ray_pts32 = ray_pts32[m]
        mask_outbbox[~mask_outbbox] |= ~m
    if P.get('inc_mask') is not None:
        im = P['inc_mask']
        m = torch.from_numpy(K.maskcache_lookup(im['mask'].numpy(), ray_pts32.numpy(), im['scale'].numpy(),
                                                im['shift'].numpy()))
        ray_pts, ray_id, viewdirs_pts, step_id = ray_pts[m], ray_id[m], viewdirs_pts[m], step_id[m]
    sdf_grid = smooth_conv(P['sdf'], P['smooth_kernel']) if P.get('smooth_kernel') is not None else P['sdf']
    sdf = dense_grid_forward(sdf_grid, ray_pts, xyz_min, xyz_max)
    grad_vol = neus_sdf_gradient(P['sdf'], voxel_size, P.get('grad_mode', 'interpolate'), P.get('grad_conv_w'))   # nerf.py:972
    gradient = dense_grid_forward(grad_vol, ray_pts, xyz_min, xyz_max)
    dist = stepsize * voxel_size
    s_val = float(P['s_param']) if P.get('s_param') is not None else s_val_schedule(global_step, P['s_ratio'], P['s_start'])
    alpha = neus_alpha_from_sdf_scatter(viewdirs, ray_id, dist, sdf, gradient, s_val, P.get('s_param'))
    mask = None
    thres = P['fast_color_thres']
    pass1 = None
    if dec is None:
        weights, alphainv_last, rec['i_end'] = alphas2weights(alpha, ray_id, N)
        pass1 = dict(ray_id=ray_id, step_id=step_id, weights=weights.detach(), raw_alpha=alpha.detach())   # bookkeeping
    if thres > 0:
        mask = dec['weight_mask'] if dec else weights > thres
        rec['weight_mask'] = mask
        ray_pts, ray_id, viewdirs_pts, step_id = ray_pts[mask], ray_id[mask], viewdirs_pts[mask], step_id[mask]
        alpha, gradient = alpha[mask], gradient[mask]
        weights, alphainv_last, rec['i_end'] = alphas2weights(alpha, ray_id, N, dec['i_end'] if dec else None)
    elif dec is not None:
        weights, alphainv_last, rec['i_end'] = alphas2weights(alpha, ray_id, N, dec['i_end'])
    normal = l2_normalize(gradient / (gradient.norm(dim=-1, keepdim=True) + 1e-7))
    rays_xyz = (ray_pts - xyz_min) / (xyz_max - xyz_min)
    xyz_emb = posenc(rays_xyz, P['posfreq'])
    k0 = dense_grid_forward(P['k0'], ray_pts, xyz_min, xyz_max)
    reflect_r = viewdirs_pts - 2. * torch.sum(viewdirs_pts * normal, dim=-1, keepdim=True) * normal
    reflect_emb = posenc(reflect_r, P['reffreq'])
    viewdirs_emb = posenc(viewdirs, P['viewfreq'])[ray_id]
    ref_feat = torch.cat([k0, xyz_emb, reflect_emb, normal, viewdirs_emb], dim=-1)
    relu_stats = {}
    rgb = torch.sigmoid(mlp_apply(P['refnet'], ref_feat, relu_masks['refnet'] if relu_masks else None, relu_stats))
    sig_rgb = torch.sigmoid(rgb)
    rgb_marched = segment_sum(weights.unsqueeze(-1) * rgb, ray_id, N)
    sigmoid_rgb = segment_sum(weights.unsqueeze(-1) * sig_rgb, ray_id, N)
    cum_weights = segment_sum(weights.unsqueeze(-1), ray_id, N)
    rgb_marched = (rgb_marched + (1 - cum_weights) * bg).clamp(0, 1)
    sigmoid_rgb = (sigmoid_rgb + (1 - cum_weights) * bg).clamp(0, 1)
    normal_marched = segment_sum(weights.unsqueeze(-1) * normal, ray_id, N) if render_grad else None
    depth = disp_map = None
    if render_depth:
        with torch.no_grad():
            depth = segment_sum(weights * step_id * dist, ray_id, N)
            disp_map = 1 / depth
    return {
        'alphainv_cum': alphainv_last, 'weights': weights, 'ray_id': ray_id, 'viewdirs': viewdirs[ray_id],
        'rgb_marched': rgb_marched, 'sigmoid_rgb': sigmoid_rgb, 'normal_marched': normal_marched,
        'normal': normal, 'raw_alpha': alpha, 'raw_rgb': rgb, 'depth': depth, 'disp': disp_map, 'mask': mask,
        'mask_outbbox': mask_outbbox, 'gradient': gradient, 's_val': s_val,
        'step_id': step_id, 'n_total': m_total, 'n_inbbox': n_inbbox, 'decisions': rec, 'pass1': pass1,
        'relu_stats': relu_stats,
    }


def dvgo_forward(P: Dict, rays_o, rays_d, viewdirs, near, stepsize, bg) -> Dict:
    """model/dvgo.py:284-357.  P: xyz_min, xyz_max, voxel_size, voxel_size_ratio, density [1,1,..],
    k0 [1,3,..], act_shift, fast_color_thres."""
    N = len(rays_o)
    xyz_min, xyz_max, voxel_size = P['xyz_min'], P['xyz_max'], P['voxel_size']
    ray_pts, ray_id, step_id, _, m_total = sample_ray(P, rays_o, rays_d, near, stepsize)
    n_inbbox = int(ray_pts.shape[0])
    interval = stepsize * P['voxel_size_ratio']
    density = dense_grid_forward(P['density'], ray_pts, xyz_min, xyz_max)
    alpha = 1 - torch.exp(-F.softplus(density + P['act_shift']) * interval)  # dvgo.py:225-227
    thres = P['fast_color_thres']
    mask = None
    if thres > 0:
        mask = alpha > thres
        ray_pts, ray_id, step_id, alpha = ray_pts[mask], ray_id[mask], step_id[mask], alpha[mask]
    weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, N)
    if thres > 0:
        mask = weights > thres
        weights, alpha, ray_pts, ray_id, step_id = weights[mask], alpha[mask], ray_pts[mask], ray_id[mask], step_id[mask]
    rgb = torch.sigmoid(dense_grid_forward(P['k0'], ray_pts, xyz_min, xyz_max))
    grad_vol = neus_sdf_gradient(P['density'], voxel_size)  # dvgo.py:271-277 is the same stencil
    gradient = dense_grid_forward(grad_vol, ray_pts, xyz_min, xyz_max)
    normals = gradient / (gradient.norm(dim=-1, keepdim=True) + 1e-7)
    rgb_marched = segment_sum(weights.unsqueeze(-1) * rgb, ray_id, N)
    rgb_marched = rgb_marched + alphainv_last.unsqueeze(-1) * bg
    normal_marched = segment_sum(weights.unsqueeze(-1) * normals, ray_id, N)
    return {'alphainv_cum': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_alpha': alpha,
            'raw_rgb': rgb, 'normal_marched': normal_marched, 'ray_id': ray_id, 'mask': mask,
            'n_total': m_total, 'n_inbbox': n_inbbox}


def fine_losses(res: Dict, target: torch.Tensor, cfg: Dict) -> torch.Tensor:
    """model/nerf_training.py:308-327 (the ray-dependent loss terms of one iteration)."""
    loss = cfg.get('weight_main', 1.0) * F.mse_loss(res['rgb_marched'], target)
    n_rays = target.shape[0]
    if cfg.get('weight_rgbper', 0) > 0:
        rgbper = (res['raw_rgb'] - target[res['ray_id']]).pow(2).sum(-1)
        loss = loss + cfg['weight_rgbper'] * (rgbper * res['weights'].detach()).sum() / n_rays
    if cfg.get('weight_entropy_last', 0) > 0:
        pout = res['alphainv_cum'][..., -1].clamp(1e-6, 1 - 1e-6)  # single element: reference quirk
        ent = -(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean()
        loss = loss + cfg['weight_entropy_last'] * ent
    if cfg.get('weight_orientation', 0) > 0:
        w = res['weights'].detach()
        n_dot_v = (res['normal'] * (-res['viewdirs'])).sum(dim=-1)
        ori = torch.mean((w * torch.fmin(torch.tensor(0.0), n_dot_v) ** 2).sum(dim=-1))  # nerf.py:469-478
        loss = loss + cfg['weight_orientation'] * ori
    if cfg.get('sigmoid_rgb_loss', 0) > 0:
        loss = loss + cfg['sigmoid_rgb_loss'] * F.mse_loss(res['sigmoid_rgb'], target)
    return loss


# --------------------------------------------------------------------------------------------------------------------
# Integrated directional encoding: model/utils.py:168-210 (coefficients) and :515-574 (generate_ide_fn), restated with
# the same torch calls on the CPU (`.cuda()` dropped, `np.math` -> `math`: numpy 2 removed the alias).  The reference never
# evaluates it and holds no fixture: parity unpinned; tests/test_ide.py pins this restatement to scipy's Y_l^m instead.
def generate_ide_fn(deg_view):
    import math
    if deg_view > 5:
        raise ValueError('Only deg_view of at most 5 is numerically stable.')

    def generalized_binomial_coeff(a, k):
        return np.prod(a - np.arange(k)) / math.factorial(k)

    def assoc_legendre_coeff(l, m, k):
        return ((-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m) *
                generalized_binomial_coeff(0.5 * (l + k + m - 1.0), l))

    def sph_harm_coeff(l, m, k):
        return (np.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * np.pi * math.factorial(l + m))) *
                assoc_legendre_coeff(l, m, k))

    ml_list = []
    for i in range(deg_view):
        l = 2 ** i
        for m in range(l + 1):
            ml_list.append((m, l))
    ml_array = np.array(ml_list).T
    l_max = 2 ** (deg_view - 1)
    mat = torch.zeros(l_max + 1, ml_array.shape[1])
    for i, (m, l) in enumerate(ml_array.T):
        for k in range(l - m + 1):
            mat[k, i] = sph_harm_coeff(l, m, k)

    def integrated_dir_enc_fn(xyz, kappa_inv):
        x, y, z = xyz[..., 0:1], xyz[..., 1:2], xyz[..., 2:3]
        ml = torch.from_numpy(ml_array)
        vmz = torch.cat([z ** i for i in range(mat.shape[0])], dim=-1)
        vmxy = torch.cat([(x + 1j * y) ** m for m in ml[0, :]], dim=-1)
        sph_harms = vmxy * (vmz @ mat.to(vmz.dtype))     # float32 inputs (the reference's case): mat as is
        sigma = 0.5 * ml[1, :] * (ml[1, :] + 1)
        ide = sph_harms * torch.exp(-sigma * kappa_inv)
        return torch.cat([torch.real(ide), torch.imag(ide)], dim=-1)

    integrated_dir_enc_fn.ml_array = ml_array
    return integrated_dir_enc_fn
