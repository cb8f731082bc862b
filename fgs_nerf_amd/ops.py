"""The reference's three native extension modules, re-created over libfgs_hip.so.

``render_utils_cuda``, ``total_variation_cuda`` and ``adam_upd_cuda`` below expose exactly the
function names, argument orders, return orders, dtypes and error behaviour of the pybind modules
built by the reference at import time (model/cuda/render_utils.cpp:170-184,
model/cuda/total_variation.cpp:29-32, model/cuda/adam_upd.cpp:79-86) so code written against
them -- model/grid.py, model/nerf.py, model/dvgo.py, model/adam.py -- runs unchanged.  The names keep
the ``_cuda`` suffix because that is the reference's API; everything underneath is HIP on gfx950.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from ._lib import call, check_input, ptr, stream

F32, I64, BOOL = torch.float32, torch.int64, torch.bool


def grid_strides(g: torch.Tensor):
    """(C, X, Y, Z, sC, sX, sY, sZ) of a [1,C,X,Y,Z] tensor that is dense in either the channel-first
    (reference) or the channel-last (this build's DenseGrid) layout."""
    if g.dim() != 5 or g.shape[0] != 1:
        raise RuntimeError(f"expected a [1,C,X,Y,Z] grid, got {tuple(g.shape)}")
    _, C, X, Y, Z = g.shape
    _, sC, sX, sY, sZ = g.stride()
    first = (sZ == 1 and sY == Z and sX == Y * Z and (C == 1 or sC == X * Y * Z))
    last = (sZ == C and sY == Z * C and sX == Y * Z * C and (C == 1 or sC == 1))
    if not (first or last):
        raise RuntimeError("grid must be dense in channel-first or channel-last (channels_last_3d) layout")
    if C == 1:
        sC = X * Y * Z  # any value works; keep the channel-first convention
    return C, X, Y, Z, sC, sX, sY, sZ


# ------------------------------------------------------------------------------ render_utils_cuda

def infer_t_minmax(rays_o, rays_d, xyz_min, xyz_max, near, far):
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (xyz_min, "xyz_min"), (xyz_max, "xyz_max")):
        check_input(t, n, F32)
    n_rays = rays_o.size(0)
    t_min = torch.empty([n_rays], dtype=F32, device=rays_o.device)
    t_max = torch.empty([n_rays], dtype=F32, device=rays_o.device)
    call("fgs_infer_t_minmax", ptr(rays_o), ptr(rays_d), ptr(xyz_min), ptr(xyz_max), float(near), float(far), n_rays,
         ptr(t_min), ptr(t_max), stream())
    return [t_min, t_max]


def infer_n_samples(rays_d, t_min, t_max, stepdist):
    for t, n in ((rays_d, "rays_d"), (t_min, "t_min"), (t_max, "t_max")):
        check_input(t, n, F32)
    n_rays = t_min.size(0)
    out = torch.empty([n_rays], dtype=I64, device=t_min.device)
    call("fgs_infer_n_samples", ptr(rays_d), ptr(t_min), ptr(t_max), float(stepdist), n_rays, ptr(out), stream())
    return out


def infer_ray_start_dir(rays_o, rays_d, t_min):
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (t_min, "t_min")):
        check_input(t, n, F32)
    n_rays = rays_o.size(0)
    rays_start, rays_dir = torch.empty_like(rays_o), torch.empty_like(rays_o)
    call("fgs_infer_ray_start_dir", ptr(rays_o), ptr(rays_d), ptr(t_min), n_rays, ptr(rays_start), ptr(rays_dir), stream())
    return [rays_start, rays_dir]


def sample_count(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """Sync-free first half of sample_pts_on_rays: (N_steps, t_min, t_max, steps_cumsum[n+1]) on device."""
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (xyz_min, "xyz_min"), (xyz_max, "xyz_max")):
        check_input(t, n, F32)
    n_rays = rays_o.size(0)
    dev = rays_o.device
    n_steps = torch.empty([n_rays], dtype=I64, device=dev)
    t_min = torch.empty([n_rays], dtype=F32, device=dev)
    t_max = torch.empty([n_rays], dtype=F32, device=dev)
    cumsum = torch.empty([n_rays + 1], dtype=I64, device=dev)
    call("fgs_sample_count", ptr(rays_o), ptr(rays_d), ptr(xyz_min), ptr(xyz_max), float(near), float(far),
         float(stepdist), n_rays, ptr(n_steps), ptr(t_min), ptr(t_max), ptr(cumsum), stream())
    return n_steps, t_min, t_max, cumsum


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """-> [rays_pts f32[M,3], mask_outbbox bool[M], ray_id i64[M], step_id i64[M], N_steps i64[N], t_min, t_max].
    Like the reference (render_utils_kernel.cu:212) this reads M back to size the outputs."""
    n_steps, t_min, t_max, cumsum = sample_count(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist)
    n_rays = rays_o.size(0)
    dev = rays_o.device
    total = int(cumsum[n_rays].item())
    rays_pts = torch.empty([total, 3], dtype=F32, device=dev)
    mask_outbbox = torch.empty([total], dtype=BOOL, device=dev)
    ray_id = torch.empty([total], dtype=I64, device=dev)
    step_id = torch.empty([total], dtype=I64, device=dev)
    call("fgs_sample_emit", ptr(rays_o), ptr(rays_d), ptr(xyz_min), ptr(xyz_max), float(stepdist), n_rays, ptr(t_min),
         ptr(cumsum), total, ptr(rays_pts), ptr(mask_outbbox), ptr(ray_id), ptr(step_id), stream())
    return [rays_pts, mask_outbbox, ray_id, step_id, n_steps, t_min, t_max]


def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples):
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (xyz_min, "xyz_min"), (xyz_max, "xyz_max")):
        check_input(t, n, F32)
    n_rays = rays_o.size(0)
    rays_pts = torch.empty([n_rays, N_samples, 3], dtype=F32, device=rays_o.device)
    mask_outbbox = torch.empty([n_rays, N_samples], dtype=BOOL, device=rays_o.device)
    call("fgs_sample_ndc_pts", ptr(rays_o), ptr(rays_d), ptr(xyz_min), ptr(xyz_max), int(N_samples), n_rays,
         ptr(rays_pts), ptr(mask_outbbox), stream())
    return [rays_pts, mask_outbbox]


def sample_bg_pts_on_rays(rays_o, rays_d, t_max, bg_preserve, N_samples):
    for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (t_max, "t_max")):
        check_input(t, n, F32)
    n_rays = rays_o.size(0)
    rays_pts = torch.empty([n_rays, N_samples, 3], dtype=F32, device=rays_o.device)
    call("fgs_sample_bg_pts", ptr(rays_o), ptr(rays_d), ptr(t_max), float(bg_preserve), int(N_samples), n_rays,
         ptr(rays_pts), stream())
    return rays_pts


def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    check_input(world, "world", BOOL)
    check_input(xyz, "xyz", F32)
    check_input(xyz2ijk_scale, "xyz2ijk_scale", F32)
    check_input(xyz2ijk_shift, "xyz2ijk_shift", F32)
    if world.dim() != 3 or xyz.dim() != 2 or xyz.size(1) != 3:
        raise RuntimeError("maskcache_lookup: world must be [X,Y,Z] and xyz [M,3]")
    n_pts = xyz.size(0)
    out = torch.empty([n_pts], dtype=BOOL, device=xyz.device)
    call("fgs_maskcache_lookup", ptr(world), ptr(xyz), ptr(xyz2ijk_scale), ptr(xyz2ijk_shift), world.size(0),
         world.size(1), world.size(2), n_pts, ptr(out), stream())
    return out


def raw2alpha(density, shift, interval):
    check_input(density, "density", F32)
    if density.dim() != 1:
        raise RuntimeError("raw2alpha: density must be 1-D")
    exp_d, alpha = torch.empty_like(density), torch.empty_like(density)
    call("fgs_raw2alpha", ptr(density), float(shift), float(interval), None, density.size(0), ptr(exp_d), ptr(alpha), stream())
    return [exp_d, alpha]


def raw2alpha_nonuni(density, shift, interval):
    check_input(density, "density", F32)
    check_input(interval, "interval", F32)
    if density.dim() != 1 or interval.shape != density.shape:
        raise RuntimeError("raw2alpha_nonuni: density and interval must be 1-D of equal length")
    exp_d, alpha = torch.empty_like(density), torch.empty_like(density)
    call("fgs_raw2alpha", ptr(density), float(shift), 0.0, ptr(interval), density.size(0), ptr(exp_d), ptr(alpha), stream())
    return [exp_d, alpha]


def raw2alpha_backward(exp_d, grad_back, interval):
    check_input(exp_d, "exp", F32)
    check_input(grad_back, "grad_back", F32)
    grad = torch.empty_like(exp_d)
    call("fgs_raw2alpha_bwd", ptr(exp_d), ptr(grad_back), float(interval), None, exp_d.size(0), ptr(grad), stream())
    return grad


def raw2alpha_nonuni_backward(exp_d, grad_back, interval):
    check_input(exp_d, "exp", F32)
    check_input(grad_back, "grad_back", F32)
    check_input(interval, "interval", F32)
    grad = torch.empty_like(exp_d)
    call("fgs_raw2alpha_bwd", ptr(exp_d), ptr(grad_back), 0.0, ptr(interval), exp_d.size(0), ptr(grad), stream())
    return grad


def alpha2weight(alpha, ray_id, n_rays):
    check_input(alpha, "alpha", F32)
    check_input(ray_id, "ray_id", I64)
    if alpha.dim() != 1 or ray_id.dim() != 1 or alpha.size(0) != ray_id.size(0):
        raise RuntimeError("alpha2weight: alpha and ray_id must be 1-D of equal length")
    n_pts, dev = alpha.size(0), alpha.device
    weight, T = torch.empty_like(alpha), torch.empty_like(alpha)
    alphainv_last = torch.empty([n_rays], dtype=F32, device=dev)
    i_start = torch.empty([n_rays], dtype=I64, device=dev)
    i_end = torch.empty([n_rays], dtype=I64, device=dev)
    call("fgs_alpha2weight_fwd", ptr(alpha), ptr(ray_id), n_pts, int(n_rays), ptr(weight), ptr(T), ptr(alphainv_last),
         ptr(i_start), ptr(i_end), stream())
    return [weight, T, alphainv_last, i_start, i_end]


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last):
    for t, n in ((alpha, "alpha"), (weight, "weight"), (T, "T"), (alphainv_last, "alphainv_last"),
                 (grad_weights, "grad_weights"), (grad_last, "grad_last")):
        check_input(t, n, F32)
    check_input(i_start, "i_start", I64)
    check_input(i_end, "i_end", I64)
    grad = torch.empty_like(alpha)
    call("fgs_alpha2weight_bwd", ptr(alpha), ptr(weight), ptr(T), ptr(alphainv_last), ptr(i_start), ptr(i_end),
         alpha.size(0), int(n_rays), ptr(grad_weights), ptr(grad_last), ptr(grad), stream())
    return grad


render_utils_cuda = SimpleNamespace(
    infer_t_minmax=infer_t_minmax, infer_n_samples=infer_n_samples, infer_ray_start_dir=infer_ray_start_dir,
    sample_pts_on_rays=sample_pts_on_rays, sample_ndc_pts_on_rays=sample_ndc_pts_on_rays,
    sample_bg_pts_on_rays=sample_bg_pts_on_rays, maskcache_lookup=maskcache_lookup,
    raw2alpha=raw2alpha, raw2alpha_backward=raw2alpha_backward,
    raw2alpha_nonuni=raw2alpha_nonuni, raw2alpha_nonuni_backward=raw2alpha_nonuni_backward,
    alpha2weight=alpha2weight, alpha2weight_backward=alpha2weight_backward)


# --------------------------------------------------------------------------- total_variation_cuda

def _tv(param, grad, mask, wx, wy, wz, dense_mode):
    if not (param.is_cuda and grad.is_cuda):
        raise RuntimeError("total_variation_add_grad: param and grad must be CUDA tensors")
    C, X, Y, Z, sC, sX, sY, sZ = grid_strides(param)
    if grad.shape != param.shape or grad.stride() != param.stride():
        raise RuntimeError("total_variation_add_grad: grad must share param's shape and memory layout")
    if mask is not None:
        if mask.dtype != F32 or mask.shape != param.shape or mask.stride() != param.stride():
            raise RuntimeError("total_variation_add_grad_new: mask must be float32 with param's shape and layout")
    call("fgs_tv_add_grad", ptr(param), ptr(grad), ptr(mask), float(wx), float(wy), float(wz), int(bool(dense_mode)),
         C, X, Y, Z, sC, sX, sY, sZ, stream())


def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode):
    _tv(param, grad, None, wx, wy, wz, dense_mode)


def total_variation_add_grad_new(param, grad, mask, wx, wy, wz, dense_mode):
    _tv(param, grad, mask, wx, wy, wz, dense_mode)


total_variation_cuda = SimpleNamespace(total_variation_add_grad=total_variation_add_grad,
                                       total_variation_add_grad_new=total_variation_add_grad_new)


# ---------------------------------------------------------------------------------- adam_upd_cuda

def _adam(mode, param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps):
    tensors = [(param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")]
    if perlr is not None:
        tensors.append((perlr, "perlr"))
    for t, n in tensors:
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
        if t.dtype != F32:
            raise RuntimeError(f"{n} must be float32")
        if t.shape != param.shape or t.stride() != param.stride():
            raise RuntimeError(f"{n} must share param's shape and memory layout")
    if not (param.is_contiguous() or (param.dim() == 5 and param.is_contiguous(memory_format=torch.channels_last_3d))):
        raise RuntimeError("param must be contiguous")
    call("fgs_adam_upd", ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), ptr(perlr), param.numel(), int(step),
         float(beta1), float(beta2), float(lr), float(eps), mode, stream())


def adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    _adam(0, param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps)


def masked_adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    _adam(1, param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps)


def adam_upd_with_perlr(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps):
    _adam(2, param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps)


adam_upd_cuda = SimpleNamespace(adam_upd=adam_upd, masked_adam_upd=masked_adam_upd, adam_upd_with_perlr=adam_upd_with_perlr)


# ------------------------------------------------------------------------------------- trilerp

def trilerp_fwd(grid, pts, xyz_min, xyz_max):
    """grid [1,C,X,Y,Z] (either dense layout), pts [M,3] world coords -> [M,C]."""
    check_input(pts, "xyz", F32)
    check_input(xyz_min, "xyz_min", F32)
    check_input(xyz_max, "xyz_max", F32)
    if not grid.is_cuda or grid.dtype != F32:
        raise RuntimeError("grid must be a float32 CUDA tensor")
    C, X, Y, Z, sC, sX, sY, sZ = grid_strides(grid)
    M = pts.size(0)
    out = torch.empty([M, C], dtype=F32, device=pts.device)
    call("fgs_trilerp_fwd", ptr(grid), C, X, Y, Z, sC, sX, sY, sZ, ptr(xyz_min), ptr(xyz_max), ptr(pts), M, ptr(out), stream())
    return out


def trilerp_bwd(grad_grid, pts, xyz_min, xyz_max, grad_out):
    """Accumulate the scatter-add of grad_out [M,C] into grad_grid (layout of the forward grid)."""
    check_input(pts, "xyz", F32)
    check_input(grad_out, "grad_out", F32)
    C, X, Y, Z, sC, sX, sY, sZ = grid_strides(grad_grid)
    M = pts.size(0)
    call("fgs_trilerp_bwd", ptr(grad_grid), C, X, Y, Z, sC, sX, sY, sZ, ptr(xyz_min), ptr(xyz_max), ptr(pts), M,
         ptr(grad_out), stream())
    return grad_grid


def _displace_array(displace):
    import ctypes
    vals = [float(d) for d in displace]
    return (ctypes.c_float * len(vals))(*vals), len(vals)


def _sdf_grid_dims(grid):
    if grid.dim() != 5 or grid.shape[0] != 1 or grid.shape[1] != 1 or not grid.is_contiguous():
        raise RuntimeError("sdf taps need a contiguous [1,1,X,Y,Z] grid")
    if not grid.is_cuda or grid.dtype != F32:
        raise RuntimeError("grid must be a float32 CUDA tensor")
    return grid.shape[2], grid.shape[3], grid.shape[4]


def sdf_taps_fwd(grid, pts, xyz_min, xyz_max, displace, want_diff=True):
    """nerf.sample_sdfs lookups (model/nerf.py:597-624): feat [M,6K] and clamped index distances diff [M,3K]."""
    check_input(pts, "xyz", F32)
    X, Y, Z = _sdf_grid_dims(grid)
    arr, K = _displace_array(displace)
    M = pts.size(0)
    feat = torch.empty([M, 6 * K], dtype=F32, device=pts.device)
    diff = torch.empty([M, 3 * K], dtype=F32, device=pts.device) if want_diff else None
    call("fgs_sdf_taps_fwd", ptr(grid), X, Y, Z, ptr(xyz_min), ptr(xyz_max), ptr(pts), M, arr, K, ptr(feat), ptr(diff), stream())
    return feat, diff


def sdf_taps_bwd(grad_grid, pts, xyz_min, xyz_max, displace, grad_feat):
    check_input(pts, "xyz", F32)
    check_input(grad_feat, "grad_feat", F32)
    X, Y, Z = _sdf_grid_dims(grad_grid)
    arr, K = _displace_array(displace)
    call("fgs_sdf_taps_bwd", ptr(grad_grid), X, Y, Z, ptr(xyz_min), ptr(xyz_max), ptr(pts), pts.size(0), arr, K,
         ptr(grad_feat), stream())
    return grad_grid
