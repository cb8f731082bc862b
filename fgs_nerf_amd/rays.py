"""Ray generation and batching behind the reference's ``model/dvgo_ray.py`` / ``model/nerf_ray.py`` surface.

Function names, argument orders and return orders follow model/dvgo_ray.py:8-258 (and its
near-copy model/nerf_ray.py, which differs in device handling and in calling
``model.sample_ray_ori`` inside the mask-cache sampler).  The arithmetic is written so that the fp32
results are bit-identical to the reference's (same elementwise product + last-axis sum for the
camera rotation, same pixel-centre offsets); tests/golden/rays_*.npz, produced by importing the
reference's own dvgo_ray.py, pin that.
"""
from __future__ import annotations

import time

import numpy as np
import torch


def _pixel_grid(H, W, device, mode):
    """Column index `px` and row index `py` as [H,W] float32 maps (model/dvgo_ray.py:9-22)."""
    px = torch.arange(W, dtype=torch.float32, device=device)[None, :].expand(H, W)
    py = torch.arange(H, dtype=torch.float32, device=device)[:, None].expand(H, W)
    if mode == 'lefttop':
        return px, py
    if mode == 'center':
        return px + 0.5, py + 0.5
    if mode == 'random':
        # the reference draws rand_like(i) then rand_like(j) on [H,W] maps: column jitter first, row-major order
        jx = torch.rand(H, W, device=device)
        jy = torch.rand(H, W, device=device)
        return px + jx, py + jy
    raise NotImplementedError(mode)


def get_rays(H, W, K, c2w, inverse_y, flip_x, flip_y, mode='center'):
    """model/dvgo_ray.py:8-36 -> (rays_o [H,W,3], rays_d [H,W,3])."""
    px, py = _pixel_grid(H, W, c2w.device, mode)
    if flip_x:
        px = px.flip((1,))
    if flip_y:
        py = py.flip((0,))
    u = (px - K[0][2]) / K[0][0]
    v = (py - K[1][2]) / K[1][1]
    one = torch.ones_like(u)
    cam = torch.stack([u, v, one], -1) if inverse_y else torch.stack([u, -v, -one], -1)
    rays_d = torch.sum(cam[..., None, :] * c2w[:3, :3], -1)   # R @ dir, as three products + last-axis sum
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays_np(H, W, K, c2w):
    """model/dvgo_ray.py:39-46 (numpy, pixel corners, OpenGL camera)."""
    px, py = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    cam = np.stack([(px - K[0][2]) / K[0][0], -(py - K[1][2]) / K[1][1], -np.ones_like(px)], -1)
    rays_d = np.sum(cam[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, 3], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """model/dvgo_ray.py:49-66: shift origins to the near plane, project to NDC."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    sx, sy = -1. / (W / (2. * focal)), -1. / (H / (2. * focal))
    oz = rays_o[..., 2]
    o = torch.stack([sx * rays_o[..., 0] / oz, sy * rays_o[..., 1] / oz, 1. + 2. * near / oz], -1)
    d = torch.stack([sx * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / oz),
                     sy * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / oz),
                     -2. * near / oz], -1)
    return o, d


def get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, mode='center', device=None):
    """model/dvgo_ray.py:69-74 -> (rays_o, rays_d, viewdirs); `device=None` keeps c2w's device
    (model/nerf_ray.py behaviour), a device moves the outputs (model/dvgo_ray.py behaviour)."""
    rays_o, rays_d = get_rays(H, W, K, c2w, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y, mode=mode)
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
    if device is not None:
        return rays_o.to(device), rays_d.to(device), viewdirs.to(device)
    return rays_o, rays_d, viewdirs


@torch.no_grad()
def get_training_rays(rgb_tr, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """model/dvgo_ray.py:77-99: per-view ray images [V,H,W,3] (all views share H, W, K)."""
    assert len(np.unique(HW, axis=0)) == 1
    assert len(np.unique(Ks.reshape(len(Ks), -1), axis=0)) == 1
    assert len(rgb_tr) == len(train_poses) and len(rgb_tr) == len(Ks) and len(rgb_tr) == len(HW)
    H, W = HW[0]
    K = Ks[0]
    t0 = time.time()
    shape = [len(rgb_tr), H, W, 3]
    rays_o_tr, rays_d_tr, viewdirs_tr = (torch.zeros(shape, device=rgb_tr.device) for _ in range(3))
    for v, c2w in enumerate(train_poses):
        o, d, vd = get_rays_of_a_view(H=H, W=W, K=K, c2w=c2w, ndc=ndc, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
        rays_o_tr[v].copy_(o.to(rgb_tr.device))
        rays_d_tr[v].copy_(d.to(rgb_tr.device))
        viewdirs_tr[v].copy_(vd.to(rgb_tr.device))
    print('get_training_rays: finish (eps time:', time.time() - t0, 'sec)')
    return rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, [1] * len(rgb_tr)


def box_interval(rays_o, rays_d, xyz_min, xyz_max, near, far):
    """Parameter interval [t_in, t_out] of each ray inside the box (slab method, as model/nerf.py:737-741 writes it: a zero
    direction component is replaced by 1e-6; both ends clamped to [near, far]).  t_out <= t_in: the ray misses the box."""
    d = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
    to_hi, to_lo = (xyz_max - rays_o) / d, (xyz_min - rays_o) / d
    t_in = torch.minimum(to_hi, to_lo).amax(-1).clamp(min=near, max=far)
    t_out = torch.maximum(to_hi, to_lo).amin(-1).clamp(min=near, max=far)
    return t_in, t_out


def _flatten_views(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, keep_mask_fn=None, rays_fn=None):
    """Shared body of the two flattening samplers: concatenates (optionally masked) per-view rays.  `rays_fn`: the calling
    module's own get_rays_of_a_view (nerf_ray's returns rays on the accelerator, which the mask filter then runs on)."""
    get_rays_of_a_view = rays_fn or globals()['get_rays_of_a_view']
    assert len(rgb_tr_ori) == len(train_poses) and len(rgb_tr_ori) == len(Ks) and len(rgb_tr_ori) == len(HW)
    dev = rgb_tr_ori[0].device
    N = sum(im.shape[0] * im.shape[1] for im in rgb_tr_ori)
    rgb_tr = torch.zeros([N, 3], device=dev)
    rays_o_tr, rays_d_tr, viewdirs_tr = torch.zeros_like(rgb_tr), torch.zeros_like(rgb_tr), torch.zeros_like(rgb_tr)
    imsz, top = [], 0
    for c2w, img, (H, W), K in zip(train_poses, rgb_tr_ori, HW, Ks):
        assert img.shape[:2] == (H, W)
        o, d, vd = get_rays_of_a_view(H=H, W=W, K=K, c2w=c2w, ndc=ndc, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
        if keep_mask_fn is None:
            n = H * W
            sel = lambda t: t.flatten(0, 1)
        else:
            keep = keep_mask_fn(o, d, img)
            n = int(keep.sum())
            sel = lambda t: t[keep.to(t.device)]
        rgb_tr[top:top + n].copy_(sel(img))
        rays_o_tr[top:top + n].copy_(sel(o).to(dev))
        rays_d_tr[top:top + n].copy_(sel(d).to(dev))
        viewdirs_tr[top:top + n].copy_(sel(vd).to(dev))
        imsz.append(n)
        top += n
    return rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, imsz, top, N


@torch.no_grad()
def get_training_rays_flatten(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """model/dvgo_ray.py:175-205: all pixels of all views as one [R,3] list."""
    t0 = time.time()
    rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, imsz, top, N = _flatten_views(
        rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y)
    assert top == N
    print('get_training_rays_flatten: finish (eps time:', time.time() - t0, 'sec)')
    return rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, imsz


def _maskcache_sampler(use_sample_ray_ori, rays_fn=None):
    @torch.no_grad()
    def get_training_rays_in_maskcache_sampling(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y,
                                                model, render_kwargs):
        """model/dvgo_ray.py:208-248 / model/nerf_ray.py: keep only rays with at least one sample that is inside the
        bbox AND inside the mask cache; 64 image rows at a time."""
        CHUNK = 64
        t0 = time.time()

        def keep_mask(rays_o, rays_d, img):
            dev = img.device
            keep = torch.ones(img.shape[:2], device=dev, dtype=torch.bool)
            for r0 in range(0, img.shape[0], CHUNK):
                if use_sample_ray_ori:
                    pts, mask_outbbox, _ = model.sample_ray_ori(rays_o=rays_o[r0:r0 + CHUNK], rays_d=rays_d[r0:r0 + CHUNK],
                                                                **render_kwargs)
                else:
                    pts, mask_outbbox = model.sample_ray(rays_o=rays_o[r0:r0 + CHUNK], rays_d=rays_d[r0:r0 + CHUNK],
                                                         **render_kwargs)
                mask_outbbox[~mask_outbbox] |= (~model.mask_cache(pts[~mask_outbbox]))
                keep[r0:r0 + CHUNK] &= (~mask_outbbox).any(-1).to(dev)
            return keep.to(rays_o.device)      # indexes the rays (and, moved back, the image) in _flatten_views

        rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr, imsz, top, N = _flatten_views(
            rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, keep_mask_fn=keep_mask, rays_fn=rays_fn)
        print('get_training_rays_in_maskcache_sampling: ratio', top / N)
        print('get_training_rays_in_maskcache_sampling: finish (eps time:', time.time() - t0, 'sec)')
        return rgb_tr[:top], rays_o_tr[:top], rays_d_tr[:top], viewdirs_tr[:top], imsz
    return get_training_rays_in_maskcache_sampling


def slerp(p0, p1, t):
    """model/dvgo_ray.py:101-105 quaternion slerp."""
    omega = np.arccos(np.dot(p0 / np.linalg.norm(p0), p1 / np.linalg.norm(p1)))
    so = np.sin(omega)
    return np.sin((1.0 - t) * omega) / so * p0 + np.sin(t * omega) / so * p1


def interp(pose1, pose2, s):
    """model/dvgo_ray.py:107-127: interpolate two c2w matrices (lerp translation, slerp rotation)."""
    from scipy.spatial.transform import Rotation
    pose1, pose2 = pose1[:3], pose2[:3]
    assert pose1.shape == (3, 4) and pose2.shape == (3, 4)
    C = (1 - s) * pose1[:, -1] + s * pose2[:, -1]
    q = slerp(Rotation.from_matrix(pose1[:, :3]).as_quat(), Rotation.from_matrix(pose2[:, :3]).as_quat(), s)
    R = Rotation.from_quat(q).as_matrix()
    transform = np.concatenate([np.concatenate([R, C[:, None]], axis=-1), [[0, 0, 0, 1]]], axis=0)
    return torch.tensor(transform, dtype=pose1.dtype)


def interp3(pose1, pose2, pose3, s12, s3):
    return interp(interp(pose1, pose2, s12).cpu(), pose3, s3)


@torch.no_grad()
def get_random_poses(train_poses, generate_poses='loaded', n_poses=20):
    """model/dvgo_ray.py:132-149."""
    if generate_poses == 'loaded':
        n_poses = min(n_poses, len(train_poses))
        return train_poses[np.random.choice(len(train_poses), size=n_poses, replace=False)]
    if generate_poses == 'interpolate_train_all':
        assert len(train_poses) >= 3
        poses = torch.zeros([n_poses, 4, 4], device=train_poses.device)
        for k in range(n_poses):
            p1, p2, p3 = train_poses[np.random.choice(len(train_poses), size=3, replace=False)].cpu()
            s12, s3 = np.random.uniform(0, 1, size=2)
            poses[k] = interp3(p1[:3, :4], p2[:3, :4], p3[:3, :4], s12, s3)
        return poses
    raise NotImplementedError(generate_poses)


@torch.no_grad()
def get_random_rays(train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, n_poses=20):
    """model/dvgo_ray.py:151-172."""
    H, W = HW[0]
    K = Ks[0]
    n_poses = min(n_poses, len(train_poses))
    shape = [n_poses, H, W, 3]
    rays_o_rd, rays_d_rd, viewdirs_rd = (torch.zeros(shape, device=train_poses.device) for _ in range(3))
    for v, c2w in enumerate(get_random_poses(train_poses, n_poses=n_poses)):
        o, d, vd = get_rays_of_a_view(H=H, W=W, K=K, c2w=c2w, ndc=ndc, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
        rays_o_rd[v].copy_(o.to(train_poses.device))
        rays_d_rd[v].copy_(d.to(train_poses.device))
        viewdirs_rd[v].copy_(vd.to(train_poses.device))
    return rays_o_rd, rays_d_rd, viewdirs_rd, [1] * n_poses


def batch_indices_generator(N, BS):
    """model/dvgo_ray.py:251-258: endless epochs of CPU numpy permutations, BS indices at a time."""
    order, top = torch.LongTensor(np.random.permutation(N)), 0
    while True:
        if top + BS > N:
            order, top = torch.LongTensor(np.random.permutation(N)), 0
        yield order[top:top + BS]
        top += BS
