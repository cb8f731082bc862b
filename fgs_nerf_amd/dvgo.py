"""The density-grid (DVGO) variant behind the reference's ``model/dvgo.py`` surface (used only with ``--dvgo_init``).

Same constructor arguments, attributes (``density``, ``k0``, ``mask_cache``, ``act_shift``, ``voxel_size_ratio`` ...),
``state_dict`` keys (``density.grid``, ``k0.grid``) and ``forward`` result keys as model/dvgo.py:25-357.  The render
chain is the operator-at-a-time HIP path: packed sampling -> trilinear density -> softplus post-activation ->
alpha2weight with early termination -> sigmoid(k0) colour -> per-ray sums; no MLP.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dvgo_ray, grid as grid_mod, ops
from .grid import trilerp
from .render import Alphas2Weights, segment_sum


class MaskCache(nn.Module):
    """model/dvgo.py:360-387: known free space from a previous DVGO checkpoint's max-pooled density."""

    def __init__(self, path=None, mask_cache_thres=None, ks=3, density=None, kwargs=None):
        super().__init__()
        if path is not None:
            from .nerf import load_checkpoint_file
            st = load_checkpoint_file(path)
            density, kwargs = st['model_state_dict']['density.grid'], st['MaskCache_kwargs']
        self.mask_cache_thres = mask_cache_thres
        self.register_buffer('xyz_min', torch.as_tensor(np.asarray(kwargs['xyz_min']), dtype=torch.float32))
        self.register_buffer('xyz_max', torch.as_tensor(np.asarray(kwargs['xyz_max']), dtype=torch.float32))
        self.register_buffer('density', F.max_pool3d(density.float(), kernel_size=ks, padding=ks // 2, stride=1).contiguous())
        self.act_shift = kwargs['act_shift']
        self.voxel_size_ratio = kwargs['voxel_size_ratio']
        self.nearest = kwargs.get('nearest', False)
        if self.nearest:
            raise NotImplementedError("nearest-voxel mask cache is not used by the reference configs")

    @torch.no_grad()
    def forward(self, xyz):
        shape = xyz.shape[:-1]
        pts = xyz.reshape(-1, 3).to(self.xyz_max.device).contiguous()
        density = ops.trilerp_fwd(self.density, pts, self.xyz_min, self.xyz_max).reshape(*shape)
        alpha = 1 - torch.exp(-F.softplus(density + self.act_shift) * float(self.voxel_size_ratio))
        return alpha >= self.mask_cache_thres


class dvgo(torch.nn.Module):
    def __init__(self, xyz_min, xyz_max, num_voxels=0, num_voxels_base=0, alpha_init=None, nearest=False,
                 mask_cache_path=None, mask_cache_thres=1e-3, fast_color_thres=0, rgbnet_width=128, num_space=1,
                 ref=False, **kwargs):
        super().__init__()
        if nearest:
            raise NotImplementedError("nearest-voxel lookup is not used by any reference config")
        self.register_buffer('xyz_min', torch.Tensor(np.asarray(xyz_min, dtype=np.float32)))
        self.register_buffer('xyz_max', torch.Tensor(np.asarray(xyz_max, dtype=np.float32)))
        self.fast_color_thres = fast_color_thres
        self.nearest = nearest
        self.num_voxels_base = num_voxels_base
        self.voxel_size_base = ((self.xyz_max - self.xyz_min).prod() / self.num_voxels_base).pow(1 / 3)
        self.alpha_init = alpha_init
        self.act_shift = np.log(1 / (1 - alpha_init) - 1)   # density bias so that the initial alpha is alpha_init
        self.ref = True
        self._set_grid_resolution(num_voxels)

        self.density = grid_mod.create_grid('DenseGrid', channels=1, world_size=self.world_size,
                                            xyz_min=self.xyz_min, xyz_max=self.xyz_max)
        gx, gy, gz = (int(w) for w in self.world_size)
        lx, ly, lz = np.mgrid[-1.0:1.0:gx * 1j, -1.0:1.0:gy * 1j, -1.0:1.0:gz * 1j]
        self.density.grid.data = torch.from_numpy((lx ** 2 + ly ** 2 + lz ** 2) ** 0.5 - 1).float()[None, None, ...]
        self.k0_dim = 3
        self.k0 = grid_mod.create_grid('DenseGrid', channels=self.k0_dim, world_size=self.world_size,
                                       xyz_min=self.xyz_min, xyz_max=self.xyz_max)
        self.mask_cache_path, self.mask_cache_thres = mask_cache_path, mask_cache_thres
        self.mask_cache = None
        self.register_buffer('nonempty_mask', None)
        if mask_cache_path is not None and mask_cache_thres:
            self.create_mask_cache(mask_cache_path)
        self.get_rays_of_a_view = dvgo_ray.get_rays_of_a_view

    def create_mask_cache(self, mask_cache_path):
        if not self.mask_cache:
            self.mask_cache = MaskCache(path=mask_cache_path, mask_cache_thres=self.mask_cache_thres).to(self.xyz_min.device)
            self._set_nonempty_mask()

    def _set_grid_resolution(self, num_voxels):
        """model/dvgo.py:101-109."""
        self.num_voxels = num_voxels
        self.voxel_size, self.world_size = grid_mod.resolution_for(self.xyz_min, self.xyz_max, num_voxels)
        self.voxel_size_ratio = self.voxel_size / self.voxel_size_base

    def get_kwargs(self):
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(),
                'num_voxels': self.num_voxels, 'num_voxels_base': self.num_voxels_base, 'alpha_init': self.alpha_init,
                'nearest': self.nearest, 'mask_cache_path': self.mask_cache_path,
                'mask_cache_thres': self.mask_cache_thres, 'fast_color_thres': self.fast_color_thres,
                'act_shift': self.act_shift, 'voxel_size_ratio': self.voxel_size_ratio}

    def get_MaskCache_kwargs(self):
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(),
                'act_shift': self.act_shift, 'voxel_size_ratio': self.voxel_size_ratio, 'nearest': self.nearest}

    def _grid_points(self):
        g = self.density.grid
        return torch.stack(torch.meshgrid(
            torch.linspace(float(self.xyz_min[0]), float(self.xyz_max[0]), g.shape[2]),
            torch.linspace(float(self.xyz_min[1]), float(self.xyz_max[1]), g.shape[3]),
            torch.linspace(float(self.xyz_min[2]), float(self.xyz_max[2]), g.shape[4]), indexing='ij'), -1).to(g.device)

    @torch.no_grad()
    def _set_nonempty_mask(self):
        self.nonempty_mask = self.mask_cache(self._grid_points()).contiguous().reshape(*self.density.grid.shape)
        self.density.grid[~self.nonempty_mask] = -100

    @torch.no_grad()
    def maskout_near_cam_vox(self, cam_o, near):
        pts = self._grid_points()
        nearest = torch.stack([(pts.unsqueeze(-2) - co).pow(2).sum(-1).sqrt().amin(-1)
                               for co in cam_o.to(pts.device).split(100)]).amin(0)
        self.density.grid[nearest[None, None] <= near] = -100

    @torch.no_grad()
    def scale_volume_grid(self, num_voxels):
        self._set_grid_resolution(num_voxels)
        self.density.scale_volume_grid(self.world_size)
        self.k0.scale_volume_grid(self.world_size)
        if self.mask_cache is not None:
            self._set_nonempty_mask()

    def density_total_variation_add_grad(self, weight, dense_mode=True):
        w = weight * self.world_size.max() / 128
        self.density.total_variation_add_grad(w, w, w, dense_mode)

    def k0_total_variation_add_grad(self, weight, dense_mode=True):
        w = weight * self.world_size.max() / 128
        self.k0.total_variation_add_grad(w, w, w, dense_mode)

    def activate_density(self, density, interval=None):
        """model/dvgo.py:225-227: alpha = 1 - exp(-softplus(d + shift) * interval)."""
        interval = interval if interval is not None else self.voxel_size_ratio
        return 1 - torch.exp(-F.softplus(density + self.act_shift) * interval)

    def grid_sampler(self, xyz, *grids):
        """model/dvgo.py:229-245 (trilinear)."""
        shape = xyz.shape[:-1]
        pts = xyz.reshape(-1, 3)
        ret = [trilerp(g, pts, self.xyz_min, self.xyz_max).reshape(*shape, g.shape[1]).squeeze() for g in grids]
        return ret[0] if len(ret) == 1 else ret

    def sample_ray(self, rays_o, rays_d, near, far, stepsize, **render_kwargs):
        """model/dvgo.py:247-269."""
        far = 1e9
        stepdist = float(stepsize * self.voxel_size)
        ray_pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max = ops.render_utils_cuda.sample_pts_on_rays(
            rays_o.contiguous(), rays_d.contiguous(), self.xyz_min, self.xyz_max, near, far, stepdist)
        inb = ~mask_outbbox
        return ray_pts[inb], ray_id[inb], step_id[inb]

    def gradient(self, density=None):
        """model/dvgo.py:271-277: interior central difference of the density grid."""
        g = torch.zeros([1, 3, *self.density.grid.shape[-3:]], device=density.device)
        g[:, 0, 1:-1, :, :] = (density[:, 0, 2:, :, :] - density[:, 0, :-2, :, :]) / 2 / self.voxel_size
        g[:, 1, :, 1:-1, :] = (density[:, 0, :, 2:, :] - density[:, 0, :, :-2, :]) / 2 / self.voxel_size
        g[:, 2, :, :, 1:-1] = (density[:, 0, :, :, 2:] - density[:, 0, :, :, :-2]) / 2 / self.voxel_size
        return g

    def forward(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        """model/dvgo.py:284-357."""
        N = len(rays_o)
        ray_pts, ray_id, step_id = self.sample_ray(rays_o=rays_o, rays_d=rays_d, **render_kwargs)
        interval = render_kwargs['stepsize'] * self.voxel_size_ratio
        if self.mask_cache is not None:
            m = self.mask_cache(ray_pts)
            ray_pts, ray_id, step_id = ray_pts[m], ray_id[m], step_id[m]
        alpha = self.activate_density(self.density(ray_pts), interval)
        mask = None
        if self.fast_color_thres > 0:
            mask = alpha > self.fast_color_thres
            ray_pts, ray_id, step_id, alpha = ray_pts[mask], ray_id[mask], step_id[mask], alpha[mask]
        weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, N)
        if self.fast_color_thres > 0:
            mask = weights > self.fast_color_thres
            weights, alpha, ray_pts = weights[mask], alpha[mask], ray_pts[mask]
            ray_id, step_id = ray_id[mask], step_id[mask]
        rgb = torch.sigmoid(self.k0(ray_pts))
        gradient = self.grid_sampler(ray_pts, self.gradient(self.density.grid)).reshape(-1, 3)
        normals = gradient / (gradient.norm(dim=-1, keepdim=True) + 1e-7)
        rgb_marched = segment_sum(weights.unsqueeze(-1) * rgb, ray_id, N)
        rgb_marched = rgb_marched + alphainv_last.unsqueeze(-1) * render_kwargs['bg']
        normal_marched = segment_sum(weights.unsqueeze(-1) * normals, ray_id, N)
        return {'alphainv_cum': alphainv_last, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_alpha': alpha,
                'raw_rgb': rgb, 'normal_marched': normal_marched, 'ray_id': ray_id, 'mask': mask}


def total_variation(v, mask=None):
    """model/dvgo.py:420-428: the mean over the valid pairs of |v[i+1] - v[i]| per axis, averaged over the three axes
    (model/nerf.py's variant divides the plain sum by mask.sum() instead).  CUDA grids: csrc/tvloss.hip."""
    if v.is_cuda:
        from . import dense
        return dense.grid_tv_loss(v, mask, per_axis_mean=True)
    from .nerf import _pair_tv
    return _pair_tv(v, mask, per_axis_mean=True)
