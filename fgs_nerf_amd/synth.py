"""Synthetic scene and ray batches for the benchmark and the parity tests.

Everything is generated on the CPU with fixed seeds (seed 777, the reference's ``seed_everything``,
model/utils.py:100-107) and is plain data: the same tensors feed the HIP path and the oracle.

  scene : bbox [-1,1]^3, G^3 voxels; sdf = |p| - radius on the np.mgrid lattice of model/nerf.py:77-82 plus
          N(0, sdf_noise); k0 ~ N(0, 0.1) with k0_dim channels; MLPs with PyTorch's default nn.Linear init.
  rays  : pixels drawn uniformly from `n_views` blender-style cameras on a circle (radius 4, 30 degrees
          elevation, looking at the origin, OpenGL axes), field of view 0.6911 rad, 800x800.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

SEED = 777

# model kwargs of the three shiny_blender stages that matter for shapes (config/shiny_blender.py:80-103,151-178,220-249)
FINE_MODEL = dict(stage='fine', fast_color_thres=1e-4, k0_dim=12, rgbnet_width=256, rgbnet_depth=4, refnet_width=256,
                  refnet_depth=4, posbase_pe=5, viewbase_pe=3, refbase_pe=8, s_ratio=50, s_start=0.05, center_sdf=True,
                  grad_feat=(0.5, 1.0, 1.5, 2.0), sdf_feat=(0.5, 1.0, 1.5, 2.0), ref=True, use_viewdir=True)
COARSE_MODEL = dict(stage='coarse', fast_color_thres=1e-4, k0_dim=12, rgbnet_width=192, rgbnet_depth=3, refnet_width=192,
                    refnet_depth=3, posbase_pe=5, viewbase_pe=1, refbase_pe=5, smooth_ksize=5, smooth_sigma=0.8,
                    s_ratio=50, s_start=0.2, ref=True, use_viewdir=True)
GEOMETRY_MODEL = dict(stage='geometry_searching', fast_color_thres=1e-4, k0_dim=6, refnet_width=128, refnet_depth=3,
                      posbase_pe=5, viewbase_pe=1, refbase_pe=3, smooth_ksize=5, smooth_sigma=0.8, s_ratio=50,
                      s_start=0.2, ref=True, use_viewdir=True)
# fine-stage loss weights (config/shiny_blender.py:180-218)
FINE_LOSS = dict(weight_main=1.0, weight_rgbper=0.0, weight_entropy_last=0.001, weight_orientation=1e-4,
                 sigmoid_rgb_loss=0.02)
COARSE_LOSS = dict(weight_main=1.0, weight_rgbper=0.2, weight_entropy_last=0.001, weight_orientation=1e-4,
                   sigmoid_rgb_loss=0.1)
RENDER_KWARGS = dict(near=2.0, far=6.0, bg=1, stepsize=0.5, inverse_y=False, flip_x=False, flip_y=False)


def look_at_origin(azimuth_deg: float, elevation_deg: float = 30.0, radius: float = 4.0) -> np.ndarray:
    """c2w [4,4] float32 of a camera on a sphere around the origin, OpenGL/blender axes (-z forward, +y up),
    world z up."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    eye = radius * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
    back = eye / np.linalg.norm(eye)
    right = np.cross(np.array([0.0, 0.0, 1.0]), back)
    right /= np.linalg.norm(right)
    up = np.cross(back, right)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, back, eye
    return c2w.astype(np.float32)


def intrinsics(H: int, W: int, fov_x: float = 0.6911) -> np.ndarray:
    """Pinhole K as lib/load_blender.py builds it: focal = 0.5 W / tan(0.5 fov)."""
    focal = 0.5 * W / math.tan(0.5 * fov_x)
    return np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)


def random_rays(n_rays: int, n_views: int = 8, H: int = 800, W: int = 800, seed: int = SEED, inverse_y: bool = False):
    """n_rays pixels drawn uniformly over `n_views` cameras -> (rays_o, rays_d, viewdirs) float32 [n_rays,3] on CPU."""
    from . import rays as rays_mod
    rng = np.random.RandomState(seed)
    view = rng.randint(0, n_views, size=n_rays)
    row = rng.randint(0, H, size=n_rays)
    col = rng.randint(0, W, size=n_rays)
    K = intrinsics(H, W)
    o_all, d_all = torch.empty(n_rays, 3), torch.empty(n_rays, 3)
    for v in range(n_views):
        sel = np.nonzero(view == v)[0]
        if len(sel) == 0:
            continue
        c2w = torch.from_numpy(look_at_origin(v * 360.0 / n_views))
        if inverse_y:  # OpenCV-style camera (DTU): flip the y and z camera axes
            c2w = c2w.clone()
            c2w[:3, 1:3] *= -1
        ro, rd = rays_mod.get_rays(H, W, K, c2w, inverse_y=inverse_y, flip_x=False, flip_y=False, mode='center')
        o_all[sel] = ro[row[sel], col[sel]]
        d_all[sel] = rd[row[sel], col[sel]]
    viewdirs = d_all / d_all.norm(dim=-1, keepdim=True)
    return o_all.contiguous(), d_all.contiguous(), viewdirs.contiguous()


def view_rays(view: int, H: int, W: int, n_views: int = 8):
    """Full frame of one synthetic camera, flattened [H*W,3] (BASELINE config 1: 200x200 render)."""
    from . import rays as rays_mod
    c2w = torch.from_numpy(look_at_origin(view * 360.0 / n_views))
    ro, rd, vd = rays_mod.get_rays_of_a_view(H, W, intrinsics(H, W), c2w, False, inverse_y=False, flip_x=False,
                                             flip_y=False)
    return ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous(), vd.reshape(-1, 3).contiguous()


def scene_tensors(G: int, k0_dim: int = 12, radius: float = 0.6, sdf_noise: float = 0.01, k0_std: float = 0.1,
                  seed: int = SEED) -> Dict[str, torch.Tensor]:
    """sdf [1,1,G,G,G] and k0 [1,k0_dim,G,G,G] (channel-first, contiguous) for the synthetic ball scene."""
    gen = torch.Generator().manual_seed(seed)
    x, y, z = np.mgrid[-1.0:1.0:G * 1j, -1.0:1.0:G * 1j, -1.0:1.0:G * 1j]
    sdf = torch.from_numpy((x ** 2 + y ** 2 + z ** 2) ** 0.5 - radius).float()[None, None]
    sdf = sdf + sdf_noise * torch.randn(sdf.shape, generator=gen)
    k0 = k0_std * torch.randn([1, k0_dim, G, G, G], generator=gen)
    return {'sdf': sdf.contiguous(), 'k0': k0.contiguous()}


def build_model(G: int, model_kwargs: dict = None, seed: int = SEED, device='cpu', **overrides):
    """A ``nerf`` model on `device` holding the synthetic scene; MLP weights = default nn.Linear init under `seed`."""
    from .nerf import nerf
    kw = dict(FINE_MODEL if model_kwargs is None else model_kwargs)
    kw.update(overrides)
    torch.manual_seed(seed)
    model = nerf(xyz_min=[-1., -1., -1.], xyz_max=[1., 1., 1.], num_voxels=G ** 3, num_voxels_base=G ** 3, **kw)
    assert model.world_size.tolist() == [G, G, G], model.world_size
    sc = scene_tensors(G, k0_dim=kw.get('k0_dim', 12), seed=seed)
    with torch.no_grad():
        model.sdf.grid.copy_(sc['sdf'])
        model.k0.grid.copy_(sc['k0'])
    return model.to(device)


def oracle_params(model) -> Dict:
    """Plain-tensor (CPU, channel-first) view of a model for oracle/oracle.py's forward_fine / forward_coarse."""
    from .nerf import mlp_layers

    def cpu(t):
        return t.detach().cpu().contiguous().clone()

    P = dict(xyz_min=cpu(model.xyz_min), xyz_max=cpu(model.xyz_max), voxel_size=model.voxel_size.detach().cpu(),
             sdf=cpu(model.sdf.grid), k0=cpu(model.k0.grid),
             refnet=[(cpu(l.weight), cpu(l.bias)) for l in mlp_layers(model.refnet)],
             rgbnet=None if model.rgbnet is None else [(cpu(l.weight), cpu(l.bias)) for l in mlp_layers(model.rgbnet)],
             posfreq=cpu(model.posfreq), viewfreq=cpu(model.viewfreq), reffreq=cpu(model.reffreq),
             fast_color_thres=model.fast_color_thres, s_ratio=model.s_ratio, s_start=model.s_start,
             grad_feat_displace=tuple(sorted(set(model.grad_feat + model.k_grad_feat))) if model.stage == 'fine' else (),
             use_grad_norm=model.use_grad_norm, center_sdf=model.center_sdf, mask_cache=None, inc_mask=None,
             smooth_kernel=None)
    if model.smooth_sdf:
        P['smooth_kernel'] = cpu(model.smooth_conv.weight)[0, 0]
    P['grad_mode'] = getattr(model, 'grad_mode', 'interpolate')
    if P['grad_mode'] == 'grad_conv':
        P['grad_conv_w'] = cpu(model.grad_conv.weight)
    if model.mask_cache is not None:
        mc = model.mask_cache
        P['mask_cache'] = dict(sdf_mask=cpu(mc.sdf_mask), xyz_min=cpu(mc.xyz_min), xyz_max=cpu(mc.xyz_max),
                               thres=mc.mask_cache_thres)
    if model.inc_mask is not None:
        im = model.inc_mask
        P['inc_mask'] = dict(mask=cpu(im.mask), scale=cpu(im.xyz2ijk_scale), shift=cpu(im.xyz2ijk_shift))
    return P
