"""Drop-in for the reference's ``model/nerf_ray.py`` (see rays.py for the implementation).

Variant behaviour kept from model/nerf_ray.py: ``get_rays`` returns its outputs on the accelerator
when one is present (:9,38), ``get_rays_of_a_view`` has no ``device`` argument (:71-76) and the
mask-cache sampler calls ``model.sample_ray_ori`` (:230-231).
"""
import torch

from . import rays as _r
from .rays import (batch_indices_generator, get_random_poses, get_random_rays, get_rays_np,  # noqa: F401
                   get_training_rays, get_training_rays_flatten, interp, interp3, ndc_rays, slerp)


def _accel():
    return torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def get_rays(H, W, K, c2w, inverse_y, flip_x, flip_y, mode='center'):
    rays_o, rays_d = _r.get_rays(H, W, K, c2w, inverse_y, flip_x, flip_y, mode=mode)
    return rays_o.to(_accel()), rays_d.to(_accel())


def get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, mode='center'):
    return _r.get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, mode=mode, device=_accel())


get_training_rays_in_maskcache_sampling = _r._maskcache_sampler(use_sample_ray_ori=True, rays_fn=get_rays_of_a_view)
