"""Drop-in for the reference's ``model/dvgo_ray.py`` (see rays.py for the implementation).

Variant behaviour kept from model/dvgo_ray.py: ``get_rays_of_a_view`` takes ``device='cpu'`` and moves
its three outputs there (:69-74); the mask-cache sampler calls ``model.sample_ray`` (:228-229).
"""
from . import rays as _r
from .rays import (batch_indices_generator, get_random_poses, get_random_rays, get_rays, get_rays_np,  # noqa: F401
                   get_training_rays, get_training_rays_flatten, interp, interp3, ndc_rays, slerp)


def get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, mode='center', device='cpu'):
    return _r.get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, mode=mode, device=device)


get_training_rays_in_maskcache_sampling = _r._maskcache_sampler(use_sample_ray_ori=False)
