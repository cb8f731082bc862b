"""Ray-dependent loss terms of one training iteration, as model/nerf_training.py:308-327 computes them from the
render result dict.  Plain torch on whatever device the results live on (host-side training-loop code, not a kernel)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def render_losses(res, target, cfg, model=None):
    loss = cfg.get('weight_main', 1.0) * F.mse_loss(res['rgb_marched'], target)
    if cfg.get('weight_rgbper', 0) > 0:
        rgbper = (res['raw_rgb'] - target[res['ray_id']]).pow(2).sum(-1)
        loss = loss + cfg['weight_rgbper'] * (rgbper * res['weights'].detach()).sum() / len(target)
    if cfg.get('weight_entropy_last', 0) > 0:
        # `[..., -1]` on the 1-D [N] tensor picks ONE ray: reference quirk (nerf_training.py:317), kept
        pout = res['alphainv_cum'][..., -1].clamp(1e-6, 1 - 1e-6)
        loss = loss + cfg['weight_entropy_last'] * (-(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean())
    if cfg.get('weight_orientation', 0) > 0:
        w = res['weights'].detach()
        n_dot_v = (res['normal'] * (-res['viewdirs'])).sum(dim=-1)
        zero = torch.zeros((), dtype=torch.float32, device=n_dot_v.device)
        loss = loss + cfg['weight_orientation'] * torch.mean((w * torch.fmin(zero, n_dot_v) ** 2).sum(dim=-1))
    if cfg.get('sigmoid_rgb_loss', 0) > 0:
        loss = loss + cfg['sigmoid_rgb_loss'] * F.mse_loss(res['sigmoid_rgb'], target)
    return loss
