"""Ray-dependent loss terms of one training iteration (model/nerf_training.py:308-327).

``render_losses``        plain torch, exactly the reference statements, on whatever device the results live on.
``fused_render_losses``  the same value and gradients from two small HIP kernels each way (csrc/losses.hip) instead
                         of ~40 autograd launches -- SURVEY.md 8f row f1 (training-loop host overhead).  Needs the
                         result dict of the fused render path (per-ray view directions, survivor ray ids).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F

from ._lib import call, dyn, ptr, stream


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


def _w5(cfg):
    return (ctypes.c_float * 5)(float(cfg.get('weight_main', 1.0)), float(cfg.get('weight_rgbper', 0)),
                                float(cfg.get('weight_entropy_last', 0)), float(cfg.get('weight_orientation', 0)),
                                float(cfg.get('sigmoid_rgb_loss', 0)))


_SCRATCH = {}     # device index -> per-block partial sums of fgs_fine_loss_fwd (first word: its arrival counter).  One per device,
                  # not per stream: a captured step must find the buffer its warm-up pass (another stream) allocated -- an
                  # allocation inside the capture would put a zero fill into every replay.  (Two loss launches of one device in
                  # flight at the same time on different streams would share it: not a pattern of this path.)


_SCRATCH_RETIRED = []     # outgrown buffers are kept, never freed: a hipGraph captured while one of them was current holds its raw
                          # address, and its replays keep writing block sums and the arrival counter there (a few KB each)


def _loss_scratch(dev, need: int) -> torch.Tensor:
    key = dev.index
    t = _SCRATCH.get(key)
    if t is None or t.numel() < need:
        if t is not None:
            _SCRATCH_RETIRED.append(t)
        t = _SCRATCH[key] = torch.zeros(max(need, 2048), dtype=torch.float32, device=dev)
    return t


class _FineLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_marched, sigmoid_rgb, alphainv_cum, normal, raw_rgb, weights, ray_id, ray_viewdirs, target, w5,
                count_ptr=None):
        N, M = rgb_marched.shape[0], weights.shape[0]
        loss = torch.empty((), dtype=torch.float32, device=rgb_marched.device)
        args = (rgb_marched.contiguous(), sigmoid_rgb.contiguous(), target.contiguous(), alphainv_cum.contiguous(),
                weights.contiguous(), normal.contiguous(), raw_rgb.contiguous(), ray_id.contiguous(),
                ray_viewdirs.contiguous())
        # (sync-free results carry the device address of their survivor count: their per-survivor arrays have CAPACITY rows)
        scratch = _loss_scratch(rgb_marched.device, (max(3 * N, M) + 255) // 256 + 1)
        call("fgs_fine_loss_fwd", N, M, *(ptr(a) for a in args), w5, ptr(loss), ptr(scratch), scratch.numel(),
             dyn(row_count=count_ptr), stream())
        ctx.save_for_backward(*args)
        ctx.w5, ctx.count_ptr = w5, count_ptr
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        args = ctx.saved_tensors
        rgb_marched, weights = args[0], args[4]
        N, M, dev = rgb_marched.shape[0], weights.shape[0], rgb_marched.device
        g_rm = torch.empty(N, 3, dtype=torch.float32, device=dev)
        g_sr = torch.empty(N, 3, dtype=torch.float32, device=dev)
        g_last = torch.empty(N, dtype=torch.float32, device=dev)
        g_normal = torch.empty(M, 3, dtype=torch.float32, device=dev)
        g_raw = torch.empty(M, 3, dtype=torch.float32, device=dev) if ctx.w5[1] > 0 else None
        call("fgs_fine_loss_bwd", N, M, *(ptr(a) for a in args), ctx.w5, ptr(grad_out.contiguous()), ptr(g_rm), ptr(g_sr),
             ptr(g_last), ptr(g_normal), ptr(g_raw), dyn(row_count=ctx.count_ptr), stream())
        return g_rm, g_sr, g_last, g_normal, g_raw, None, None, None, None, None, None


UNIT_SEEDS = {}          # data_ptr() -> device scalar known to hold 1.0 (bench.py / CapturedFineStep pass them to loss.backward).
                         # The tensors are KEPT: a freed seed's address would be handed to some other gradient scalar.


def register_unit_seed(t: torch.Tensor) -> torch.Tensor:
    UNIT_SEEDS[t.data_ptr()] = t
    return t


class _FusedLossNode(torch.autograd.Function):
    """The loss a fused forward pass already computed (fgs_fine_render_loss, fused_common.set_loss_spec): forward hands out the
    scalar; backward marks the stash as the gradient source of the forward node and passes d loss / d rgb_marched on as a
    placeholder (the forward node's backward reads the stash, not its incoming gradients).  A seed other than the one the kernel
    assumed scales the stash (small torch launches: not a path of the captured step)."""

    @staticmethod
    def forward(ctx, rgb_marched, stash):
        ctx.stash = stash
        stash['used'] = True
        return stash['loss'].detach()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        st = ctx.stash
        assumed = st['seed_ptr']
        if assumed is not None:
            if grad_out.data_ptr() != assumed:
                raise RuntimeError("fused loss: loss.backward() was given another seed than set_loss_spec announced")
        elif grad_out.data_ptr() not in UNIT_SEEDS:
            # the kernel assumed d total / d loss = 1 and this scalar is not known to be 1: scale what it left
            for k in ('d_out', 'd_w', 'g_normal', 'g_last', 'g_rm'):
                st[k].mul_(grad_out)
        return st['g_rm'], None


def fused_render_losses(res, target, cfg, model=None):
    """Same scalar and gradients as ``render_losses`` for a fused-path result dict (``res['ray_viewdirs']`` [N,3])."""
    rv = res.get('ray_viewdirs') if hasattr(res, 'get') else None
    if rv is None or not res['rgb_marched'].is_cuda:
        return render_losses(res, target, cfg, model)
    fl = res.get('_fused_loss')
    if fl is not None and not fl['used'] and fl['target_ptr'] == target.data_ptr() and fl['w5_key'] == tuple(float(v) for v in _w5(cfg)):
        return _FusedLossNode.apply(res['rgb_marched'], fl)
    return _FineLoss.apply(res['rgb_marched'], res['sigmoid_rgb'], res['alphainv_cum'], res['normal'], res['raw_rgb'],
                           res['weights'], res['ray_id'], rv, target, _w5(cfg), res.get('survivor_count_ptr'))
