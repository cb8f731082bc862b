"""Per-iteration body of the reference's training loop (model/nerf_training.py:237-456) as a sync-free stepper
(SURVEY.md 8f row f1).

The reference loop touches the host many times per iteration: a CPU index batch copied to the GPU, ``psnr.item()``, five
``.cpu().numpy()`` statistics, ``empty_cache``.  Here everything an iteration needs stays on the device:

* ray batches come from a device-resident permutation (same epoch semantics as ``batch_indices_generator``);
* the loss terms are the two-launch HIP loss (`losses.fused_render_losses`) when the model ran the fused path;
* statistics are kept as device scalars and reduced only when `stats()` is called (the reference's print interval);
* the only device->host read left in an iteration is the survivor count inside the fused forward.

What is mirrored from the reference, statement by statement: optimizer construction from ``lrate_*`` keys (:9-37), the
voxel-increment schedule (:288-295), the loss terms (:306-327), the TV schedule in both forms (autograd TV losses
:330-345, TV add-grad after backward :353-371), exponential / cosine LR decay with ``decay_step_module`` (:389-436),
``tv_updates`` / ``s_updates`` / ``smooth_updates`` (:438-456), progressive grid scaling with optimizer re-creation
(:243-253).  Out of scope: dataset loading, logging, checkpoint writing, evaluation.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .adam import MaskedAdam
from .losses import fused_render_losses


class Cfg(dict):
    """dict with attribute access (what the reference gets from mmcv.Config)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v

    def __setattr__(self, k, v):
        self[k] = v


def create_optimizer_or_freeze_model(model, cfg_train, global_step, logger=None):
    """model/nerf_training.py:9-37."""
    cfg_train = Cfg(cfg_train)
    decay_steps = cfg_train.lrate_decay * 1000
    decay_factor = 0.1 ** (global_step / decay_steps)
    groups = []
    for k in cfg_train.keys():
        if not k.startswith('lrate_'):
            continue
        k = k[len('lrate_'):]
        if not hasattr(model, k):
            continue
        param = getattr(model, k)
        if param is None:
            continue
        lr = cfg_train[f'lrate_{k}'] * decay_factor
        if lr > 0:
            if isinstance(param, nn.Module):
                param = param.parameters()
            groups.append({'params': param, 'lr': lr, 'name': k,
                           'skip_zero_grad': (k in cfg_train.get('skip_zero_grad_fields', []))})
        else:
            param.requires_grad = False
    return MaskedAdam(groups, betas=(0.9, 0.99))


def cosine_lr_func(it, warm_up_iters, warm_up_min_ratio, max_steps, const_warm_up=False, min_ratio=0):
    """model/nerf_training.py:397-406."""
    if it < warm_up_iters:
        if not const_warm_up:
            return warm_up_min_ratio + (1 - warm_up_min_ratio) * (it / warm_up_iters)
        return warm_up_min_ratio
    return (1 + math.cos((it - warm_up_iters) / (max_steps - warm_up_iters) * math.pi)) * 0.5 * (1 - min_ratio) + min_ratio


def lr_decay_factor(cfg_train, global_step) -> float:
    """The factor every param group's lr is multiplied by after iteration `global_step` (model/nerf_training.py:389-430)."""
    cfg_train = Cfg(cfg_train)
    if not cfg_train.get('cosine_lr', ''):
        return 0.1 ** (1 / (cfg_train.lrate_decay * 1000))
    c = cfg_train.get('cosine_lr_cfg', {})
    wu, wr = c.get('warm_up_iters', 0), c.get('warm_up_min_ratio', 1.0)
    const, cmin = c.get('const_warm_up', False), c.get('cos_min_ratio', False)
    g_ = global_step - 1
    pre = 1.0 if global_step == 0 else cosine_lr_func(g_ - 1, wu, wr, cfg_train.N_iters, const, cmin)
    pos = cosine_lr_func(g_, wu, wr, cfg_train.N_iters, const, cmin)
    return pos / pre


class DeviceBatchSampler:
    """``batch_indices_generator`` (model/dvgo_ray.py:251-258) with the permutation resident on the device: endless
    epochs, `BS` indices per call, reshuffle when fewer than `BS` remain.  No per-iteration host->device copy."""

    def __init__(self, n: int, batch: int, device, seed: Optional[int] = None):
        self.n, self.batch, self.device = int(n), int(batch), device
        self.gen = torch.Generator(device=device)
        if seed is not None:
            self.gen.manual_seed(int(seed))
        self.order, self.top = torch.randperm(self.n, generator=self.gen, device=device), 0

    def __call__(self) -> torch.Tensor:
        if self.top + self.batch > self.n:
            self.order, self.top = torch.randperm(self.n, generator=self.gen, device=self.device), 0
        sel = self.order[self.top:self.top + self.batch]
        self.top += self.batch
        return sel


class TrainStepper:
    """One stage of ``scene_rep_reconstruction`` from the point where rays are gathered (model/nerf_training.py:151-186)
    to the end of the per-iteration body.  `rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr` are flat [R,3] device tensors."""

    def __init__(self, model, cfg_train, cfg_model, render_kwargs, rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr,
                 stage: str = 'fine', optimizer=None, averager=None, seed: Optional[int] = None,
                 poses_train=None, near=None):
        self.model, self.stage = model, stage
        self.cfg_train, self.cfg_model = Cfg(cfg_train), Cfg(cfg_model or {})
        self.render_kwargs = dict(render_kwargs)
        self.rgb_tr, self.rays_o_tr, self.rays_d_tr, self.viewdirs_tr = rgb_tr, rays_o_tr, rays_d_tr, viewdirs_tr
        self.optimizer = optimizer or create_optimizer_or_freeze_model(model, self.cfg_train, global_step=0)
        self.averager = averager
        self._bind_averager()
        self.poses_train, self.near = poses_train, near
        if self.cfg_train.get('ray_sampler', 'flatten') not in ('flatten', 'in_maskcache', 'random'):
            raise NotImplementedError(self.cfg_train.ray_sampler)
        self.sampler = DeviceBatchSampler(len(rgb_tr), self.cfg_train.N_rand, rgb_tr.device, seed)
        self._rand = torch.Generator(device=rgb_tr.device)
        if seed is not None:
            self._rand.manual_seed(int(seed) + 1)
        ct = self.cfg_train
        if ct.get('voxel_inc', False):   # model/nerf_training.py:196-211
            lo = [ct.x_mid - ct.x_init_ratio * ct.x_mid, ct.y_mid - ct.y_init_ratio * ct.y_mid,
                  ct.z_mid - ct.z_init_ratio * ct.z_mid]
            hi = [ct.x_mid + ct.x_init_ratio * (1 - ct.x_mid), ct.y_mid + ct.y_init_ratio * (1 - ct.y_mid),
                  ct.z_mid + ct.z_init_ratio * (1 - ct.z_mid)]
            self.inc_lower_init, self.inc_upper_init = torch.tensor(lo), torch.tensor(hi)
        self._stat_keys = ('psnr', 'wmax', 'wsum', 'wnonzero', 's_val')
        self._stats = {k: [] for k in self._stat_keys}
        self.last_result = None

    # ------------------------------------------------------------------------------------------------ helpers
    def _tv_active(self, global_step) -> bool:
        ct = self.cfg_train
        return ct.tv_from < global_step < ct.tv_end and global_step % ct.tv_every == 0

    def _select_indices(self) -> torch.Tensor:
        ct = self.cfg_train
        if ct.get('ray_sampler', 'flatten') == 'random':   # flat equivalent of the reference's [B,H,W] triple randint
            return torch.randint(len(self.rgb_tr), [ct.N_rand], generator=self._rand, device=self.rgb_tr.device)
        return self.sampler()

    def _select_rays(self):
        sel = self._select_indices()
        return self.rgb_tr[sel], self.rays_o_tr[sel], self.rays_d_tr[sel], self.viewdirs_tr[sel]

    # ------------------------------------------------------------------------------------------------ one iteration
    def _bind_averager(self) -> None:
        """(Re)connect the gradient exchange to the model's CURRENT parameters and optimizer."""
        av = self.averager
        if av is None:
            # one GPU: k0's Adam pass is issued in place from inside the fused backward pass (beside the weight-gradient
            # launch) whenever nothing else writes into k0.grad between backward and step -- no TV of either form on k0
            from . import fused
            ct = self.cfg_train
            k0_tv = ct.get('weight_tv_k0', 0) > 0
            if hasattr(self.optimizer, 'early_update') and not k0_tv and getattr(self.model, 'k0', None) is not None \
                    and self.model.k0.grid.is_cuda:
                fused.enable_early_update(self.model, self.optimizer, None, inline=True)
            else:
                fused.disable_early_update(self.model)
            return
        av.rebind(self.model.parameters())
        av.attach(self.model)
        av.attach_optimizer(self.optimizer)                  # k0's exchange is waited for when the optimizer reaches k0
        if (av.world_size > 1 or av.force) and self.cfg_train.get('weight_tv_k0', 0) == 0 and hasattr(self.optimizer, 'early_update'):
            from . import fused
            fused.enable_early_update(self.model, self.optimizer, av)   # ... or applied right behind it (no TV on k0)

    def _maybe_rescale(self, global_step: int) -> bool:
        """Progressive growing (model/nerf_training.py:243-253), at the head of iteration `global_step`: new grids (and, on a
        `reset_iter` entry, re-initialised voxels and MLPs), a new optimizer, the exchange re-bound to the new parameters."""
        model, ct = self.model, self.cfg_train
        if global_step not in ct.get('pg_scale', []):
            return False
        model.scale_volume_grid(model.num_voxels * ct.scale_ratio)
        if global_step in ct.get('reset_iter', []):
            model.reset_voxel_and_mlp()
            if self.cfg_model.get('maskout_near_cam_vox', False) and self.poses_train is not None:
                model.maskout_near_cam_vox(self.poses_train[:, :3, 3], self.near)
        self.optimizer = create_optimizer_or_freeze_model(model, ct, global_step=0)
        self._bind_averager()                            # new grids, new optimizer
        return True

    def _update_tables(self):
        return (self.cfg_train.get('decay_step_module', {}), self.cfg_train.get('tv_updates', {}),
                self.cfg_model.get('s_updates', {}), self.cfg_model.get('smooth_updates', {}))

    def _has_table_update(self, global_step: int) -> bool:
        return any((global_step - 1) in t for t in self._update_tables())

    def _apply_table_updates(self, global_step: int) -> None:
        """The entries the reference applies at the END of iteration `global_step` (keyed by global_step - 1,
        model/nerf_training.py:431-456): per-module lr factors, TV-term updates, s_val schedule and smoothing-kernel updates."""
        model, ct, opt = self.model, self.cfg_train, self.optimizer
        g_ = global_step - 1
        dsm, tvu, su, smu = self._update_tables()
        if g_ in dsm:
            for group in opt.param_groups:
                if group['name'] in dsm[g_]:
                    group['lr'] = group['lr'] * dsm[g_][group['name']]
        if g_ in tvu:
            terms = dict(ct.get('tv_terms', {}))
            terms.update(tvu[g_])
            ct['tv_terms'] = terms
        if g_ in su:
            for k, v in su[g_].items():
                setattr(model, k, v)
        if g_ in smu:
            model.init_smooth_conv(**smu[g_])

    def step(self, global_step: int) -> torch.Tensor:
        model, ct = self.model, self.cfg_train
        self._maybe_rescale(global_step)
        opt = self.optimizer
        target, rays_o, rays_d, viewdirs = self._select_rays()
        # voxel increment (:288-295)
        if ct.get('voxel_inc', False):
            if global_step <= ct.inc_steps:
                w = min(global_step * 1.0 / ct.inc_steps, 1.0)
                model.set_inc_mask(self.inc_lower_init - w * self.inc_lower_init,
                                   self.inc_upper_init + w * (1 - self.inc_upper_init))
        else:
            model.unset_inc_mask()
        # render (:301)
        res = model(rays_o, rays_d, viewdirs, global_step=global_step, **self.render_kwargs)
        self.last_result = res
        opt.zero_grad(set_to_none=True)
        # losses (:306-327): HIP loss kernels on the fused path, the same torch expressions otherwise
        loss = fused_render_losses(res, target, ct, model)
        with torch.no_grad():
            mse = F.mse_loss(res['rgb_marched'].detach(), target)
        hinted = False
        if self.averager is not None and hasattr(res, 'get') and res.get('survivor_pts') is not None:
            hinted = True
        tv_now = self._tv_active(global_step)
        tv_terms = Cfg(ct.get('tv_terms', {}))
        ori_tv = bool(ct.get('ori_tv', False))
        if tv_now and ct.get('weight_tv_density', 0) > 0:    # autograd TV terms (:330-345)
            sdf_tv, smooth_grad_tv = tv_terms.get('sdf_tv', 0), tv_terms.get('smooth_grad_tv', 0)
            if smooth_grad_tv > 0:
                loss = model.density_total_variation(sdf_tv=0, smooth_grad_tv=smooth_grad_tv, weight=ct.weight_tv_density, add_to=loss)
            if ori_tv:
                loss = model.density_total_variation(sdf_tv=sdf_tv, smooth_grad_tv=0, weight=ct.weight_tv_density, add_to=loss)
                if ct.get('weight_tv_k0', 0) > 0:
                    loss = loss + ct.weight_tv_k0 * model.k0_total_variation(**ct.get('k0_tv_terms', {}))
                    hinted = False            # k0 receives a dense gradient: the survivor-point occupancy does not cover it
        if self.averager is not None:                # the averager itself refuses a hint while a dense k0 term is active
            self.averager.set_dense_source(model.k0.grid, tv_now and ori_tv and ct.get('weight_tv_density', 0) > 0
                                           and ct.get('weight_tv_k0', 0) > 0)
        if hinted:
            self.averager.hint_touched(model.k0.grid, res['survivor_pts'], model.xyz_min, model.xyz_max)
        loss.backward()
        if self.averager is not None:
            self.averager.average()
        n_rays = len(rays_o) * (self.averager.world_size if self.averager is not None else 1)
        if tv_now and not ori_tv:                            # TV add-grad (:353-371)
            dense = global_step < ct.get('tv_dense_before', 0)
            if ct.get('weight_tv_density', 0) > 0 and tv_terms.get('sdf_tv', 0) > 0:
                model.sdf_total_variation_add_grad(ct.weight_tv_density * tv_terms.sdf_tv / n_rays, dense)
            if ct.get('weight_tv_k0', 0) > 0:
                if self.averager is not None:
                    self.averager.wait_for(model.k0.grid)    # the TV pass writes into k0.grad: its exchange must be done
                model.k0_total_variation_add_grad(ct.weight_tv_k0 / n_rays, dense)
        opt.step()
        # statistics (:374-385), kept on the device
        # (`weights` is the flat [M] list, so the reference's `.max(-1)` / `.sum(-1)` are over all samples of the batch;
        # its per-iteration `render_result['mask'].float().mean()` would force the lazily built mask: not collected)
        with torch.no_grad():
            w = res['weights'].detach()
            st = self._stats
            st['psnr'].append(-10.0 * torch.log10(mse))
            st['wmax'].append(w.max() if w.numel() else w.new_zeros(()))
            st['wsum'].append(w.sum())
            st['wnonzero'].append((w.sum() > 0).float())
            st['s_val'].append(float(res['s_val']) if 's_val' in res else 0.0)      # a host number already
        # schedules (:389-456)
        g_ = global_step - 1
        f = lr_decay_factor(ct, global_step)
        for group in opt.param_groups:
            group['lr'] = group['lr'] * f
        self._apply_table_updates(global_step)
        return loss

    # ------------------------------------------------------------------------------------------------ captured windows
    def run_captured(self, first_step: int, n_steps: int, capacity: Optional[int] = None):
        """Iterations first_step .. first_step + n_steps - 1 of a stage (fine, coarse, geometry_searching) as hipGraph replays
        (graph_step.CapturedFineStep): per iteration a batch gather + one graph launch, nothing read by the host.  Equivalent
        to calling `step()` for each of them, one GPU.  The window is cut -- and captured anew -- wherever an iteration changes
        what the graphs were built from: a `pg_scale` rescale, a `decay_step_module` / `tv_updates` / `s_updates` /
        `smooth_updates` entry.  `ori_tv` (the coarse stages' autograd TV terms, every iteration in the shipped configs) is part of
        the captured iteration, and so is the voxel-increment phase (`voxel_inc`, global_step <= inc_steps: the iteration's mask
        is rebuilt on the device from index bounds in the schedule table).  Only a TV term on k0 is refused.  Iterations with and without the TV schedule active (sdf TV add-grad +
        the autograd smooth-gradient TV term, every `tv_every`-th iteration) are two graphs over the same state.  The learning-rate decay (model/nerf_training.py:389-436) and the NeuS s_val schedule
        become rows of the device-resident table.  Returns (losses [n_steps] device tensor, overflowed: bool); on overflow
        (more survivors than `capacity` in some iteration: that iteration's update was skipped) the caller re-runs with a
        larger capacity or falls back to `step()`.  Per-iteration statistics are not collected in a captured window."""
        # A progressive-growing iteration (pg_scale) changes every shape behind the captured graphs: the window is cut there, the
        # grids are rescaled and the optimizer re-created exactly as step() does at the head of that iteration, and the rest of
        # the window is captured anew (one warm-up pass + one capture per cut: ~1 s, against thousands of iterations between
        # two cuts in the shipped configs: model/nerf_training.py:244-253, config/shiny_blender.py:203-204).
        # The same holds for an iteration with an end-of-iteration table entry (decay_step_module, tv_updates, s_updates,
        # smooth_updates: they change learning rates, loss terms or the schedule the device table was built from): it becomes
        # the LAST iteration of its window, the entry is applied as step() applies it, the next window is captured on the result.
        ct = self.cfg_train
        end = first_step + n_steps
        starts = sorted({first_step} | {g for g in range(first_step + 1, end)
                                        if g in ct.get('pg_scale', []) or self._has_table_update(g - 1)})
        cuts = starts + [end]
        losses, overflow = [], False
        for a, b in zip(cuts[:-1], cuts[1:]):
            rescaled = self._maybe_rescale(a)
            l, o = self._run_captured_window(a, b - a, capacity if (a == first_step and not rescaled) else None)
            losses.append(l)
            overflow = overflow or o
            if self._has_table_update(b - 1):
                self._apply_table_updates(b - 1)
        return (torch.cat(losses) if len(losses) > 1 else losses[0]), overflow

    def _run_captured_window(self, first_step: int, n_steps: int, capacity: Optional[int] = None):
        """One capture, n_steps replays (see run_captured); no shape-changing iteration inside."""
        from . import fused
        from .graph_step import CapturedFineStep
        model, ct, opt = self.model, self.cfg_train, self.optimizer
        steps = range(first_step, first_step + n_steps)
        covered = fused.supports(model) if self.stage == 'fine' else fused.supports_coarse(model)
        if not covered or self.averager is not None:
            raise RuntimeError("run_captured covers the fused paths (fine, coarse, geometry_searching) on one GPU")
        # voxel-increment phase (:286-291, global_step <= inc_steps: a new mask every iteration): the six index bounds of every
        # iteration's mask become columns of the device table, the captured iteration rebuilds the mask in place from them
        # (fgs_box_mask_fill) -- the voxels step()'s set_inc_mask would set, by the same linspace comparison (inc_index_bounds)
        inc_bounds_of = None
        if ct.get('voxel_inc', False):
            def inc_box(g):
                w = min(g * 1.0 / ct.inc_steps, 1.0)
                return (self.inc_lower_init - w * self.inc_lower_init, self.inc_upper_init + w * (1 - self.inc_upper_init))
            if any(g <= ct.inc_steps for g in steps):
                inc_bounds_of = lambda it: model.inc_index_bounds(*inc_box(min(first_step + it, ct.inc_steps)))   # noqa: E731
        else:
            model.unset_inc_mask()
        if any(g in ct.get('pg_scale', []) for g in list(steps)[1:]):
            raise RuntimeError("_run_captured_window: pg_scale inside the window (run_captured cuts windows there)")
        if any(self._has_table_update(g) for g in list(steps)[:-1]):
            raise RuntimeError("_run_captured_window: a table update inside the window (run_captured cuts windows there)")
        # Iterations come in (at most) two shapes: plain ones, and those in which the TV schedule is active -- the sdf TV
        # add-grad after the backward pass and the autograd smooth-gradient TV term added to the loss (:330-371; every
        # `tv_every`-th iteration in the shipped configs).  One captured graph per shape, chosen per iteration.
        tv_terms = Cfg(ct.get('tv_terms', {}))
        tv_flags = [bool(self._tv_active(g)) for g in steps]
        if any(tv_flags) and ct.get('weight_tv_k0', 0) > 0:
            raise RuntimeError("run_captured: TV add-grad on k0 is not part of the captured iteration; use step()")
        dense = {g < ct.get('tv_dense_before', 0) for g, on in zip(steps, tv_flags) if on}
        if len(dense) > 1:
            raise RuntimeError("run_captured: tv_dense_before falls inside the window")
        # the TV schedule of an active iteration, as step() issues it: autograd terms added to the loss (:330-345: the smooth-gradient
        # term always, the sdf TV term under `ori_tv`), the sdf TV add-grad after the backward pass otherwise (:353-371)
        tv, extra = None, None
        ori_tv = bool(ct.get('ori_tv', False))
        if any(tv_flags) and ct.get('weight_tv_density', 0) > 0:
            w_tv, sdf_tv, s_tv = ct.weight_tv_density, tv_terms.get('sdf_tv', 0), tv_terms.get('smooth_grad_tv', 0)
            if sdf_tv > 0 and not ori_tv:
                tv = (w_tv * sdf_tv / ct.N_rand, dense.pop())
            pieces = []
            if s_tv > 0:
                pieces.append(lambda m, acc: m.density_total_variation(sdf_tv=0, smooth_grad_tv=s_tv, weight=w_tv, add_to=acc))
            if ori_tv and sdf_tv > 0:
                pieces.append(lambda m, acc: m.density_total_variation(sdf_tv=sdf_tv, smooth_grad_tv=0, weight=w_tv, add_to=acc))
            if pieces:
                def extra(m, loss, _pieces=tuple(pieces)):      # (the terms add themselves to the loss: no addition launches)
                    for f in _pieces:
                        loss = f(m, loss)
                    return loss
        variants = [dict(tv=None, extra_loss=None)]
        tv_variant = 0
        if tv is not None or extra is not None:
            if all(tv_flags):
                variants = [dict(tv=tv, extra_loss=extra)]
            else:
                variants.append(dict(tv=tv, extra_loss=extra))
                tv_variant = 1
        which = [tv_variant if on else 0 for on in tv_flags]
        # learning rates: iteration i runs with lr_now * prod_{j < i} decay(first_step + j)   (step() decays AFTER its update)
        factors = [1.0]
        for g in steps:
            factors.append(factors[-1] * lr_decay_factor(ct, g))
        base_lr = {id(g): g['lr'] for g in opt.param_groups}
        first = self._select_rays()
        if capacity is None:
            if inc_bounds_of is not None:        # the mask grows through the window: size the buffers for its last iteration
                model.set_inc_mask(*inc_box(min(first_step + n_steps - 1, ct.inc_steps)))
            with torch.no_grad():
                probe = model(first[1], first[2], first[3], global_step=first_step, **self.render_kwargs)
            capacity = (int(probe['weights'].shape[0] * 1.5) + 4095) // 4096 * 4096
        if inc_bounds_of is not None:
            model.set_inc_mask(*inc_box(first_step))     # THE mask object of this window (rewritten in place by every replay)
        cap = CapturedFineStep(model, opt, ct, self.render_kwargs, ct.N_rand, n_iters=n_steps,
                               global_step_of=lambda it: first_step + it,
                               lr_of=lambda it, g: base_lr[id(g)] * factors[it], capacity=capacity, variants=variants,
                               inc_bounds_of=inc_bounds_of)
        batch = (first[1], first[2], first[3], first[0])           # (rays_o, rays_d, viewdirs, target)
        self.last_result = None          # (an earlier step()'s result would keep its autograd graph -- and the leaves'
        cap.capture(batch)               # AccumulateGrad nodes, bound to the default stream -- alive across the capture)
        losses = torch.empty(n_steps, dtype=torch.float32, device=self.rgb_tr.device)
        for i in range(n_steps):
            if i:          # (iteration 0's batch was loaded by capture())
                cap.load_selected(self._select_indices(), self.rays_o_tr, self.rays_d_tr, self.viewdirs_tr, self.rgb_tr)
            loss_i = cap.replay(None if i else batch, variant=which[i]).detach()
            if loss_i.dtype == torch.float32 and loss_i.is_cuda and loss_i.is_contiguous():
                from ._lib import call, ptr, stream
                call("fgs_copy_f32", ptr(loss_i), ptr(losses[i:]), 1, stream())      # (a launch: the blit path costs 5 us + a 5 us gap)
            else:
                losses[i:i + 1].copy_(loss_i.reshape(1))
        for g in opt.param_groups:                                  # the host's copy of the schedule catches up
            g['lr'] = base_lr[id(g)] * factors[-1]
        overflow, _ = cap.check()
        return losses, overflow

    def stats(self, reset: bool = True) -> Dict[str, float]:
        """Means of the per-iteration statistics since the last call (one device->host transfer)."""
        if not self._stats['psnr']:
            return {}
        dev_keys = [k for k in self._stat_keys if k != 's_val']
        stacked = torch.stack([torch.stack(self._stats[k]).float().mean() for k in dev_keys]).cpu()
        out = {k: float(v) for k, v in zip(dev_keys, stacked)}
        out['s_val'] = sum(self._stats['s_val']) / len(self._stats['s_val'])
        if reset:
            self._stats = {k: [] for k in self._stat_keys}
        return out
