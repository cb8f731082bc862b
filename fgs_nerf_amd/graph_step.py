"""A whole fine-stage training iteration as ONE hipGraph replay (the step with no host round trip).

The reference's iteration (model/nerf_training.py:237-300,374-385) touches the host about a dozen times; the fused path of
this build still read the survivor count back once per step and spent ~2 ms of Python per 2.3 ms step.  Here nothing of a
step's data ever reaches the host:

* the survivor count M_s stays in device memory; result tensors, activations and gradients are allocated once for a fixed
  CAPACITY of rows and every kernel behind the C ABI clamps to the device-side count (fgs_set_row_count_ptr);
* the per-iteration scalars -- Adam's step size of every parameter group, NeuS 1/s -- are rows of a table the host fills once
  per stage with the same float arithmetic the per-call entry points use (fgs_adam_step_size; the reference's s_val formula);
  a one-wave kernel copies this iteration's row into device scalars and advances a device counter (fgs_step_scalars_tick);
* with no host-visible value left in the step, forward + losses + backward + TV + MaskedAdam are captured in a hipGraph
  (torch.cuda.CUDAGraph: PyTorch provides the capture plumbing and the private memory pool; every node is one of this
  library's kernels or a memset) and an iteration is: one device copy of the (packed) ray batch + one graph launch.
* a survivor count above the capacity cannot corrupt anything: a guard kernel flags it, the optimizer kernels skip the update
  of that step, and the host learns about it at its next `check()` (one read, at a logging interval) and can redo the batch
  eagerly or re-capture with a larger capacity.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Dict, Optional, Sequence

import numpy as np
import torch

from . import fused
from ._lib import call, lib, ptr, stream
from .losses import fused_render_losses


class CapturedFineStep:
    """One captured iteration (fine stage, or a coarse stage: same kernels behind the same device-side row count).

    model, optimizer : a fused-path `nerf` model (stage 'fine') and its MaskedAdam
    loss_cfg         : the loss weights (config keys of model/nerf_training.py:308-327)
    render_kwargs    : near / far / bg / stepsize ... as passed to model.forward
    n_rays           : rays per step (static)
    n_iters          : rows of the schedule table (iterations this stage may run; the last row repeats beyond it)
    global_step_of   : iteration index (0-based) -> global_step handed to the render (s_val schedule, model/nerf.py:514)
    lr_of            : (iteration index, param-group dict) -> learning rate used by that iteration's Adam update
    tv               : None, or (weight, dense) for model.sdf_total_variation_add_grad after the backward pass
    capacity         : rows the survivor buffers hold (see fused.set_sync_free)
    variants         : None, or a list of dicts {'tv': ..., 'extra_loss': callable(model) -> scalar tensor or None}: one graph
                       per entry over the SAME static inputs, schedule table and counters, chosen per iteration by
                       `replay(batch, variant=k)` -- iterations of different SHAPE inside one window (the shipped fine config
                       runs the TV add-grad and the autograd smooth-gradient TV term every third iteration,
                       model/nerf_training.py:330-371).  `tv` is variant 0 when `variants` is None.
    """

    def __init__(self, model, optimizer, loss_cfg: Dict, render_kwargs: Dict, n_rays: int, n_iters: int,
                 global_step_of: Callable[[int], int], lr_of: Callable[[int, Dict], float], tv=None,
                 capacity: int = 131072, variants=None):
        coarse = getattr(model, 'stage', 'fine') in ('coarse', 'geometry_searching')
        if not (fused.supports_coarse(model) if coarse else fused.supports(model)):
            raise RuntimeError("CapturedFineStep needs a model the fused path covers")
        self.model, self.opt, self.loss_cfg, self.kw = model, optimizer, dict(loss_cfg), dict(render_kwargs)
        self.variants = [dict(v) for v in variants] if variants else [dict(tv=tv, extra_loss=None)]
        self.n_rays, self.capacity = int(n_rays), int(capacity)
        dev = model.sdf.grid.device
        self.dev = dev
        # ---- schedule table: column 0 = inv_s, column 1 + g = Adam step size of param group g, last column = s_val itself
        groups = optimizer.param_groups
        optimizer.ensure_state()
        base_step = [max([optimizer.state[p]['step'] for p in g['params'] if p in optimizer.state] or [0]) for g in groups]
        step_size = lib().fgs_adam_step_size
        table = np.zeros((n_iters, 2 + len(groups)), dtype=np.float32)
        for it in range(n_iters):
            s_val = model._s_val_for(global_step_of(it), True)
            table[it, 0] = np.float32(1.0) / np.float32(s_val)          # model/nerf.py:522: ones(1) / s_val in float32
            table[it, -1] = np.float32(s_val)                           # model/nerf.py:520: the s_val parameter's new value
            for gi, g in enumerate(groups):
                b1, b2 = g['betas']
                table[it, 1 + gi] = step_size(base_step[gi] + it + 1, float(b1), float(b2), float(lr_of(it, g)))
        self.table = torch.from_numpy(table).to(dev)
        self.n_iters, self.n_cols = n_iters, table.shape[1]
        self.scalars = torch.zeros(self.n_cols, dtype=torch.float32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.iteration = 0
        # ---- static inputs: one buffer, so that a batch that arrives packed as [4, n_rays, 3] is ONE copy
        self.inputs = torch.zeros(4, n_rays, 3, device=dev)
        self.rays_o, self.rays_d, self.viewdirs, self.target = self.inputs.unbind(0)
        self._seed = torch.ones((), dtype=torch.float32, device=dev)
        self.graphs = [None] * len(self.variants)
        self.losses = [None] * len(self.variants)

    # ------------------------------------------------------------------------------------------------ pieces
    def _enter(self):
        fused.set_sync_free(self.model, self.capacity, inv_s_dev=self.scalars[0:1])
        st = self.model._fused_cache['sync_free']
        self.opt.use_device_schedule({gi: self.scalars[1 + gi:2 + gi].data_ptr() for gi in range(len(self.opt.param_groups))},
                                     skip_ptr=st['flags'][1:2].data_ptr())

    def _leave(self):
        fused.set_sync_free(self.model, None)
        self.opt.use_device_schedule(None)

    @property
    def graph(self):
        return self.graphs[0]

    @property
    def loss(self):
        return self.losses[0]

    def _body(self, update: bool, variant: int = 0):
        # (the warm-up pass ticks too, so that it renders with a real 1/s; capture() rewinds the counter.)  The tick also
        # writes this iteration's s_val into the model's parameter (model/nerf.py:520 refreshes it in every forward).
        call("fgs_step_scalars_tick", ptr(self.table), self.n_iters, self.n_cols, ptr(self.counter), ptr(self.scalars),
             self.n_cols - 1, ptr(self.model.s_val.data), stream())
        # (global_step only selects the training branch here: 1/s comes from the device scalars)
        res = self.model(self.rays_o, self.rays_d, self.viewdirs, global_step=1, **self.kw)
        loss = fused_render_losses(res, self.target, self.loss_cfg, self.model)
        var = self.variants[variant]
        if var.get('extra_loss') is not None:
            loss = loss + var['extra_loss'](self.model)
        self.opt.zero_grad(set_to_none=True)
        # An update issued from inside the backward pass (fused.enable_early_update: k0's Adam pass) belongs to the update: the
        # warm-up pass (update=False) must not apply it, and no record of an earlier pass may make this one skip it.
        cache = self.model.__dict__.setdefault('_fused_cache', {})
        hook = None if update else cache.pop('opt_hook', None)
        if hasattr(self.opt, '_early'):
            self.opt._early = {}
        try:
            loss.backward(self._seed)         # (d loss / d loss given: autograd would launch a ones_like fill per step)
        finally:
            if hook is not None:
                cache['opt_hook'] = hook
        if update:
            if var.get('tv') is not None:
                self.model.sdf_total_variation_add_grad(var['tv'][0], var['tv'][1])
            self.opt.step()
        return loss

    def _drop_autograd_leftovers(self) -> None:
        self.model.gradient = None                 # rebuilt by (coarse) or on first read after (fine) the next forward

    def capture(self, batch: Sequence[torch.Tensor]) -> None:
        """Warm up (one eager forward + backward in the sync-free form on `batch`, no update: allocator pools, cached host
        copies of the geometry) and capture the step."""
        self.load(batch)
        self._enter()
        try:
            # A leaf's AccumulateGrad node lives as long as any autograd graph that reaches it, and it remembers the stream it
            # was created on; backward synchronises with that stream.  The coarse stages keep their gradient volume (a node
            # over sdf.grid) on the model between iterations: after eager iterations that node sits on the DEFAULT stream,
            # and a capture whose backward touches the default stream dies inside hipStreamEndCapture.  Drop the old graph
            # before the warm-up (its nodes are then created on the warm-up's stream) and again before the capture.
            for k in range(len(self.variants)):
                self.losses[k] = None
                self._drop_autograd_leftovers()
                side = torch.cuda.Stream(device=self.dev)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._body(update=False, variant=k)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self.opt.zero_grad(set_to_none=True)
                self._drop_autograd_leftovers()
                fused.reset_grid_grad(self.model)  # the warm-up's k0.grad was not consumed: the captured step starts clean
                self.graphs[k] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graphs[k]):
                    # (detached: the scalar lives in the graph's pool either way, and a loss that kept its autograd graph
                    # -- the leaves' AccumulateGrad nodes, bound to THIS capture's stream -- would reach into the next capture)
                    self.losses[k] = self._body(update=True, variant=k).detach()
                self._drop_autograd_leftovers()    # (the next variant's warm-up must not find this capture's autograd nodes)
        finally:
            self._leave()
        # the capture pass itself launches nothing; schedule and counters start from a clean state
        self.counter.zero_()
        self.clear_counters()

    # ------------------------------------------------------------------------------------------------ per iteration
    def load(self, batch) -> None:
        """batch: (rays_o, rays_d, viewdirs, target), each [n_rays, 3] -- or the four stacked as one [4, n_rays, 3] tensor."""
        if torch.is_tensor(batch):
            self.inputs.copy_(batch, non_blocking=True)
            return
        ro, rd, vd, target = batch
        self.rays_o.copy_(ro, non_blocking=True)
        self.rays_d.copy_(rd, non_blocking=True)
        self.viewdirs.copy_(vd, non_blocking=True)
        self.target.copy_(target, non_blocking=True)

    def replay(self, batch: Optional[Sequence[torch.Tensor]] = None, variant: int = 0) -> torch.Tensor:
        """One training iteration: (optionally) copy the batch into the static inputs, launch the graph (of `variant`).
        Returns the device scalar holding this iteration's loss (overwritten by the next replay of the same variant)."""
        if batch is not None:
            self.load(batch)
        self.graphs[variant].replay()
        self.iteration += 1
        for g in self.opt.param_groups:                    # host mirror of the step counters (state_dict, schedules)
            for p in g['params']:
                st = self.opt.state.get(p)
                if st:
                    st['step'] += 1
        return self.losses[variant]

    def clear_counters(self) -> None:
        buf = self.model._fused_cache['sync_free_buffers']
        buf['flags'].zero_()
        buf['total'].zero_()

    def check(self):
        """(overflowed, survivors processed since the last clear) -- one device->host read; not for every step."""
        return fused.sync_free_state(self.model)


CapturedStep = CapturedFineStep       # (the class serves both stages; the fine stage came first)
