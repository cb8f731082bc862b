"""A whole fine-stage training iteration as ONE hipGraph replay (the step with no host round trip).

The reference's iteration (model/nerf_training.py:237-300,374-385) touches the host about a dozen times; the fused path of
this build still read the survivor count back once per step and spent ~2 ms of Python per 2.3 ms step.  Here nothing of a
step's data ever reaches the host:

* the survivor count M_s stays in device memory; result tensors, activations and gradients are allocated once for a fixed
  CAPACITY of rows and every kernel behind the C ABI clamps to the device-side count (fgs_dyn_t.row_count);
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
import os
from typing import Callable, Dict, Optional, Sequence

import numpy as np
import torch

from . import fused
from . import fused_ops as fo
from ._lib import call, lib, ptr, stream
from .fused_common import join_pending_side, set_loss_spec
from .losses import fused_render_losses, register_unit_seed


def variant_allows_defer(var) -> bool:
    """k0's deferred Adam pass needs an iteration in which nothing but the trilinear scatter writes k0.grad: no autograd term on the
    grid (an `extra_loss` may hold one -- the shipped fine config's only acts on sdf, but the step cannot know)."""
    return var.get('extra_loss') is None


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
    averager         : None, or a dist.GradAverager (N > 1 ranks, or a forced single-rank group): the gradient exchange becomes
                       part of the captured step -- RCCL collectives as graph nodes, the k0 brick exchange in its
                       device-counted form (GradAverager.use_device_counts: fixed-capacity buffer, the union's brick count
                       stays on the device), so a multi-GPU iteration is one graph launch per rank as well.  Every rank must
                       construct, capture and replay in lockstep (same variants, same order).
    exchange_capacity: bricks the k0 exchange buffer holds (the same number on every rank; default: the averager's
                       `suggested_capacity` from the host-counted exchanges of the warm-up steps, else 1/2 of the grid)
    inc_bounds_of    : None, or iteration index -> the six index bounds of that iteration's voxel-increment mask
                       (`model.inc_index_bounds(lower, upper)`; model/nerf_training.py:286-291): the mask of `model.inc_mask` (set
                       before capture, never replaced afterwards) is rewritten in place by every replay
    variants         : None, or a list of dicts {'tv': ..., 'extra_loss': callable(model, loss) -> loss + extra terms, or None}: one graph
                       per entry over the SAME static inputs, schedule table and counters, chosen per iteration by
                       `replay(batch, variant=k)` -- iterations of different SHAPE inside one window (the shipped fine config
                       runs the TV add-grad and the autograd smooth-gradient TV term every third iteration,
                       model/nerf_training.py:330-371).  `tv` is variant 0 when `variants` is None.
    """

    def __init__(self, model, optimizer, loss_cfg: Dict, render_kwargs: Dict, n_rays: int, n_iters: int,
                 global_step_of: Callable[[int], int], lr_of: Callable[[int, Dict], float], tv=None,
                 capacity: int = 131072, variants=None, averager=None, exchange_capacity: Optional[int] = None,
                 inc_bounds_of: Optional[Callable[[int], Sequence[int]]] = None):
        coarse = getattr(model, 'stage', 'fine') in ('coarse', 'geometry_searching')
        if not (fused.supports_coarse(model) if coarse else fused.supports(model)):
            raise RuntimeError("CapturedFineStep needs a model the fused path covers")
        self.model, self.opt, self.loss_cfg, self.kw = model, optimizer, dict(loss_cfg), dict(render_kwargs)
        self.variants = [dict(v) for v in variants] if variants else [dict(tv=tv, extra_loss=None)]
        self.n_rays, self.capacity = int(n_rays), int(capacity)
        dev = model.sdf.grid.device
        self.dev = dev
        self.averager = averager if (averager is not None and (averager.world_size > 1 or averager.force)) else None
        self.exchange_capacity = None
        if self.averager is not None:
            k0 = model.k0.grid
            cap = exchange_capacity or self.averager.suggested_capacity(k0)
            if cap is None:
                _, _, X, Y, Z = k0.shape
                cap = max(1, (X // 4) * (Y // 4) * (Z // 4) // 2)
            self.exchange_capacity = int(cap)
        # ---- schedule table: column 0 = inv_s, column 1 + g = Adam step size of param group g, last column = s_val itself
        groups = optimizer.param_groups
        optimizer.ensure_state()
        base_step = [max([optimizer.state[p]['step'] for p in g['params'] if p in optimizer.state] or [0]) for g in groups]
        step_size = lib().fgs_adam_step_size
        # voxel-increment phase (model/nerf_training.py:286-291): six more columns, the index bounds of the iteration's mask
        self.inc_col = 1 + len(groups) if inc_bounds_of is not None else None
        table = np.zeros((n_iters, 2 + len(groups) + (6 if inc_bounds_of is not None else 0)), dtype=np.float32)
        for it in range(n_iters):
            s_val = model._s_val_for(global_step_of(it), True)
            table[it, 0] = np.float32(1.0) / np.float32(s_val)          # model/nerf.py:522: ones(1) / s_val in float32
            table[it, -1] = np.float32(s_val)                           # model/nerf.py:520: the s_val parameter's new value
            if inc_bounds_of is not None:
                table[it, self.inc_col:self.inc_col + 6] = np.asarray(inc_bounds_of(it), dtype=np.float32)
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
        self._seed = register_unit_seed(torch.ones((), dtype=torch.float32, device=dev))
        self.graphs = [None] * len(self.variants)
        self.losses = [None] * len(self.variants)
        self._updated = [[] for _ in self.variants]
        self._pinned = None
        self._exchange_state = None
        self._sdf_exchange_state = None
        self.sdf_exchange_capacity = None
        self._inc_mask = None
        # ---- k0's Adam pass one iteration late (one GPU, fine stage; FGS_K0_ADAM_DEFER=1; default 0: in place, inside the backward
        # pass -- MEASURED SLOWER, round 4: 1.702 against 1.686 ms / step on one box.  Without that memory-bound kernel beside it the
        # sdf scatter kernel meets the weight-gradient launch in full swing and takes 285 us instead of 113 -- the branch ends where
        # it ended before -- and the march kernel of the next forward pass takes 74 us instead of 54 beside the deferred pass, plus
        # 11 us for the join: profiles/r04_fine_graph_step_timeline_k0_adam_deferred.txt.  Kept, tested, off.)
        # Inside the backward pass it runs beside the weight-gradient launch (161 us there, 35 alone) at the end of a scatter branch
        # that outlives that launch by ~76 us.  Nothing reads k0 before the NEXT iteration's feature lookup, 76 us into that
        # iteration, and the march kernel in front of it reads only sdf: the pass is issued at the head of the next replay on a side
        # branch that joins in front of the lookup (fused_fine: 'pre_k0_read').  Same arithmetic on the same gradient with the same
        # step size (the tick latches the previous iteration's before it overwrites it): bit-identical parameters once flush()
        # has applied the last iteration's update -- check() / release() do, and so must anybody who reads k0 in between.
        self.defer_k0 = (os.environ.get("FGS_K0_ADAM_DEFER", "0") == "1" and self.averager is None and not coarse)
        self._k0_group = next((gi for gi, g in enumerate(groups) if any(p is model.k0.grid for p in g['params'])), None)
        if self._k0_group is None:
            self.defer_k0 = False
        self._latch_ss = torch.zeros(1, dtype=torch.float32, device=dev)
        self._latch_skip = torch.zeros(1, dtype=torch.int32, device=dev)
        self._defer_stream = None
        self._k0_pending = False
        self._deferred_in_graph = [False] * len(self.variants)

    # ------------------------------------------------------------------------------------------------ pieces
    def _enter(self):
        fused.set_sync_free(self.model, self.capacity, inv_s_dev=self.scalars[0:1])
        st = self.model._fused_cache['sync_free']
        self.opt.use_device_schedule({gi: self.scalars[1 + gi:2 + gi].data_ptr() for gi in range(len(self.opt.param_groups))},
                                     skip_ptr=st['flags'][1:2].data_ptr())
        if self.averager is not None and self._exchange_is_sparse():
            self.averager.use_device_counts(self.model.k0.grid, self.exchange_capacity, guard_flags=st['flags'])
            # (the graphs will hold the addresses of the exchange buffer and of its sticky flag: this reference outlives _leave())
            self._exchange_state = self.averager._static[id(self.model.k0.grid)]
        if self.averager is not None and self._sdf_exchange_is_sparse():
            # the 1-channel sdf gradient, brick-sparse as well (occupancy from the gradient itself).  Its capacity is not known
            # before a backward pass has run: the first warm-up pass of capture() runs with a provisional one (half the grid,
            # the limit above which the exchange would rather be dense) and _size_sdf_exchange() then sets the real one.
            sdf = self.model.sdf.grid
            if self.sdf_exchange_capacity is None:
                _, _, X, Y, Z = sdf.shape
                self.sdf_exchange_capacity = max(1, (X // 4) * (Y // 4) * (Z // 4) // 2)
                self._sdf_capacity_provisional = True
            self.averager.use_device_counts(sdf, self.sdf_exchange_capacity, guard_flags=st['flags'])
            self._sdf_exchange_state = self.averager._static[id(sdf)]

    def _leave(self):
        fused.set_sync_free(self.model, None)
        self.opt.use_device_schedule(None)
        if self.averager is not None:
            self.averager.use_device_counts(self.model.k0.grid, None)
            self.averager.use_device_counts(self.model.sdf.grid, None)

    def _sdf_exchange_is_sparse(self) -> bool:
        sdf = self.model.sdf.grid
        return self.averager._sparse_1ch(sdf) and os.environ.get("FGS_SDF_SPARSE", "1") == "1"

    def _size_sdf_exchange(self) -> None:
        """After the first (eager) warm-up pass: capacity of the sdf exchange = 1.5 x the union's brick count that pass saw,
        rounded to 256 -- the same number on every rank (the count is the all-reduced union's)."""
        if not getattr(self, '_sdf_capacity_provisional', False):
            return
        n = self.averager.last_device_count(self.model.sdf.grid)
        self._sdf_capacity_provisional = False
        if n is None:
            return
        sdf = self.model.sdf.grid
        _, _, X, Y, Z = sdf.shape
        total = (X // 4) * (Y // 4) * (Z // 4)
        self.sdf_exchange_capacity = min(total, (int(1.5 * n) + 255) // 256 * 256)
        st = self.model._fused_cache['sync_free']
        self.averager.use_device_counts(sdf, self.sdf_exchange_capacity, guard_flags=st['flags'])
        self._sdf_exchange_state = self.averager._static[id(sdf)]
        st['flags'].zero_()

    def _exchange_is_sparse(self) -> bool:
        """dist.GradAverager's own (shape-only) predicate for the brick-sparse k0 exchange."""
        k0, av = self.model.k0.grid, self.averager
        _, C, X, Y, Z = k0.shape
        return not (X % 4 or Y % 4 or Z % 4 or C == 1 or k0.numel() < av.sparse_min_numel)

    @property
    def graph(self):
        return self.graphs[0]

    @property
    def loss(self):
        return self.losses[0]

    def _body(self, update: bool, variant: int = 0):
        # (the warm-up pass ticks too, so that it renders with a real 1/s; capture() rewinds the counter.)  The tick also
        # writes this iteration's s_val into the model's parameter (model/nerf.py:520 refreshes it in every forward).
        cache = self.model.__dict__.setdefault('_fused_cache', {})
        flags = cache['sync_free']['flags']
        gi = self._k0_group
        call("fgs_step_scalars_tick2", ptr(self.table), self.n_iters, self.n_cols, ptr(self.counter), ptr(self.scalars),
             self.n_cols - 1, ptr(self.model.s_val.data), 1 + (gi or 0), ptr(self._latch_ss) if gi is not None else None,
             ptr(flags[1:2]), ptr(self._latch_skip), stream())
        fo.stamps_begin_step()            # (bench.py's in-kernel timing of the matrix-core launches, when it is on)
        gb = cache.get('k0_grad')
        head = update and self.defer_k0 and gb is not None and cache.get('opt_hook') is not None
        defer = head and variant_allows_defer(self.variants[variant])
        if head:
            # the previous iteration's k0 update (whichever variant ran it: every graph opens with this pass; nothing pending: a
            # no-op): beside the march kernel, joined in front of the first reader of k0
            if self._defer_stream is None:
                self._defer_stream = torch.cuda.Stream(device=self.dev)
            side, cur = self._defer_stream, torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self.opt.voxel_update_from_buffer(self.model.k0.grid, gb, self._latch_ss.data_ptr(), self._latch_skip.data_ptr())
                done = torch.cuda.Event()
                done.record(side)
            cache['pre_k0_read'] = lambda: torch.cuda.current_stream().wait_event(done)
        if self.inc_col is not None:
            # this iteration's voxel-increment mask, rebuilt in place from the row the tick just copied (the render below reads
            # the same bytes through model.inc_mask)
            im = self.model.inc_mask
            if im is None or im.mask.dtype != torch.bool or not im.mask.is_contiguous():
                raise RuntimeError("CapturedFineStep(inc_bounds_of=...): call model.set_inc_mask(...) before capture")
            if self._inc_mask is None:
                self._inc_mask = im
            elif im is not self._inc_mask:
                raise RuntimeError("CapturedFineStep: model.inc_mask was replaced after the first pass (the graphs hold the "
                                   "address of the old mask)")
            X, Y, Z = (int(v) for v in im.mask.shape)
            call("fgs_box_mask_fill", ptr(im.mask), X, Y, Z, ptr(self.scalars[self.inc_col:self.inc_col + 6]), stream())
        # (global_step only selects the training branch here: 1/s comes from the device scalars)
        set_loss_spec(self.model, self.target, self.loss_cfg)
        res = self.model(self.rays_o, self.rays_d, self.viewdirs, global_step=1, **self.kw)
        loss = fused_render_losses(res, self.target, self.loss_cfg, self.model)
        var = self.variants[variant]
        if var.get('extra_loss') is not None:
            loss = var['extra_loss'](self.model, loss)
        av = self.averager
        if av is not None:
            # the bricks this step's k0 gradient can touch, their union over ranks and the union's brick list -- all on a side
            # stream under the MLP forward / backward, the count never leaving the device (GradAverager.use_device_counts)
            av.hint_touched(self.model.k0.grid, res['survivor_pts'], self.model.xyz_min, self.model.xyz_max,
                            count_ptr=res['survivor_count_ptr'])
        self.opt.zero_grad(set_to_none=True)
        # An update issued from inside the backward pass (fused.enable_early_update: k0's Adam pass) belongs to the update: the
        # warm-up pass (update=False) must not apply it, and no record of an earlier pass may make this one skip it.
        hook = None if update else cache.pop('opt_hook', None)
        inline_hook = None
        if defer:
            # ... and THIS iteration's gradient stays in the persistent buffer for the next replay's head (defer_update() declines
            # when the record does not describe the gradient: the in-place update then runs as before)
            inline_hook = cache['opt_hook']
            k0 = self.model.k0.grid

            def _defer(p, g, _opt=self.opt, _inline=inline_hook, _k0=k0):
                if p is _k0 and _opt.defer_update(p, g):
                    self._deferred_in_graph[variant] = True
                    return True
                return _inline(p, g)
            cache['opt_hook'] = _defer
        after = None
        if av is not None and not update:
            after, av.after_early = av.after_early, None
        if hasattr(self.opt, '_early'):
            self.opt._early = {}
        # one GPU: the weight-gradient branch is joined in front of the MLP's Adam launch (fused_common._DEFER_WGRAD_JOIN)
        defer_join = av is None and update
        if defer_join:
            cache['defer_side_join'] = True
        try:
            loss.backward(self._seed)         # (d loss / d loss given: autograd would launch a ones_like fill per step)
        finally:
            cache.pop('defer_side_join', None)
            if hook is not None:
                cache['opt_hook'] = hook
            if inline_hook is not None:
                cache['opt_hook'] = inline_hook
            cache.pop('pre_k0_read', None)        # (a forward that never reached the join: nothing may outlive this body)
            if after is not None:
                av.after_early = after
        if av is not None:
            av.average()                      # sdf.grad (dense); k0 and the MLP gradients were exchanged inside the backward pass
            if not update:
                av.wait_all()                 # (no optimizer pass will wait for the k0 exchange: join its stream here)
        if update:
            if var.get('tv') is not None:
                self.model.sdf_total_variation_add_grad(var['tv'][0], var['tv'][1])
            prev_hook = getattr(self.opt, 'before_small', None)
            if hasattr(self.opt, 'before_small'):
                self.opt.before_small = lambda: join_pending_side(self.model)
            try:
                self.opt.step()
            finally:
                if hasattr(self.opt, 'before_small'):
                    self.opt.before_small = prev_hook
                join_pending_side(self.model)     # (nothing of the branch may outlive the body: an optimizer without the hook)
        else:
            join_pending_side(self.model)
        return loss

    def _drop_autograd_leftovers(self) -> None:
        self.model.gradient = None                 # rebuilt by (coarse) or on first read after (fine) the next forward

    def capture(self, batch: Sequence[torch.Tensor]) -> None:
        """Warm up (one eager forward + backward in the sync-free form on `batch`, no update: allocator pools, cached host
        copies of the geometry) and capture the step."""
        self.load(batch)
        if hasattr(self.model, '_world_max'):
            self.model._world_max()       # host copy of world_size.max() (the TV weight): read now, not inside the capture -- the
                                          # warm-up pass runs no TV pass, and a grid rescale has just invalidated the cached value
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
                if self.averager is not None and k == 0:
                    self._size_sdf_exchange()
                self.opt.zero_grad(set_to_none=True)
                self._drop_autograd_leftovers()
                fused.reset_grid_grad(self.model)  # the warm-up's k0.grad was not consumed: the captured step starts clean
                self.graphs[k] = torch.cuda.CUDAGraph()
                gkw = {}
                if self.averager is not None:
                    # ProcessGroupNCCL's watchdog THREAD polls the events of earlier (eager) collectives with hipEventQuery; under
                    # the default "global" capture mode that call is illegal from ANY thread while this one captures, and the
                    # watchdog then takes the process down ("operation not permitted when stream is capturing" -> terminate:
                    # seen when the warm-up pass's collectives were still on its list).  Two measures: thread-local capture mode
                    # (only the capturing thread is policed), and the watchdog's list given time to drain (it wakes every 100 ms).
                    gkw['capture_error_mode'] = 'thread_local'
                    import time
                    time.sleep(0.35)
                with torch.cuda.graph(self.graphs[k], **gkw):
                    # (detached: the scalar lives in the graph's pool either way, and a loss that kept its autograd graph
                    # -- the leaves' AccumulateGrad nodes, bound to THIS capture's stream -- would reach into the next capture)
                    self.losses[k] = self._body(update=True, variant=k).detach()
                # the parameters this variant's body updated: those that had a gradient when its optimizer pass ran
                self._updated[k] = [p for g in self.opt.param_groups for p in g['params']
                                    if p.grad is not None]
                self._drop_autograd_leftovers()    # (the next variant's warm-up must not find this capture's autograd nodes)
        finally:
            self._leave()
        # the capture pass itself launches nothing; schedule and counters start from a clean state
        self.counter.zero_()
        self.clear_counters()
        self._pin_static_buffers()

    # ------------------------------------------------------------------------------------------------ per iteration
    def load(self, batch) -> None:
        """batch: (rays_o, rays_d, viewdirs, target), each [n_rays, 3] -- or the four stacked as one [4, n_rays, 3] tensor."""
        if torch.is_tensor(batch):
            if (batch.is_cuda and batch.dtype == torch.float32 and batch.is_contiguous() and batch.numel() == self.inputs.numel()
                    and batch.data_ptr() % 16 == 0):
                call("fgs_copy_f32", ptr(batch), ptr(self.inputs), int(batch.numel()), stream())     # (a launch, not the blit path)
            else:
                self.inputs.copy_(batch, non_blocking=True)
            return
        ro, rd, vd, target = batch
        self.rays_o.copy_(ro, non_blocking=True)
        self.rays_d.copy_(rd, non_blocking=True)
        self.viewdirs.copy_(vd, non_blocking=True)
        self.target.copy_(target, non_blocking=True)

    def load_selected(self, sel: torch.Tensor, rays_o_src, rays_d_src, viewdirs_src, target_src) -> None:
        """The batch `src[sel]` of four flat [R,3] float32 device tensors straight into the static inputs: one launch instead of
        four gathers and four copies (model/nerf_training.py:256-261)."""
        srcs = (rays_o_src, rays_d_src, viewdirs_src, target_src)
        R = int(rays_o_src.shape[0])
        if not (sel.is_cuda and sel.dtype == torch.int64 and sel.is_contiguous() and sel.numel() == self.inputs.shape[1]
                and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (R, 3) for t in srcs)):
            self.load(tuple(t[sel] for t in srcs))
            return
        call("fgs_gather_batch", ptr(sel), int(sel.numel()), R, *(ptr(t) for t in srcs), ptr(self.inputs), stream())

    def replay(self, batch: Optional[Sequence[torch.Tensor]] = None, variant: int = 0) -> torch.Tensor:
        """One training iteration: (optionally) copy the batch into the static inputs, launch the graph (of `variant`).
        Returns the device scalar holding this iteration's loss (overwritten by the next replay of the same variant)."""
        if batch is not None:
            self.load(batch)
        self._check_static_buffers()
        self.graphs[variant].replay()
        self.iteration += 1
        if self._deferred_in_graph[variant]:
            self._k0_pending = True
            gb = self.model.__dict__.get('_fused_cache', {}).get('k0_grad')
            if gb is not None:
                gb['pending'] = self
        for p in self._updated[variant]:                   # host mirror of the step counters (state_dict, schedules): the
            st = self.opt.state.get(p)                     # parameters this variant's captured body updated
            if st:
                st['step'] += 1
        return self.losses[variant]

    def _pin_static_buffers(self) -> None:
        """The graphs hold raw addresses of buffers that live OUTSIDE their memory pool: the persistent k0 gradient buffer and
        its voxel flags (fused._take_grid_grad REPLACES the buffer when an eager step in between still holds the old one as
        p.grad), the packed weight images of the MLP chains, the sync-free counters.  Keep them alive for as long as the
        graphs exist, and remember the addresses that may legitimately change hands."""
        cache = self.model.__dict__.setdefault('_fused_cache', {})
        gb = cache.get('k0_grad')
        self._pinned = dict(k0_grad=None if gb is None else (gb, gb['buf'], gb['flags']),
                            rc_images=dict(fused.fo._RC_IMAGES), sync_free=cache.get('sync_free_buffers'),
                            hints=None if self.averager is None else dict(self.averager._hints))

    def _check_static_buffers(self) -> None:
        pin = self._pinned
        if not pin or pin['k0_grad'] is None:
            return
        gb, buf, flags = pin['k0_grad']
        now = self.model.__dict__.get('_fused_cache', {}).get('k0_grad')
        if now is not gb or gb['buf'] is not buf or gb['flags'] is not flags:
            raise RuntimeError("CapturedFineStep.replay: the model's persistent k0 gradient buffer was replaced after the capture "
                               "(an eager step in between kept k0.grad alive, or the grid changed shape): capture() again")
        if not gb['clean']:
            raise RuntimeError("CapturedFineStep.replay: the persistent k0 gradient buffer holds an unconsumed gradient (an eager "
                               "backward pass without optimizer step): call fused.reset_grid_grad(model) first")

    def flush(self) -> None:
        """Apply the k0 update the last replay left pending (deferred Adam pass; a no-op otherwise).  Anybody who reads k0 -- or
        trains eagerly -- between two replays calls this first; check() and release() do."""
        if not self._k0_pending:
            return
        cache = self.model.__dict__.get('_fused_cache', {})
        gb = cache.get('k0_grad')
        buf = cache.get('sync_free_buffers')
        if gb is not None and buf is not None:
            # (the scalars still hold the last iteration's row -- the tick of the next replay has not run -- and flags[1] its skip flag)
            self.opt.voxel_update_from_buffer(self.model.k0.grid, gb, self.scalars[1 + self._k0_group:2 + self._k0_group].data_ptr(),
                                              buf['flags'][1:2].data_ptr())
            gb['pending'] = None
        self._k0_pending = False

    def release(self) -> None:
        """Drop the captured graphs (and their memory pool).  With an averager, call this BEFORE
        torch.distributed.destroy_process_group(): destroying an RCCL communicator whose collectives are still nodes of a live
        hipGraph aborts the process (seen on ROCm 7.0 / RCCL 2.26: SIGABRT inside destroy_process_group, no message)."""
        self.flush()
        self.graphs = [None] * len(self.variants)
        self.losses = [None] * len(self.variants)
        self._pinned = None
        torch.cuda.synchronize(self.dev)

    def clear_counters(self) -> None:
        buf = self.model._fused_cache['sync_free_buffers']
        buf['flags'].zero_()
        buf['total'].zero_()

    def check(self):
        """(overflowed, survivors processed since the last clear) -- one device->host read; not for every step.  `overflowed`
        covers both capacities: a survivor list that did not fit (that step's update was skipped, on every rank) and, with an
        averager, a k0 exchange that did not fit (`exchange_overflowed()`: every update since then was skipped).  Applies a pending
        k0 update first (flush())."""
        self.flush()
        return fused.sync_free_state(self.model)

    def exchange_overflowed(self) -> bool:
        """The union of touched k0 (or sdf) bricks exceeded its exchange capacity in some replay (one device->host read).  The
        replicas are still identical -- every rank skipped every update from that step on -- but the persistent gradient buffer
        holds leftovers: call fused.reset_grid_grad(model), then capture again with a larger capacity.
        One exception to "every update": k0's Adam pass runs inside the backward pass, behind k0's own exchange, BEFORE the sdf
        exchange's occupancy guard exists (sdf.grad is only final at the end of the backward pass).  In the iteration in which the
        SDF exchange overflows, k0 has therefore already stepped while sdf and the MLPs are skipped: one torn iteration, the same
        on every rank (the union count is all-reduced), after which everything is frozen.  A caller that must not keep it restores
        parameters and optimizer state from its last checkpoint before the logging window in which the flag was raised."""
        return any(bool(int(st['sticky'].cpu()[0])) for st in (self._exchange_state, self._sdf_exchange_state) if st is not None)


CapturedStep = CapturedFineStep       # (the class serves both stages; the fine stage came first)
