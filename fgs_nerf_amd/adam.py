"""``MaskedAdam`` behind the reference's ``model/adam.py`` surface (model/adam.py:167-221).

Contract kept: constructor ``MaskedAdam(params, lr, betas=(0.9, 0.99), eps)``, param-group keys
``lr / betas / eps / skip_zero_grad``, per-parameter state ``step / exp_avg / exp_avg_sq``,
``set_pervoxel_lr(count)``, and the rule that picks one of three fused update kernels per tensor
(per-voxel lr if a same-shape table is set, else masked if the group says ``skip_zero_grad``, else
dense).  The kernels are csrc/gridopt.hip (``fgs_adam_upd``), not the JIT-built ``adam_upd_cuda``.
Small tensors (the 16 MLP weights / biases) are updated by ONE ``fgs_adam_upd_multi`` launch per
(betas, eps) instead of one launch each; the arithmetic per element is identical.
The reference's unused ``Adam`` class (model/adam.py:16-161) is outside the hot path.
"""
from __future__ import annotations

import ctypes

import torch

from ._lib import call, stream
from .ops import adam_upd_cuda

_SMALL = 1 << 20  # tensors below this many elements go through the multi-tensor launch


def _as_layout_of(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """`t` re-laid out with `like`'s strides (grids are stored channel-last; autograd may hand back another layout)."""
    if t.stride() == like.stride():
        return t
    return torch.empty_strided(like.shape, like.stride(), dtype=t.dtype, device=t.device).copy_(t)


class MaskedAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.99), eps=1e-8):
        for ok, what in ((lr >= 0.0, f"Invalid learning rate: {lr}"),
                         (eps >= 0.0, f"Invalid epsilon value: {eps}"),
                         (0.0 <= betas[0] < 1.0, f"Invalid beta parameter at index 0: {betas[0]}"),
                         (0.0 <= betas[1] < 1.0, f"Invalid beta parameter at index 1: {betas[1]}")):
            if not ok:
                raise ValueError(what)
        self.per_lr = None
        self.before_param = None      # optional callable(param), invoked right before a parameter is updated (dist.py)
        self.before_small = None      # optional callable(), invoked before the first MLP tensor is updated (graph_step.py: the
                                      # weight-gradient branch of a captured step is joined there, not inside the backward pass)
        self._early = {}              # id(param) -> event of an update already applied by early_update() in this step
        self._early_stream = None
        self._dev = None              # device-resident schedule (use_device_schedule): {'ss': {group index: ptr}, 'skip': ptr}
        self._skip = None             # skip-flag device pointer of a host-scheduled sync-free step (use_skip_flag)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def use_device_schedule(self, step_size_ptrs=None, skip_ptr=None) -> None:
        """Captured (hipGraph) steps: the step size lr_t * sqrt(1 - b2^t) / (1 - b1^t) of parameter group g is read by the
        update kernels from the device float at `step_size_ptrs[g]` (written each step by fgs_step_scalars_tick from a table
        built with fgs_adam_step_size, the host entry points' own arithmetic), and the whole update is skipped while the
        device int at `skip_ptr` is non-zero.  `step()` then passes no host scalar that changes from step to step, so a
        captured step replays correctly; the host-side `state[p]['step']` counters are advanced by whoever replays.
        `use_device_schedule(None)` returns to host scalars."""
        self._dev = None if step_size_ptrs is None else {'ss': dict(step_size_ptrs), 'skip': skip_ptr}

    def zero_grad(self, set_to_none: bool = True):
        """(also forgets updates recorded by early_update() for a backward pass whose step() never came)"""
        self._early = {}
        return super().zero_grad(set_to_none=set_to_none)

    def ensure_state(self) -> None:
        """Create every parameter's moment buffers now (a captured step must not allocate-and-zero them inside the graph)."""
        for group in self.param_groups:
            for p in group['params']:
                if p.requires_grad:
                    self._state_of(p)

    def set_pervoxel_lr(self, count):
        self.per_lr = count.float() / count.max()

    @staticmethod
    def _check_layout(p, st) -> None:
        if st['exp_avg'].stride() != p.stride() or st['exp_avg_sq'].stride() != p.stride():
            raise RuntimeError("MaskedAdam: exp_avg / exp_avg_sq must share the parameter's memory layout (the update kernels "
                               "walk all four tensors with one flat index); load_state_dict re-lays loaded moments out")

    def load_state_dict(self, state_dict) -> None:
        """torch.optim.Optimizer.load_state_dict keeps the strides the file's tensors were saved with: moments written by the
        reference (NCDHW-contiguous, model/adam.py state) would sit next to a channel-last `k0` here, and the kernels -- which
        walk parameter, gradient and both moments with ONE flat offset -- would pair each element with another voxel's
        moments.  Re-lay every loaded moment out like its parameter (a no-op when the layouts already agree)."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            for p in group['params']:
                st = self.state.get(p)
                if st:
                    for k in ('exp_avg', 'exp_avg_sq'):
                        if torch.is_tensor(st.get(k)) and st[k].shape == p.shape:
                            st[k] = _as_layout_of(st[k].to(p.device), p)

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def use_skip_flag(self, skip_ptr=None) -> None:
        """Sync-free steps that are NOT captured (fused.set_sync_free without a device schedule): step sizes stay host scalars,
        but every update kernel honours the device int at `skip_ptr` -- a step whose survivor list overflowed its buffers
        consumes its gradient and changes nothing, exactly like a captured one.  None switches it off."""
        self._skip = skip_ptr

    def _sched(self, group):
        """(step-size device pointer or None, skip-flag device pointer or None) for `group` under the current schedule."""
        if self._dev is not None:
            gi = next(i for i, gr in enumerate(self.param_groups) if gr is group)
            return self._dev['ss'][gi], self._dev['skip']
        return None, self._skip

    def _flush_small(self, batch, b1, b2, eps, skip=None):
        """One launch for every small tensor of the step that shares (betas, eps): rows (param, grad, exp_avg, exp_avg_sq, step,
        lr, masked, step-size device pointer or None) -- the groups' learning rates / step sizes are per tensor."""
        n = len(batch)
        if n == 0:
            return
        P = ctypes.c_void_p
        tables = [(P * n)(*[row[k].data_ptr() for row in batch]) for k in range(4)]
        sizes = (ctypes.c_int64 * n)(*[row[0].numel() for row in batch])
        masked = (ctypes.c_int * n)(*[int(bool(row[6])) for row in batch])
        steps = (ctypes.c_int * n)(*[row[4] for row in batch])
        lrs = (ctypes.c_float * n)(*[row[5] for row in batch])
        if skip is not None or any(row[7] is not None for row in batch):
            ss = (P * n)(*[row[7] for row in batch]) if all(row[7] is not None for row in batch) else None
            if ss is None and any(row[7] is not None for row in batch):
                raise RuntimeError("MaskedAdam: a device schedule must cover every parameter group")
            call("fgs_adam_upd_multi_dev", n, *tables, sizes, ss, steps, lrs, masked, float(b1), float(b2), float(eps), skip,
                 stream())
            return
        call("fgs_adam_upd_multi", n, *tables, sizes, steps, lrs, masked, float(b1), float(b2), float(eps), stream())

    def _bricks(self, p, g, group, st, ss_ptr=None, skip=None) -> bool:
        """The masked update of a feature grid whose gradient lives in fused.py's persistent self-cleaning buffer: visit
        only the 4x4x4-voxel bricks recorded for this step (`p._fgs_touched`, fused._publish_touched / dist.GradAverager)
        and zero the gradient consumed (fgs_adam_upd_bricks; per element the arithmetic of masked_adam_upd,
        model/cuda/adam_upd_kernel.cu:25-40).  False -> the record is absent or no longer describes `g` (something else
        wrote into the gradient): the caller takes the dense kernels, and the buffer is zero-filled before its next use."""
        t = getattr(p, '_fgs_touched', None)
        if t is None:
            return False
        p._fgs_touched = None
        gb = t['state']
        if not (t['valid'] and group['skip_zero_grad'] and g.data_ptr() == t['grad_ptr'] and gb['buf']._version == t['version']
                and tuple(p.shape) == gb['key'][0] and p.stride() == gb['key'][1] and p.data_ptr() != g.data_ptr()):
            return False
        if (self.per_lr is not None and p.shape == self.per_lr.shape) or (t['exchange'] and t['idx'] is None):
            return False
        from ._lib import ptr
        b1, b2 = group['betas']
        idx, n = t['idx'], t['n']
        self._check_layout(p, st)
        tail = (int(st['step']), float(b1), float(b2), float(group['lr']), float(group['eps']), ss_ptr, skip, stream())
        if idx is None:       # this rank's own scatter: the voxels recorded in the 64-bit brick masks
            call("fgs_adam_upd_voxels", ptr(p), ptr(g), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), *t['dims'], ptr(t['flags']),
                 *tail)
        else:                 # after a brick-sparse exchange: the union's bricks, whole (other ranks' voxels are not in the masks)
            # (device-counted exchange, dist.GradAverager.use_device_counts: the list's length is a device int64, n its capacity)
            call("fgs_adam_upd_bricks", ptr(p), ptr(g), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), *t['dims'], ptr(idx),
                 ptr(t.get('count_dev')), int(n), None, *tail)
            t['flags'].zero_()
        gb['clean'] = True
        return True

    @torch.no_grad()
    def voxel_update_from_buffer(self, p, gb, ss_ptr, skip_ptr) -> None:
        """The masked update of a feature grid straight from fused.py's persistent gradient buffer `gb` (its recorded voxels:
        fgs_adam_upd_voxels), step size and skip flag read from device memory -- the form graph_step.CapturedFineStep issues at
        the HEAD of an iteration for the gradient the previous iteration left (nothing pending: a no-op).  Consumes what it
        applies: gradient zeroed, voxel flags cleared."""
        from ._lib import ptr
        group = next(gr for gr in self.param_groups if any(q is p for q in gr['params']))
        st = self._state_of(p)
        self._check_layout(p, st)
        b1, b2 = group['betas']
        call("fgs_adam_upd_voxels", ptr(p), ptr(gb['buf']), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), *gb['dims'], ptr(gb['flags']),
             int(st['step']), float(b1), float(b2), float(group['lr']), float(group['eps']), ss_ptr, skip_ptr, stream())

    def defer_update(self, p, grad) -> bool:
        """Called from inside the backward pass in place of early_update(): the gradient of `p` stays in the persistent buffer,
        step() leaves the parameter alone, and the caller (CapturedFineStep) applies the update later with
        voxel_update_from_buffer().  False: the record does not describe this gradient (see _bricks) -- update now instead."""
        group = next((gr for gr in self.param_groups if any(q is p for q in gr['params'])), None)
        t = getattr(p, '_fgs_touched', None)
        if group is None or t is None:
            return False
        gb = t['state']
        if not (t['valid'] and group['skip_zero_grad'] and grad.data_ptr() == t['grad_ptr'] and gb['buf']._version == t['version']
                and tuple(p.shape) == gb['key'][0] and p.stride() == gb['key'][1] and t['idx'] is None and not t['exchange']
                and not (self.per_lr is not None and p.shape == self.per_lr.shape)):
            return False
        p._fgs_touched = None
        gb['clean'] = True           # (a promise: the deferred pass zeroes what it consumes before the buffer is scattered into again)
        self._early[id(p)] = None    # step() skips the parameter
        return True

    def _big(self, p, g, st, group, ss_ptr, skip) -> None:
        """One launch for one big tensor: the reference's per-tensor rule (model/adam.py:205-221) -- per-voxel lr if a
        same-shape table is set, else masked if the group says skip_zero_grad, else dense."""
        from ._lib import ptr
        b1, b2 = group['betas']
        if self.per_lr is not None and p.shape == self.per_lr.shape:
            mode, perlr = 2, _as_layout_of(self.per_lr, p)
        else:
            mode, perlr = (1 if group['skip_zero_grad'] else 0), None
        m, v = st['exp_avg'], st['exp_avg_sq']
        if ss_ptr is None and skip is None:        # the reference's own three entry points (ops.adam_upd_cuda)
            hyper = (st['step'], b1, b2, group['lr'], group['eps'])
            if mode == 2:
                adam_upd_cuda.adam_upd_with_perlr(p, g, m, v, perlr, *hyper)
            elif mode == 1:
                adam_upd_cuda.masked_adam_upd(p, g, m, v, *hyper)
            else:
                adam_upd_cuda.adam_upd(p, g, m, v, *hyper)
            return
        self._check_layout(p, st)
        if not (p.is_cuda and g.is_cuda and p.dtype == torch.float32 and g.stride() == p.stride()):
            raise RuntimeError("MaskedAdam: device-schedule / skip-flag updates need float32 CUDA tensors of one layout")
        call("fgs_adam_upd_dev", ptr(p), ptr(g), ptr(m), ptr(v), ptr(perlr), p.numel(), ss_ptr, int(st['step']),
             float(group['lr']), float(b1), float(b2), float(group['eps']), mode, skip, stream())

    def _update_one(self, p, g, group):
        """The update of one big tensor under the current schedule (early_update's body)."""
        ss_ptr, skip = self._sched(group)
        st = self._state_of(p)
        if self._dev is None:
            st['step'] += 1           # (with a device schedule the host mirror is advanced by whoever replays the step)
        g = _as_layout_of(g, p)
        if not self._bricks(p, g, group, st, ss_ptr=ss_ptr, skip=skip):
            self._big(p, g, st, group, ss_ptr, skip)

    @torch.no_grad()
    def early_update(self, p, grad, on_stream=None) -> bool:
        """Apply this step's update of ONE parameter now, from inside the backward pass, as soon as its gradient is
        final (fused.py calls this for the feature grid right after its scatter kernel): the update -- and, on several
        GPUs, the gradient exchange in front of it -- then runs beside the rest of the backward pass instead of after
        it.  `on_stream`: the stream the caller is already on (dist.GradAverager's exchange stream); otherwise a
        high-priority stream of this optimizer, ordered after the caller's current stream.  `step()` skips the parameter
        and ends by making the current stream wait for the update, so everything after `step()` sees the new values.
        Only valid when nothing else adds to this parameter's gradient between backward and step (no TV on it)."""
        group = next((gr for gr in self.param_groups if any(q is p for q in gr['params'])), None)
        if group is None or not p.is_cuda:
            return False
        if id(p) in self._early:
            # a second backward pass before step() (gradient accumulation, two losses backpropagated separately): the first
            # pass's gradient has already been applied and consumed, this one would be dropped without a word
            raise RuntimeError("MaskedAdam.early_update: this parameter was already updated from inside an earlier backward pass "
                               "of the same step; with several backward passes per step switch the in-backward update off "
                               "(fused.disable_early_update)")
        if on_stream is not None:
            self._update_one(p, grad, group)
            if on_stream == 'inline':          # the caller's stream IS the stream step() runs on: ordered by issue
                done = None
            else:
                done = torch.cuda.Event()
                done.record()
        else:
            if self._early_stream is None:
                self._early_stream = torch.cuda.Stream(device=p.device, priority=-1)
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(self._early_stream):
                self._early_stream.wait_event(ready)
                self._update_one(p, grad, group)
                done = torch.cuda.Event()
                done.record()
            grad.record_stream(self._early_stream)
        self._early[id(p)] = done
        return True

    @torch.no_grad()
    def step(self):
        """One update of every parameter that has a gradient.  Host schedule (default): step sizes from `state[p]['step']`
        and the group's lr, as model/adam.py:195-221; device schedule (`use_device_schedule`): step sizes and the skip flag
        are read from device memory, nothing that changes from step to step is passed by value."""
        early, self._early = self._early, {}
        host = self._dev is None
        from . import fused_ops as _fo
        _fo.WEIGHTS_EPOCH[0] += 1           # (packed weight images of the MLP chains are stale from here on: fused_ops.rc2_pack)
        small = {}        # (b1, b2, eps, skip) -> rows: the small tensors of ALL groups with the same hyper-parameters, ONE launch
        for group in self.param_groups:
            b1, b2 = group['betas']
            ss_ptr, skip = self._sched(group)
            masked = group['skip_zero_grad']
            for p in group['params']:
                if p.grad is None or id(p) in early:    # already updated by early_update()
                    continue
                if self.before_param is not None:       # dist.GradAverager.wait_for: this gradient's exchange is done
                    self.before_param(p)
                st = self._state_of(p)
                if host:
                    st['step'] += 1
                g = _as_layout_of(p.grad, p)
                if self._bricks(p, g, group, st, ss_ptr=ss_ptr, skip=skip):
                    continue
                per_lr = self.per_lr is not None and p.shape == self.per_lr.shape
                if (not per_lr and p.is_cuda and p.numel() < _SMALL and p.is_contiguous() and g.is_contiguous()
                        and p.dtype == torch.float32):
                    small.setdefault((float(b1), float(b2), float(group['eps']), skip), []).append(
                        (p, g, st['exp_avg'], st['exp_avg_sq'], st['step'], group['lr'], masked, ss_ptr))
                else:
                    if p.dim() <= 2 and self.before_small is not None:    # (an MLP tensor too large for the shared launch)
                        self.before_small()
                    self._big(p, g, st, group, ss_ptr, skip)
        if small and self.before_small is not None:     # (graph_step.CapturedFineStep: the deferred join of the weight-gradient branch)
            self.before_small()
        for (b1, b2, eps, skip), rows in small.items():
            self._flush_small(rows, b1, b2, eps, skip=skip)
        for done in early.values():                     # everything after step() sees the early updates
            if done is not None:
                torch.cuda.current_stream().wait_event(done)
