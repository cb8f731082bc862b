"""Data-parallel ray sharding: one process per GPU, grids and MLPs replicated, one gradient exchange per step.

The reference has no distributed code; this is the north_star's multi-GPU extension.
Rank r renders rays [r*N/P, (r+1)*N/P) of the global batch with a loss that is a mean over its local rays; averaging
the gradients over ranks (sum all-reduce, then 1/P) gives exactly the gradient of the global-batch mean loss.  TV and
MaskedAdam then run identically on every rank, so the replicas stay bit-identical without a parameter broadcast.

Collectives go through torch.distributed: backend "nccl" is RCCL over xGMI on ROCm, "gloo" is used by the CPU tests.

What is exchanged (160^3: k0.grad 197 MB, sdf.grad 16 MB, MLP grads 1.7 MB; 320^3: 1.57 GB + 131 MB):
  * small tensors (MLP weights / biases): packed into one bucket, one all-reduce;
  * medium dense tensors (the 1-channel sdf gradient): one all-reduce each, in place on the flat storage;
  * the multi-channel feature-grid gradient: **brick-sparse**.  Rays only touch voxels near the surface, so a few per cent
    of k0.grad is non-zero on a rank and, the scene being the same on all ranks, the union over ranks is barely larger.
    The grid is cut into 4x4x4-voxel bricks; ranks OR their brick-occupancy flags (one tiny all-reduce), gather the
    union's bricks into a dense [n_bricks, 64*C] buffer, sum-all-reduce that, and scatter it back.  A brick whose flag is
    clear is zero on every rank, so the result equals the dense all-reduce bit for bit in which elements are non-zero
    (masked_adam_upd keys on grad != 0) and, up to the collective's summation order, in value.  Falls back to the dense
    exchange when more than `sparse_max_fill` of the bricks are occupied or the grid sides are not multiples of 4.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

BRICK = 4


def shard_rays(n_total: int, rank: int, world_size: int) -> slice:
    """Contiguous ray shard of rank `rank` (remainder rays go to the lowest ranks)."""
    base, rem = divmod(n_total, world_size)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


def _brick_view(g: torch.Tensor):
    """[1,C,X,Y,Z] gradient -> view [X/4, 4, Y/4, 4, Z/4, 4, C] over the same storage, or None if not applicable."""
    if g.dim() != 5 or g.shape[0] != 1:
        return None
    _, C, X, Y, Z = g.shape
    if X % BRICK or Y % BRICK or Z % BRICK:
        return None
    v = g[0].permute(1, 2, 3, 0)                      # [X,Y,Z,C] logical view (contiguous for channel-last storage)
    return v.reshape(X // BRICK, BRICK, Y // BRICK, BRICK, Z // BRICK, BRICK, C) if v.is_contiguous() else None


class GradAverager:
    """Averages `.grad` of the given parameters over the process group (see the module docstring for the scheme)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None,
                 big_numel: int = 1 << 20, sparse_min_numel: int = 1 << 24, sparse_max_fill: float = 0.5,
                 force: bool = False, sparse_1ch_min_numel: Optional[int] = 1 << 24, sparse_1ch_eager: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.big_numel = big_numel
        self.sparse_min_numel = sparse_min_numel
        self.sparse_max_fill = sparse_max_fill
        # 1-channel grids (the sdf gradient: 16 MB at 160^3, 131 MB at 320^3, touched in a shell around the surface like k0's) go
        # brick-sparse too from this size on (None: always dense; default 256^3 = 67 MB: below that the six extra launches of the
        # sparse form -- flags, compact, guard, gather, scatter, the flags' own all-reduce -- cost about what the smaller
        # collective saves; single-rank rehearsal at 160^3: +35 us per step); their occupancy is read from the gradient itself.  In the
        # device-counted form (use_device_counts, the captured step) that costs one streaming pass; the host-counted form needs
        # a blocking nonzero() at the very end of the backward pass, so eager steps keep the dense all-reduce unless
        # `sparse_1ch_eager` asks otherwise (the gloo tests do).
        self.sparse_1ch_min_numel = sparse_1ch_min_numel
        self.sparse_1ch_eager = sparse_1ch_eager
        self.last_sparse_fill_1ch = None
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()      # run the exchange even in a group of one (path rehearsal)
        # RCCL averages inside the collective (ncclAvg); gloo only sums, so the CPU tests scale afterwards
        import os
        self.avg_in_collective = (dist.is_initialized() and dist.get_backend(group) == "nccl"
                                  and os.environ.get("FGS_DIST_SUM") != "1")
        # The early exchanges (issued from inside the backward pass on a side stream) get a communicator of their own: with a
        # single one, collectives execute in issue order, and the sdf all-reduce issued at the end of the backward pass
        # would queue behind the much larger k0 exchange instead of running beside it.
        self.early_group = group
        if dist.is_initialized() and (self.world_size > 1 or self.force) and os.environ.get("FGS_DIST_ONE_COMM") != "1":
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self.early_group = dist.new_group(ranks=ranks, backend=dist.get_backend(group))
        self._deferred = {}               # id(param) -> event of its early exchange, waited on by wait_for()
        self.after_early = None           # callable(param, grad): runs on the exchange stream right after an early grid
                                          # exchange (MaskedAdam.early_update through fused.enable_early_update)
        self.defer_to_optimizer = False   # set by attach_optimizer(): the optimizer waits per parameter, not average()
        self._bucket = None
        self.last_sparse_fill = None      # fraction of bricks exchanged by the last sparse reduction (diagnostics)
        self._hints = {}                  # id(param) -> state of hint_touched()
        self._static = {}                 # id(param) -> device-counted exchange (use_device_counts): capacity, buffer, guard
        self.max_union_bricks = {}        # id(param) -> largest union brick count a host-counted exchange has seen
        self.dense_sources = {}           # id(param) -> True while a dense term (TV loss) also writes that gradient
        self.last_hint_wait_us = 0.0      # host time spent waiting for the hinted brick count (diagnostics)

    # ------------------------------------------------------------------------------------------------ early occupancy
    def hint_touched(self, param: torch.nn.Parameter, pts: torch.Tensor, xyz_min, xyz_max, count_ptr=None) -> None:
        """Tell the averager, right after the forward, which sample points the backward of the DenseGrid `param` will
        scatter into (`pts` [M,3], the survivor list).  The brick occupancy (a superset of the non-zero bricks), its union
        over ranks and the compacted brick list are then produced on a side stream while the MLP forward/backward runs,
        and `average()` sizes the exchange from a count that is already on the host: no blocking nonzero(), no pass over
        the 197 MB gradient.  Only valid when every gradient of `param` in this step comes from trilinear lookups at
        `pts`; without a hint the occupancy is read from the gradient itself.  `count_ptr`: device address of the survivor count
        when `pts` has CAPACITY rows (the result dict's 'survivor_count_ptr' of a sync-free forward)."""
        if (self.world_size == 1 and not self.force) or not (pts.is_cuda and param.dim() == 5):
            return
        _, C, X, Y, Z = param.shape
        # The same predicate as the consumer (`early('k0')` / `average()` take the sparse path only for grids of at least
        # `sparse_min_numel` elements): every condition that decides whether the collective below is issued is a function
        # of the parameter's SHAPE, identical on all ranks -- never of rank-local state (survivor counts, allocator
        # addresses), or ranks would disagree on the collective sequence of `self.group` and hang.
        if X % BRICK or Y % BRICK or Z % BRICK or C == 1 or param.numel() < self.sparse_min_numel:
            return
        if self.dense_sources.get(id(param), False):
            return        # something else (a TV loss on this grid) adds a dense gradient: occupancy comes from the gradient
        import ctypes
        from ._lib import call, dyn, ptr, stream
        total = (X // BRICK) * (Y // BRICK) * (Z // BRICK)
        h = self._hints.get(id(param))
        if h is None or h['total'] != total:
            dev = pts.device
            # (one entry more than there are bricks: the device-counted form sends this rank's "my survivor list overflowed"
            # flag along with the occupancy, so that the decision to skip a step's update is the same on every rank)
            h = dict(total=total, flags=torch.zeros(total + 1, dtype=torch.int32, device=dev),
                     idx=torch.empty(total, dtype=torch.int64, device=dev), count=torch.empty(1, dtype=torch.int64, device=dev),
                     count_host=torch.empty(1, dtype=torch.int64).pin_memory(), stream=torch.cuda.Stream(device=dev),
                     event=torch.cuda.Event(), armed=False)
            self._hints[id(param)] = h
        box_key = (id(xyz_min), getattr(xyz_min, '_version', 0), id(xyz_max), getattr(xyz_max, '_version', 0))
        if h.get('box_key') != box_key:      # one device->host read per box, not per step
            h['lo'] = (ctypes.c_float * 3)(*[float(v) for v in torch.as_tensor(xyz_min).flatten().tolist()])
            h['hi'] = (ctypes.c_float * 3)(*[float(v) for v in torch.as_tensor(xyz_max).flatten().tolist()])
            h['box_key'] = box_key
            h['box_objs'] = (xyz_min, xyz_max)      # kept alive: an id() in the key cannot be reused by a successor
        lo, hi = h['lo'], h['hi']
        pts = pts.detach().contiguous()
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(h['stream']):
            h['stream'].wait_event(ready)
            sc = self._static.get(id(param))
            h['flags'].zero_()
            call("fgs_brick_flags_pts", ptr(pts), pts.shape[0], lo, hi, X, Y, Z, ptr(h['flags']), dyn(row_count=count_ptr), stream())
            if sc is not None and sc['guard_flags'] is not None:
                h['flags'][total:].copy_(sc['guard_flags'][1:2])                               # this rank's skip flag rides along
            dist.all_reduce(h['flags'], op=dist.ReduceOp.MAX, group=self.group)               # union over ranks
            call("fgs_brick_compact", ptr(h['flags']), total, ptr(h['idx']), ptr(h['count']), stream())
            if sc is not None:     # the count stays on the device: checked against the capacity there (csrc/bricks.hip)
                call("fgs_brick_count_guard", ptr(h['count']), sc['capacity'], ptr(sc['flags']), ptr(sc['sticky']),
                     ptr(h['flags'][total:]) if sc['guard_flags'] is not None else None, stream())
            else:
                h['count_host'].copy_(h['count'], non_blocking=True)
            h['event'].record()
        pts.record_stream(h['stream'])
        h['armed'] = True
        h['unordered'] = True      # (its all-reduce ran on `self.group` from a stream of its own: see average())

    # ------------------------------------------------------------------------------------------------ pieces
    def _dense(self, g: torch.Tensor, async_op: bool, group=None):
        flat = g.as_strided((g.numel(),), (1,))       # the dense storage as a flat view (layout-agnostic, no copy)
        return dist.all_reduce(flat, op=self._op(), group=group or self.group, async_op=async_op), flat

    def _op(self):
        return dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM

    def _post_scale(self, t: torch.Tensor, inv: float) -> None:
        if not self.avg_in_collective:
            t.mul_(inv)

    def _sparse(self, g: torch.Tensor, inv: float, param=None, group=None) -> bool:
        """Brick-sparse exchange of one multi-channel grid gradient.  Returns False if the dense path should be used."""
        group = group or self.group
        bv = _brick_view(g)
        if bv is None:
            return False
        nbx, _, nby, _, nbz, _, C = bv.shape
        total = nbx * nby * nbz
        on_gpu = g.is_cuda
        dims = (C, nbx * BRICK, nby * BRICK, nbz * BRICK)
        h = self._hints.get(id(param)) if param is not None else None
        sc = self._static.get(id(param)) if param is not None else None
        if on_gpu and sc is not None:
            # device-counted form (use_device_counts): fixed-capacity buffer, fixed-size collective, nothing read by the host
            from ._lib import call, ptr, stream
            cap, buf = sc['capacity'], sc['buf']
            if h is not None and h['armed'] and h['total'] == total:
                # occupancy hinted from the survivor points: union, brick list and guard were produced on the side stream
                h['armed'] = False
                torch.cuda.current_stream().wait_event(h['event'])
                idx, count = h['idx'], h['count']
            else:
                # no hint (the 1-channel sdf gradient: every alive sample and every tap writes it): occupancy from the
                # gradient itself, one streaming pass; then the same union / list / guard, all on this stream
                u = sc.get('own')
                if u is None or u['flags'].numel() != total + 1:
                    u = sc['own'] = dict(flags=torch.zeros(total + 1, dtype=torch.int32, device=g.device),
                                         idx=torch.empty(total, dtype=torch.int64, device=g.device),
                                         count=torch.zeros(1, dtype=torch.int64, device=g.device))
                call("fgs_brick_flags", ptr(g), *dims, ptr(u['flags']), stream())
                if sc['guard_flags'] is not None:
                    u['flags'][total:].copy_(sc['guard_flags'][1:2])
                else:
                    u['flags'][total:].zero_()
                dist.all_reduce(u['flags'], op=dist.ReduceOp.MAX, group=group)
                call("fgs_brick_compact", ptr(u['flags']), total, ptr(u['idx']), ptr(u['count']), stream())
                call("fgs_brick_count_guard", ptr(u['count']), cap, ptr(sc['flags']), ptr(sc['sticky']), ptr(u['flags'][total:]),
                     stream())
                idx, count = u['idx'], u['count']
            call("fgs_brick_gather_dev", ptr(g), *dims, ptr(idx), ptr(count), cap, ptr(buf), stream())
            dist.all_reduce(buf, op=self._op(), group=group)
            call("fgs_brick_scatter_dev", ptr(g), *dims, ptr(idx), ptr(count), cap, ptr(buf),
                 1.0 if self.avg_in_collective else float(inv), stream())
            self._note_union(param, g, idx, cap, count_dev=count)
            self.last_sparse_fill = None
            return True
        if on_gpu and h is not None and h['armed'] and h['total'] == total:
            from ._lib import call, ptr, stream
            h['armed'] = False
            # The brick count sizes a collective, so every rank must read the SAME number here: a non-blocking "use it
            # if it has arrived, else go dense" would let ranks choose different collectives.  The count was produced on
            # the side stream right behind the survivor list (before the MLP forward ran on the device), so this wait is
            # over as soon as the device has passed the march kernels of THIS step -- one MLP forward + backward before
            # the gradient this exchange needs exists; the device never idles on it.  `last_hint_wait_us` records it.
            if not h['event'].query():
                import time
                t0 = time.perf_counter()
                h['event'].synchronize()
                self.last_hint_wait_us = (time.perf_counter() - t0) * 1e6
            else:
                self.last_hint_wait_us = 0.0
            torch.cuda.current_stream().wait_event(h['event'])
            n = int(h['count_host'][0])
            self.max_union_bricks[id(param)] = max(self.max_union_bricks.get(id(param), 0), n)
            self.last_sparse_fill = n / max(total, 1)
            if n > self.sparse_max_fill * total:
                return False
            self._note_union(param, g, h['idx'], n)
            if n == 0:
                return True
            idx = h['idx'][:n]
            buf = torch.empty(n, BRICK ** 3 * C, dtype=g.dtype, device=g.device)
            call("fgs_brick_gather", ptr(g), *dims, ptr(idx), n, ptr(buf), stream())
            dist.all_reduce(buf, op=self._op(), group=group)
            call("fgs_brick_scatter", ptr(g), *dims, ptr(idx), n, ptr(buf), 1.0 if self.avg_in_collective else float(inv),
                 stream())
            return True
        if sc is not None:
            return self._sparse_counted_host(g, bv, inv, sc, group)
        if on_gpu:       # csrc/bricks.hip: one streaming pass over the gradient
            from ._lib import call, ptr, stream
            flags = torch.empty(total, dtype=torch.int32, device=g.device)
            call("fgs_brick_flags", ptr(g), *dims, ptr(flags), stream())
        else:            # host tensors (gloo tests): the same thing with torch indexing
            flags = (bv != 0).any(dim=6).any(dim=5).any(dim=3).any(dim=1).to(torch.int32).reshape(-1)
        dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)                              # union of occupancy
        idx = flags.nonzero(as_tuple=False).squeeze(1)                                          # identical on all ranks
        n = int(idx.numel())
        if param is not None:
            self.max_union_bricks[id(param)] = max(self.max_union_bricks.get(id(param), 0), n)
        self.last_sparse_fill = n / max(total, 1)
        if n > self.sparse_max_fill * total:
            return False
        if on_gpu:
            self._note_union(param, g, idx, n)
        if n == 0:
            return True
        if on_gpu:
            buf = torch.empty(n, BRICK ** 3 * C, dtype=g.dtype, device=g.device)
            call("fgs_brick_gather", ptr(g), *dims, ptr(idx), n, ptr(buf), stream())
            dist.all_reduce(buf, op=self._op(), group=group)
            call("fgs_brick_scatter", ptr(g), *dims, ptr(idx), n, ptr(buf), 1.0 if self.avg_in_collective else float(inv),
                 stream())
            return True
        bx = idx // (nby * nbz)
        by = (idx // nbz) % nby
        bz = idx % nbz
        buf = bv[bx, :, by, :, bz, :, :].contiguous()                                           # [n,4,4,4,C] gather
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        buf.mul_(inv)
        bv[bx, :, by, :, bz, :, :] = buf                                                        # scatter back
        return True

    def _sparse_counted_host(self, g, bv, inv, sc, group) -> bool:
        """The device-counted exchange on HOST tensors (gloo, the CPU tests): the same protocol with torch indexing -- the
        occupancy and this rank's skip flag are all-reduced together, the union's count is checked against the capacity
        with `sc`'s own state (the guard kernel's rules: csrc/bricks.hip k_brick_count_guard), the collective always
        carries `capacity` rows, rows behind the count are zero, bricks behind the capacity are left alone."""
        nbx, _, nby, _, nbz, _, C = bv.shape
        total = nbx * nby * nbz
        flags = torch.zeros(total + 1, dtype=torch.int32)
        flags[:total] = (bv != 0).any(dim=6).any(dim=5).any(dim=3).any(dim=1).reshape(-1)
        if sc['guard_flags'] is not None:
            flags[total] = sc['guard_flags'][1]
        dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
        idx = flags[:total].nonzero(as_tuple=False).squeeze(1)
        n, cap = int(idx.numel()), sc['capacity']
        if n > cap or int(sc['sticky'][0]):
            sc['sticky'][0] = 1
            sc['flags'][0] = 1
            sc['flags'][1] = 1
        if int(flags[total]):
            sc['flags'][0] = 1
            sc['flags'][1] = 1
        idx = idx[:cap]
        bx, by, bz = idx // (nby * nbz), (idx // nbz) % nby, idx % nbz
        buf = sc['buf'].view(cap, BRICK, BRICK, BRICK, C)
        buf.zero_()
        buf[:idx.numel()] = bv[bx, :, by, :, bz, :, :]
        dist.all_reduce(sc['buf'], op=dist.ReduceOp.SUM, group=group)                           # always `capacity` rows
        bv[bx, :, by, :, bz, :, :] = buf[:idx.numel()] * inv
        self.last_sparse_fill = None
        return True

    @staticmethod
    def _note_union(param, g, idx, n, count_dev=None) -> None:
        """After a brick-sparse exchange the gradient is non-zero only inside the union's bricks: hand that list to
        MaskedAdam's brick update (adam.MaskedAdam._bricks; the record was opened by fused._publish_touched).  A dense
        exchange reports nothing, and the update then goes dense as well."""
        t = getattr(param, '_fgs_touched', None) if param is not None else None
        if t is not None and t['exchange'] and t['grad_ptr'] == g.data_ptr():
            t['idx'], t['n'], t['count_dev'] = idx, int(n), count_dev     # (count_dev: n is then the list's capacity)

    def use_device_counts(self, param, capacity: Optional[int], guard_flags: Optional[torch.Tensor] = None) -> None:
        """Switch the brick-sparse exchange of the grid `param` to its DEVICE-COUNTED form (or back, capacity=None): the
        union's brick count never reaches the host.  The exchange buffer holds `capacity` bricks and the all-reduce always
        carries all of them (rows behind the count are zero); gather / scatter / MaskedAdam's brick update read the count
        from device memory.  What a host-counted step decides from the number -- "too full: go dense" -- cannot be decided
        inside a captured step, where every rank must issue the same fixed-size collective: a union that does not fit
        raises `guard_flags[0]` and keeps `guard_flags[1]` (the optimizer kernels' skip flag, fused.set_sync_free) raised
        from then on, on every rank alike (the count is the all-reduced union's), until the host has looked
        (`device_count_state`), reset the gradient buffer and chosen a larger capacity.  `guard_flags`: the 2-int flag
        buffer of fused.set_sync_free; this rank's flags[1] (its survivor list overflowed) travels with the occupancy
        all-reduce, so that a step one rank must skip is skipped by all of them.
        `capacity` must be the same on every rank: derive it from `suggested_capacity` (the union count is all-reduced)."""
        if capacity is None:
            self._static.pop(id(param), None)
            return
        _, C, X, Y, Z = param.shape
        total = (X // BRICK) * (Y // BRICK) * (Z // BRICK)
        capacity = max(1, min(int(capacity), total))
        sc = self._static.get(id(param))
        if sc is None or sc['capacity'] != capacity or sc['buf'].device != param.device:
            sc = dict(capacity=capacity, buf=torch.zeros(capacity, BRICK ** 3 * C, dtype=param.dtype, device=param.device),
                      sticky=torch.zeros(1, dtype=torch.int32, device=param.device))
        own = guard_flags is None
        sc['flags'] = torch.zeros(2, dtype=torch.int32, device=param.device) if own else guard_flags
        sc['guard_flags'] = None if own else guard_flags
        self._static[id(param)] = sc

    def suggested_capacity(self, param, margin: float = 1.25, granule: int = 256) -> Optional[int]:
        """Capacity for `use_device_counts` from the largest union brick count the host-counted exchanges of `param` have
        seen so far (identical on every rank: the count is the all-reduced union's), or None before the first one."""
        n = self.max_union_bricks.get(id(param))
        if not n:
            return None
        return (int(n * margin) + granule - 1) // granule * granule

    def last_device_count(self, param) -> Optional[int]:
        """The union brick count of the most recent device-counted exchange of `param` whose occupancy came from the gradient
        itself (one device->host read; None before the first one).  CapturedFineStep sizes the sdf exchange from it after its
        eager warm-up pass."""
        sc = self._static.get(id(param))
        if sc is None or sc.get('own') is None:
            return None
        return int(sc['own']['count'].cpu()[0])

    def device_count_state(self, param, clear: bool = False):
        """(exchange overflowed: bool) of the device-counted exchange of `param` -- one device->host read; `clear` lowers
        the sticky flag again (after the caller has reset the gradient buffer: fused.reset_grid_grad)."""
        sc = self._static.get(id(param))
        if sc is None:
            return False
        over = bool(int(sc['sticky'].cpu()[0]))
        if clear:
            sc['sticky'].zero_()
        return over

    def _sparse_1ch(self, g) -> bool:
        """Shape-only predicate (identical on every rank) for the brick-sparse exchange of a 1-channel grid gradient."""
        return (self.sparse_1ch_min_numel is not None and g.dim() == 5 and g.shape[0] == 1 and g.shape[1] == 1
                and g.numel() >= self.sparse_1ch_min_numel and not (g.shape[2] % BRICK or g.shape[3] % BRICK or g.shape[4] % BRICK)
                and g.is_contiguous())

    def tune_sparse_1ch(self, param, fill_estimate: float = 0.3, kernel_overhead_us: float = 35.0, reps: int = 7) -> dict:
        """Decide by MEASUREMENT on the ranks at hand whether the 1-channel grid `param` (the sdf gradient) is exchanged
        brick-sparse or dense, and set `sparse_1ch_min_numel` accordingly -- the same on every rank (rank 0's verdict is
        broadcast).  Timed, eagerly, with this group's collectives: the dense all-reduce of the whole grid against the occupancy
        all-reduce (one int per brick) plus the all-reduce of an exchange buffer of 1.5 x `fill_estimate` x the bricks, plus
        `kernel_overhead_us` for the five extra launches of the sparse form (flags, compact, guard, gather, scatter: measured
        35 us at 160^3).  With one rank there is nothing to measure: the shape threshold stays.  Returns what it measured."""
        out = dict(tuned=False)
        if (self.world_size == 1 and not self.force) or not (param.dim() == 5 and param.shape[1] == 1):
            return out
        if self.sparse_1ch_min_numel is None:        # "always dense", asked for explicitly: nothing to decide
            return out
        _, _, X, Y, Z = param.shape
        if X % BRICK or Y % BRICK or Z % BRICK:
            return out
        total = (X // BRICK) * (Y // BRICK) * (Z // BRICK)
        seen = self.max_union_bricks.get(id(param))      # a union count measured on this grid beats the estimate
        if seen:
            fill_estimate = seen / total
            out['fill_measured'] = round(fill_estimate, 4)
        cap = min(total, (int(1.5 * fill_estimate * total) + 255) // 256 * 256)
        dev = param.device
        dense = torch.zeros(param.numel(), dtype=torch.float32, device=dev)
        flags = torch.zeros(total + 1, dtype=torch.int32, device=dev)
        buf = torch.zeros(cap * BRICK ** 3, dtype=torch.float32, device=dev)

        import time
        on_gpu = param.is_cuda

        def timed(fn):
            for _ in range(2):
                fn()
            ts = []
            for _ in range(reps):
                dist.barrier(group=self.group)
                if on_gpu:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize(dev)
                    e0.record()
                    fn()
                    e1.record()
                    torch.cuda.synchronize(dev)
                    ts.append(e0.elapsed_time(e1) * 1e3)
                else:                       # (host tensors: the gloo tests walk the same protocol)
                    t0 = time.perf_counter()
                    fn()
                    ts.append((time.perf_counter() - t0) * 1e6)
            return sorted(ts)[len(ts) // 2]

        t_dense = timed(lambda: dist.all_reduce(dense, op=self._op(), group=self.group))
        t_flags = timed(lambda: dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group))
        t_buf = timed(lambda: dist.all_reduce(buf, op=self._op(), group=self.group))
        t_sparse = t_flags + t_buf + kernel_overhead_us
        verdict = torch.tensor([1 if t_sparse < 0.9 * t_dense else 0], dtype=torch.int32, device=dev)
        dist.broadcast(verdict, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        sparse = bool(int(verdict.item()))
        self.sparse_1ch_min_numel = min(self.sparse_1ch_min_numel or (1 << 62), param.numel()) if sparse \
            else max(self.sparse_1ch_min_numel or 0, param.numel() + 1)
        out.update(tuned=True, sparse=sparse, dense_us=round(t_dense, 1), flags_us=round(t_flags, 1), buffer_us=round(t_buf, 1),
                   kernel_overhead_us=kernel_overhead_us, buffer_bricks=cap, dense_bytes=4 * param.numel(),
                   buffer_bytes=4 * cap * BRICK ** 3)
        return out

    def _disarm(self, param) -> None:
        h = self._hints.get(id(param)) if param is not None else None
        if h is not None:
            h['armed'] = False

    def set_dense_source(self, param, active: bool) -> None:
        """Declare that this step adds a dense term to `param.grad` besides the trilinear scatter at the survivors (an
        autograd TV loss on the grid).  While active, `hint_touched` is ignored for the parameter and the brick occupancy
        is read from the gradient itself (or the exchange goes dense)."""
        self.dense_sources[id(param)] = bool(active)

    def attach(self, model) -> None:
        """Let the fused backward pass of `model` (fused.py) hand gradients over as soon as they are final (see `early`)."""
        if self.world_size > 1 or self.force:
            # (issuing hint_touched from inside the forward, right after the survivor list exists, was measured: its side
            # stream then competes with the persistent MLP kernel and the step gets 0.09 ms slower, not faster)
            model.__dict__.setdefault('_fused_cache', {})['grad_hook'] = self.early

    def rebind(self, params) -> None:
        """The model replaced its parameters (scale_volume_grid builds new grids): average the new ones from now on."""
        self.params = [p for p in params if p.requires_grad]
        self._hints.clear()
        self._deferred.clear()
        # everything keyed by id(param): the old grids are freed and a new parameter may get a recycled id -- a stale capacity
        # (sized from the smaller grid's union count) would overflow at once, a stale dense_sources entry would suppress hints
        self._static.clear()
        self.max_union_bricks.clear()
        self.dense_sources.clear()

    def attach_optimizer(self, optimizer) -> None:
        """Let the optimizer wait for an early exchange only when it reaches that parameter (MaskedAdam.before_param):
        the TV pass and the updates of the other parameters then run under the tail of the k0 exchange."""
        if (self.world_size > 1 or self.force) and hasattr(optimizer, 'before_param'):
            optimizer.before_param = self.wait_for
            self.defer_to_optimizer = True

    def wait_for(self, param) -> None:
        ev = self._deferred.pop(id(param), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def wait_all(self) -> None:
        for ev in self._deferred.values():
            torch.cuda.current_stream().wait_event(ev)
        self._deferred.clear()

    # ------------------------------------------------------------------------------------------------ early exchange
    def early(self, kind: str, params, tensor: Optional[torch.Tensor] = None) -> None:
        """Called from INSIDE the fused backward pass (fused.py) as soon as a group of gradients is final, so that its
        exchange runs on a side stream under the kernels the backward pass still has to launch:
          early('k0',  [k0 param],  grad)  after the feature-grid scatter, which the backward pass runs right behind the
                                           data-gradient chain: the weight-gradient launch (~440 us) and ~175 us of sdf
                                           scatter kernels follow;
          early('mlp', mlp params,  flat)  after the weight-gradient launch; `flat` is the one buffer all MLP gradients are
                                           views of, so a single in-place all-reduce replaces the bucket pack / unpack;
          early('join', ...)               before the backward pass copies anything out of `flat`: waits for the 'mlp'
                                           exchange only.
        `average()` then skips these parameters; the main stream waits for the k0 exchange at the end of `average()`, or,
        with `attach_optimizer`, when the optimizer reaches k0."""
        if (self.world_size == 1 and not self.force) or tensor is None and kind != 'join':
            return
        inv = 1.0 / self.world_size
        st = self.__dict__.setdefault('_early', dict(stream=None, mlp_done=None, params=set()))
        if kind == 'join':
            if st['mlp_done'] is not None:
                torch.cuda.current_stream().wait_event(st['mlp_done'])
                st['mlp_done'] = None
            return
        if not tensor.is_cuda:
            return
        if st['stream'] is None:
            # high priority: HIP maps streams onto a few hardware queues, and a second normal-priority stream can land on
            # the queue of the main stream, where its kernels serialise with the backward pass (seen in a kernel trace:
            # gather / reduce / scatter of the k0 exchange sitting between the scatter kernels of the main stream)
            prio = int(os.environ.get("FGS_DIST_STREAM_PRIORITY", "-1"))
            st['stream'] = torch.cuda.Stream(device=tensor.device, priority=prio)
        ready = torch.cuda.Event()
        ready.record()
        with torch.no_grad(), torch.cuda.stream(st['stream']):
            st['stream'].wait_event(ready)
            if kind == 'k0':
                g = tensor
                ok = (g.numel() >= self.sparse_min_numel and g.dim() == 5 and g.shape[1] > 1
                      and self._sparse(g, inv, params[0], group=self.early_group))
                if not ok:
                    self._disarm(params[0])
                    _, flat = self._dense(g, async_op=False, group=self.early_group)
                    self._post_scale(flat, inv)
            else:
                dist.all_reduce(tensor, op=self._op(), group=self.early_group)
                self._post_scale(tensor, inv)
            updated = False
            if kind == 'k0' and self.after_early is not None:
                updated = bool(self.after_early(params[0], tensor))   # the optimizer's pass over k0, same stream
            done = torch.cuda.Event()
            done.record()
        tensor.record_stream(st['stream'])
        st['params'].update(id(p) for p in params)
        if kind == 'mlp':
            st['mlp_done'] = done                       # waited for by early('join') at the end of the backward pass
        elif not updated:                               # (an early update is waited for by optimizer.step() itself)
            for p in params:
                self._deferred[id(p)] = done

    # ------------------------------------------------------------------------------------------------ driver
    @torch.no_grad()
    def average(self) -> None:
        if self.world_size == 1 and not self.force:
            return
        inv = 1.0 / self.world_size
        # Collectives of ONE communicator must start in the same order on every rank.  Synchronous collectives sit on the stream
        # they are issued from (and, in a captured step, on that stream's graph branch: scripts/diag/rccl_capture_order_probe.py);
        # the occupancy all-reduce of hint_touched() used `self.group` on a side stream, and the collectives below use
        # `self.group` from this one.  RCCL chains the launches of a communicator itself, but the order is cheaper to state than
        # to rely on: this stream waits for that all-reduce first (an event that completed a whole backward pass ago).
        for h in self._hints.values():
            if h.pop('unordered', False) and os.environ.get("FGS_DIST_ORDER_EDGE", "1") == "1":
                torch.cuda.current_stream().wait_event(h['event'])
        handles, small, sparse_later, dense_big = [], [], [], []
        owner = {}
        early = self.__dict__.get('_early')
        skip = set()
        if early is not None and early['params']:
            skip, early['params'] = early['params'], set()
        for p in self.params:
            g = p.grad
            if g is None or id(p) in skip:
                continue
            if g.numel() >= self.big_numel:
                if not (g.is_contiguous() or (g.dim() == 5 and g.is_contiguous(memory_format=torch.channels_last_3d))):
                    g = g.contiguous()
                    p.grad = g
                if g.numel() >= self.sparse_min_numel and g.dim() == 5 and g.shape[1] > 1:
                    sparse_later.append(g)
                    owner[id(g)] = p
                elif self._sparse_1ch(g) and (self.sparse_1ch_eager or id(p) in self._static):
                    sparse_later.append(g)
                    owner[id(g)] = p
                else:
                    dense_big.append(g)
            else:
                small.append(g)
        # one dense grid (the sdf gradient of a fine-stage step): a synchronous collective on THIS stream -- the asynchronous form
        # goes through the process group's internal stream, i.e. two cross-queue joins (~10 us of idle device each in a captured
        # step) around a collective that has nothing to overlap with here; several: asynchronous, so that they overlap each other
        for g in dense_big:
            if len(dense_big) == 1 and g.is_cuda and os.environ.get("FGS_DIST_DENSE_SYNC", "1") == "1":
                _, flat = self._dense(g, async_op=False)
                self._post_scale(flat, inv)
            else:
                handles.append(self._dense(g, async_op=True))
        if small:
            n = sum(g.numel() for g in small)
            if self._bucket is None or self._bucket.numel() != n or self._bucket.device != small[0].device:
                self._bucket = torch.empty(n, dtype=small[0].dtype, device=small[0].device)
            torch.cat([g.reshape(-1) for g in small], out=self._bucket)                    # one launch in
            dist.all_reduce(self._bucket, op=self._op(), group=self.group)
            self._post_scale(self._bucket, inv)
            views, off = [], 0
            for g in small:
                views.append(self._bucket[off:off + g.numel()].view(g.shape))
                off += g.numel()
            torch._foreach_copy_(small, views)                                             # a few launches out
        for g in sparse_later:
            one_ch, prev = g.shape[1] == 1, self.last_sparse_fill
            ok = self._sparse(g, inv, owner.get(id(g)))
            if one_ch:       # (`last_sparse_fill` keeps describing the multi-channel exchange)
                self.last_sparse_fill_1ch, self.last_sparse_fill = self.last_sparse_fill, prev
            if not ok:
                self._disarm(owner.get(id(g)))
                h, flat = self._dense(g, async_op=False)
                self._post_scale(flat, inv)
        for h, flat in handles:
            h.wait()
            self._post_scale(flat, inv)
        if not self.defer_to_optimizer:
            self.wait_all()
