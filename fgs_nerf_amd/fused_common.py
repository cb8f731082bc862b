"""What the fused fine- and coarse-stage paths share (fused_fine.py, fused_coarse.py; `fused` re-exports all of it): host copies
of the model geometry, the per-call run record and its device-resident values (fgs_dyn_t), the sync-free switch, roofline
timing hooks, the MLP launch helpers (GEMM wrappers, the forked weight-gradient launch, the exchange hooks), the persistent
self-cleaning feature-grid gradient buffer, the per-ray record workspace, the in-backward optimizer hook, the lazy result dict.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import fused_ops as fo
from ._lib import call, dyn, ptr, stream
from .ops import grid_strides

F32, I64, I32 = torch.float32, torch.int64, torch.int32

# event pairs recorded around the dominant kernel family (the MLP GEMMs) when profiling is switched on by bench.py
PROFILE = {"enabled": False, "gemm_events": [], "open": None}      # see set_profiling()


def _f32(x) -> float:
    """The fp32 value of a python / tensor scalar, as a python float."""
    return float(torch.as_tensor(x, dtype=F32))


class _Geom:
    """Host copies of the model geometry (cached on the model; refreshed when the grid is rescaled)."""

    def __init__(self, model):
        self.lo = model.xyz_min.detach().cpu().float().numpy().copy()
        self.hi = model.xyz_max.detach().cpu().float().numpy().copy()
        self.lo_c = (ctypes.c_float * 3)(*self.lo.tolist())
        self.hi_c = (ctypes.c_float * 3)(*self.hi.tolist())
        self.X, self.Y, self.Z = (int(s) for s in model.sdf.grid.shape[2:])
        self.voxel_size = _f32(model.voxel_size)
        self.diag = float(np.linalg.norm(self.hi.astype(np.float64) - self.lo.astype(np.float64)))
        self.mask = None
        if model.mask_cache is not None:
            mc = model.mask_cache
            mlo = mc.xyz_min.detach().cpu().float().numpy()
            mhi = mc.xyz_max.detach().cpu().float().numpy()
            self.mask = ((ctypes.c_float * 3)(*mlo.tolist()), (ctypes.c_float * 3)(*mhi.tolist()),
                         tuple(int(s) for s in mc.sdf_mask.shape[2:]), float(mc.mask_cache_thres))


def _geom(model) -> _Geom:
    # keyed on the grid shape AND on the identity / in-place version of everything _Geom copies to the host (a new bbox or
    # voxel size with an unchanged grid shape must not serve stale lo / hi / voxel_size); the keyed objects are kept alive
    # by the cache entry so that an id() cannot be reused by a successor
    objs = (model.voxel_size, model.xyz_min, model.xyz_max, model.mask_cache)
    key = (tuple(model.sdf.grid.shape),) + tuple((id(o), getattr(o, '_version', 0)) for o in objs)
    g = getattr(model, '_fused_geom', None)
    if g is None or getattr(model, '_fused_geom_key', None) != key:
        g = _Geom(model)
        g._keyed = objs
        model._fused_geom, model._fused_geom_key = g, key
    return g


class _Run:
    """Everything one forward produced that the backward needs (plain attribute bag)."""


def _rows(run):
    """Device address of the survivor count of a sync-free run (fgs_dyn_t.row_count), or None: the per-survivor entry points then
    take their host row count as the CAPACITY of the buffers and read the actual count from the device."""
    return run.count_ptr if run.sync_free else None


def _inv_s(run):
    """Device address of NeuS 1/s of a sync-free run whose schedule lives on the device (a captured step), or None."""
    sf = run.sync_free
    return ptr(sf['inv_s_dev']) if (sf and sf.get('inv_s_dev') is not None) else None


def set_sync_free(model, capacity=None, inv_s_dev=None) -> None:
    """Switch the fused path of `model` (fine or coarse stage) to the sync-free form (or back, with capacity=None): the survivor count
    is never read by the host; result tensors, activations and gradients of the survivors are allocated for `capacity` rows
    and every kernel clamps to the device-side count; a device-side guard records a count above the capacity (see
    `sync_free_state`) and makes the optimizer skip that step.  `inv_s_dev`: optional 1-element float32 device tensor the
    march kernels read 1/s from (a captured step cannot pass the iteration-dependent s_val by value).
    Needs the register-resident MLP path (FGS_MLP=rc, the default) -- the split-K GEMMs partition by a host count."""
    cache = model.__dict__.setdefault('_fused_cache', {})
    if capacity is None:
        cache.pop('sync_free', None)
        return
    dev = model.sdf.grid.device
    buf = cache.get('sync_free_buffers')      # the guard's counters live as long as the model (captured kernels point at them)
    if buf is None:
        buf = cache['sync_free_buffers'] = dict(flags=torch.zeros(2, dtype=torch.int32, device=dev),
                                                total=torch.zeros(1, dtype=I64, device=dev))
    cache['sync_free'] = dict(capacity=int(capacity), inv_s_dev=inv_s_dev, flags=buf['flags'], total=buf['total'])


def sync_free_state(model):
    """(overflowed: bool, survivors_processed: int) since the counters were last cleared -- ONE device->host read; call it
    at a logging interval, not per step."""
    st = model.__dict__.get('_fused_cache', {}).get('sync_free_buffers')
    if st is None:
        return False, 0
    flags, total = st['flags'].cpu(), st['total'].cpu()
    return bool(flags[0]), int(total[0])


def _detached(d):
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()}


def set_profiling(on: bool, clear: bool = False) -> None:
    """bench.py's switch for the roofline timing: HIP events around the matrix-core launches (fused_ops._timed for the
    one-launch kernels of the default path, _gemm_group for the per-product k_gemm chains of the other paths)."""
    PROFILE["enabled"] = bool(on)
    fo.TIMING["enabled"] = bool(on)
    if clear:
        PROFILE["gemm_events"].clear()
        fo.TIMING["events"].clear()


class _gemm_group:
    """HIP-event bracket around an uninterrupted run of k_gemm launches (the forward chain, the backward chain): two
    events per chain instead of two per launch -- 42 event records per step cost ~0.5 ms of launch latency."""

    def __init__(self, label):
        self.label = label

    def __enter__(self):
        if PROFILE["enabled"]:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            PROFILE["open"] = [0, 0.0, None]     # launches, algorithmic FLOP, side stream used by the chain (or None)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE["enabled"] and PROFILE.get("open") is not None:
            self.e1.record()
            n, fl, side = PROFILE["open"]
            PROFILE["open"] = None
            e1s = None
            if side is not None:                 # the chain also ran launches on a side stream: it ends when both ends do
                e1s = torch.cuda.Event(enable_timing=True)
                e1s.record(side)
            if n:
                PROFILE["gemm_events"].append((self.e0, self.e1, e1s, self.label, n, fl))
        return False


def _gemm(op, A, B, C, M, N, K, logical=None, **kw):
    """`logical` = un-padded (M, N, K) of the product, for the algorithmic FLOP count of the roofline report."""
    lm, ln, lk = logical or (M, N, K)
    if _MLP_IMPL == "rc":
        kw = dict(kw, stamp=("k_gemm (first-layer data gradients)", 2.0 * lm * ln * lk))   # (fused_ops.STAMPS, when on)
    if _MLP_IMPL == "rc" and fo.TIMING["enabled"]:      # the rc path times every launch on its own (fused_ops._timed)
        fo._timed("k_gemm (first-layer data gradients)", 2.0 * lm * ln * lk, lambda: fo.gemm(op, A, B, C, M, N, K, **kw))
        return
    fo.gemm(op, A, B, C, M, N, K, **kw)
    grp = PROFILE.get("open")
    if grp is not None:
        grp[0] += 1
        grp[1] += 2.0 * lm * ln * lk


# How the two products of a Linear layer's backward are issued (FGS_LINEAR_BWD), measured ms/step fine / coarse:
#   "one"     (default) both products in one k_linear_bwd launch: the split-K weight-gradient workgroups fill the partly
#             occupied last round of data-gradient tiles                                              2.58 / 1.65
#   "split"   two k_gemm launches on the main stream                                                  2.63 / 1.60
#   "overlap" data gradient on the main stream, weight gradient on a side stream as soon as its dY exists: the two
#             launches of a layer run concurrently (each ~175 us instead of 98 + 89)                  2.55 / 1.62
#   "late"    weight gradients on the side stream after the whole data-gradient chain, under the atomics-bound scatter
#             kernels of the feature / march backward                                                 2.61-2.9 / 1.52
#   "chain"   (fine stage) every data gradient in ONE persistent k_mlp_fwd<true> launch on transposed weights (ReLU masks
#             and bias-gradient column sums in its epilogues), then 7 weight-gradient k_gemm launches: the chain takes
#             655 us and each weight gradient 91 us (1340 us with the encodings' columns, vs 7 x 181 = 1266)  2.65 / -
# The differences are within 4 %; "one" is the default because every launch then runs alone and per-kernel durations
# in a trace mean what they say.
_LINEAR_BWD_MODE = os.environ.get("FGS_LINEAR_BWD", "one")
# forward chain of the fine stage: one persistent k_mlp_fwd launch (default) or one k_gemm launch per layer (FGS_MLP_FWD=layers)
_MLP_FWD_ONE_LAUNCH = os.environ.get("FGS_MLP_FWD", "one") == "one"
# MLP kernels: "rc" (default) = register-resident chains (csrc/mlp_rc.hip: forward chain and backward data-gradient chain, one
# launch each) + every weight / bias gradient in one launch (csrc/mlp_wgrad.hip); "lds" = the LDS-resident forward chain and
# one k_linear_bwd launch per layer (csrc/mlp_fused.hip, gemm_f32.hip)
_MLP_IMPL = os.environ.get("FGS_MLP", "rc")
_SIDE = {}   # device index -> (side stream, list of tensors to keep alive until the join)


def _side(dev):
    st = _SIDE.get(dev.index)
    if st is None:
        st = (torch.cuda.Stream(device=dev), [])
        _SIDE[dev.index] = st
    return st


def _linear_bwd(dY, W, X, dX, dW, M, n_out, k_in, mask=None, colsum=None, logical_k_in=None):
    """Data- and weight-gradient product of one Linear layer (see _LINEAR_BWD_MODE).  `logical_k_in`: un-padded input
    width for the algorithmic FLOP count of the roofline report."""
    lk = logical_k_in or k_in
    if _LINEAR_BWD_MODE == "one":
        fo.linear_bwd(dY, W, X, dX, dW, M, n_out, k_in, mask=mask, colsum=colsum)
        grp = PROFILE.get("open")
        if grp is not None:
            grp[0] += 1
            grp[1] += 4.0 * M * n_out * lk
        return
    if _LINEAR_BWD_MODE == "overlap":
        side, keep = _side(dY.device)
        ready = torch.cuda.Event()
        ready.record()                         # dY (and the zero-filled dW) exist on the main stream from here on
        with torch.cuda.stream(side):
            side.wait_event(ready)
            fo.gemm(fo.GEMM_TN, dY, X, dW, n_out, k_in, M)
        keep.extend((dY, X, dW))
        grp = PROFILE.get("open")
        if grp is not None:                      # counted in the chain; the chain's end is the later of the two streams
            grp[0] += 1
            grp[1] += 2.0 * n_out * lk * M
            grp[2] = side
    elif _LINEAR_BWD_MODE == "late":
        _side(dY.device)[1].append((dY, X, dW, n_out, k_in, M))   # issued by _flush_tn() after the data-gradient chain
    else:
        _gemm(fo.GEMM_TN, dY, X, dW, n_out, k_in, M, logical=(n_out, lk, M))
    _gemm(fo.GEMM_NN, dY, W, dX, M, k_in, n_out, mask=mask, colsum=colsum, logical=(M, lk, n_out))



# FGS_MLP_FORM: which of the two one-launch chain kernels the fine stage uses when both cover the model (width 256): 2 (default) =
# csrc/mlp_rc2.hip, the waves split the features and a CU is dealt whole sample tiles (no round quantisation; the two narrow
# backward products ride along as side layers); 1 = csrc/mlp_rc.hip, activations resident in registers (every other width).
_MLP_FORM = int(os.environ.get("FGS_MLP_FORM", "2"))


# FGS_RC2_HEAD=1: the 256 -> 3 output head rides in the form-2 forward chain as its last side layer (`side` = 2: bias, sigmoid)
# instead of the k_head_fwd launch.  Measured, round 4: the chain grows by 11 us (one feature tile still walks all 32 k-groups of
# both slabs), k_head_fwd and its launch were 17: 1.6566 against 1.6569 ms per step -- nothing; default off.
_RC2_HEAD = os.environ.get("FGS_RC2_HEAD", "0") == "1"


def _rc2_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) -> bool:
    """Shapes the feature-split chains cover: 256-wide trunks, <= 52 appended columns, <= 10 layers incl. the side layers."""
    return (_MLP_FORM == 2 and _rc_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) and rw == 256 and fw == 256 and ldz - rw <= 52
            and n_rgb + n_ref - 1 <= 8)


def _rc_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) -> bool:
    """Shapes the register-resident chains cover (a function of the model only: identical on every rank)."""
    return (_MLP_IMPL == "rc" and rw == fw and rw % 32 == 0 and rw <= 256 and ldx0 <= 256 and 0 < ldz - rw <= 64 and
            n_rgb + n_ref - 1 <= 8)


# The weight-gradient launch (k_mlp_wgrad: 57 + 256 registers per lane, one 256-thread workgroup per CU, 132 KB of LDS, matrix
# pipe busy) needs nothing that the rest of the backward pass produces and nothing after it needs its result before the
# optimizer: with FGS_WGRAD_FORK=1 (default) it goes to a side stream behind the data-gradient chain and the narrow products,
# and the gather / scatter / atomics-bound kernels that follow on the main stream (feature backward, march backward, sdf
# scatter: ~200 us of memory latency, < 192 registers, < 25 KB of LDS) take the free issue slots of the same SIMDs.  In a
# captured step the fork / join become graph edges.  Not with a gradient exchange attached (the MLP gradients are exchanged
# from inside the backward pass there).
_WGRAD_FORK = os.environ.get("FGS_WGRAD_FORK", "1") == "1"
# FGS_MARCH_FIRST=1: the vector-bound kernels of the sdf path (encoding backward, march backward) are issued BEFORE the fork and
# only the memory-bound ones beside the weight-gradient launch.  Measured 1.89-1.90 ms/step against 1.87 for the default order
# (the LDS-atomic sdf scatter, then under the matrix kernel for its whole length, costs it more than the march kernel saves).
_MARCH_FIRST = os.environ.get("FGS_MARCH_FIRST", "0") == "1"
# dX0 (d loss / d first-layer input) computed and read in compact form: without the columns of the xyz / view-direction encodings
_DX0_COMPACT = os.environ.get("FGS_DX0_COMPACT", "1") == "1"
# One GPU: where on the main branch k0's in-backward Adam pass (memory-bound; 35 us alone, ~180 us beside k_mlp_wgrad, whose
# registers and LDS leave its waves two slots per SIMD) is issued: right behind the feature-grid scatter (0), or as the branch's
# last kernel (1), where it mostly runs after the weight-gradient launch has drained.
_K0_ADAM_LATE = os.environ.get("FGS_K0_ADAM_LATE", "0") == "1"
# FGS_MLP_COLLAPSE=1 (a LABELLED mode, never the default: bench.py marks its line): rgbnet's last Linear has no activation and feeds
# refnet's first Linear (model/nerf.py:135-142,877-884), so  V0[:, :256] (W3 h + b3) + V0[:, 256:] e + c0  =  (V0a W3) h + V0b e +
# (V0a b3 + c0): ONE 256 x 256 layer with a per-step pre-multiplied weight instead of two -- 65 536 of 434 176 MAC per survivor in
# each of the forward, data-gradient and weight-gradient passes.  The gradients of the original parameters follow from the
# collapsed layer's by three 256^3 products per step (dW3 = V0a^T dWc, dV0a = dWc W3^T + dbc b3^T, db3 = V0a^T dbc, dc0 = dbc).
# Values differ from the reference order by float32 re-association only (tests/test_fullsize_parity_gpu.py passes unchanged).
_MLP_COLLAPSE = os.environ.get("FGS_MLP_COLLAPSE", "0") == "1"
_SIDE_PENDING = set()


def _wgrad(dev, M, items, flop, fork: bool, post=None, rows_dev=None) -> None:
    if not (fork and _WGRAD_FORK):
        fo.mlp_wgrad(M, items, flop=flop, rows_dev=rows_dev)
        if post is not None:
            post()
        return
    side, keep = _side(dev)
    ready = torch.cuda.Event()
    ready.record()                      # dY tensors, layer inputs and the zero-filled gradient buffer exist from here on
    with torch.cuda.stream(side):
        side.wait_event(ready)
        fo.mlp_wgrad(M, items, flop=flop, rows_dev=rows_dev)
        if post is not None:
            post()                      # (_MLP_COLLAPSE: the original parameters' gradients from the collapsed layer's)
    # allocated on the main stream: alive until the join.  Only the launch's INPUTS: a held gradient view would make autograd's
    # AccumulateGrad clone the gradient it is handed (16 copy launches) when the join comes after the backward pass (below).
    keep.append([(it[0], it[1]) for it in items])
    _SIDE_PENDING.add(dev.index)


# With a gradient exchange attached the weight-gradient launch goes to the side branch as well (issued BEFORE the feature-grid
# scatter, like on one GPU), and the exchange of the MLP gradients is issued from that branch -- it needs that launch's result
# and nothing else -- AFTER the host has issued k0's exchange: collectives of one communicator execute in issue order, and
# k0's (the long one) must not queue behind a collective that waits ~430 us for the weight-gradient launch.
# FGS_WGRAD_FORK_DIST=0: weight gradients and their exchange on the main stream, in issue order (the round-2 form).
_WGRAD_FORK_DIST = os.environ.get("FGS_WGRAD_FORK_DIST", "1") == "1"


def _exchange_mlp(dev, wgrad, forked, hook, mlp, flat) -> None:
    if forked:
        side, keep = _side(dev)
        with torch.cuda.stream(side):
            hook('mlp', mlp, flat)
        keep.append(flat)
        return
    if wgrad is not None:
        wgrad(False)
    hook('mlp', mlp, flat)


def _flush_tn(dev) -> None:
    """"late" mode: all weight-gradient products on the side stream, started when the data-gradient chain is done, so
    that they run under the atomics-bound scatter kernels that follow on the main stream."""
    if _LINEAR_BWD_MODE != "late":
        return
    side, jobs = _side(dev)
    ready = torch.cuda.Event()
    ready.record()
    with torch.cuda.stream(side):
        side.wait_event(ready)
        for dY, X, dW, n_out, k_in, M in jobs:
            fo.gemm(fo.GEMM_TN, dY, X, dW, n_out, k_in, M)


# FGS_DEFER_WGRAD_JOIN=1 (default): a captured single-GPU step (graph_step.CapturedFineStep sets cache['defer_side_join']) joins
# the weight-gradient branch in front of the MLP's Adam launch instead of at the end of the backward pass.  The scatter branch
# is the longer one since round 4, so the join is already satisfied there, and sdf's TV + Adam passes follow the scatter kernels
# on their own queue: the cross-queue wait in front of them cost ~10 us of idle device per step.  0: join inside backward.
_DEFER_WGRAD_JOIN = os.environ.get("FGS_DEFER_WGRAD_JOIN", "1") == "1"


def _join_side_or_defer(run, dev, allowed: bool) -> None:
    if allowed and _DEFER_WGRAD_JOIN and run.cache.get('defer_side_join') and dev.index in _SIDE_PENDING:
        run.cache['side_join_pending'] = dev
        return
    _join_side(dev)


def join_pending_side(model) -> None:
    """The deferred join of the weight-gradient branch (see _DEFER_WGRAD_JOIN): called by the step that asked for the deferral,
    in front of the first consumer of the MLP gradients."""
    cache = model.__dict__.get('_fused_cache', {})
    dev = cache.pop('side_join_pending', None)
    if dev is not None:
        _join_side(dev)


def _join_side(dev) -> None:
    """Main stream waits for the weight-gradient launches on the side stream (before the gradients are handed back)."""
    if _LINEAR_BWD_MODE not in ("overlap", "late") and dev.index not in _SIDE_PENDING:
        return
    _SIDE_PENDING.discard(dev.index)
    side, keep = _side(dev)
    done = torch.cuda.Event()
    done.record(side)
    torch.cuda.current_stream().wait_event(done)
    keep.clear()


_PRE_FILL_AT_READ = os.environ.get("FGS_PRE_FILL", "read") == "read"


def _count_begin(run, offsets, n):
    """The one host read of a step, first half: offsets[n] (the survivor count) starts travelling to pinned memory.  Work
    queued between _count_begin and _count_end sits BEHIND the copy in the stream: the device executes it during the
    ~50 us the host needs to wake up from the wait and launch the next kernels, instead of idling -- the weight pads and
    the largest zero fill of the backward pass (k0.grad, 197 MB at 160^3) go there."""
    if not _PRE_FILL_AT_READ:
        return None, offsets, n
    host = run.cache.get('count_host')
    if host is None:
        host = run.cache['count_host'] = torch.empty(1, dtype=I64).pin_memory()
    host.copy_(offsets[n:n + 1], non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    return done, host, 0


def _count_end(token) -> int:
    done, src, i = token
    if done is None:
        return int(src[i].item())
    done.synchronize()
    return int(src[i])


def _zeros_like_strided(t):
    return torch.empty_strided(t.shape, t.stride(), dtype=F32, device=t.device).zero_()


# ---- the feature grid's gradient: one persistent, self-cleaning buffer instead of a fresh zero-filled one per step ----------
# Rays touch a thin shell of the feature grid, yet a step used to zero-fill all of k0.grad (197 MB at 160^3, 1.57 GB at
# 320^3) and MaskedAdam then read all of it back to find the few non-zero elements (model/adam.py:205-221 has no other way to
# know).  Here the backward pass scatters into a buffer that is all-zero by construction, records the voxels the survivors'
# trilinear corners fall on (fgs_brick_masks_pts: a 64-bit mask per 4x4x4-voxel brick), and MaskedAdam's update of this tensor
# visits those voxels only and zeroes what it consumed (fgs_adam_upd_voxels; after a multi-GPU exchange: the union's bricks,
# fgs_adam_upd_bricks).  Anything that breaks the "non-zero only inside the recorded
# bricks" invariant (a dense TV term, an autograd accumulation into the same tensor, a dense gradient exchange) is detected
# or declared (`_fgs_touched['valid']`, tensor version, storage use count) and falls back to dense update + zero fill.
# (switches a test may flip at run time live in a dict: the three modules of the fused path share the OBJECT, not a copy of a name)
FLAGS = {"brick_adam": os.environ.get("FGS_BRICK_ADAM", "1") != "0",       # persistent self-cleaning k0 gradient buffer
         "coarse_vol4": os.environ.get("FGS_COARSE_VOL4", "1") != "0"}     # coarse march samples the voxel-interleaved volume


def _storage_users(t) -> int:
    try:
        return int(torch._C._storage_Use_Count(t.untyped_storage()._cdata))
    except Exception:       # private API: without it the buffer is never reused while anything could still alias it
        return 1 << 30


def _grid_grad_state(cache, k0_grid, create: bool):
    key = (tuple(k0_grid.shape), tuple(k0_grid.stride()), k0_grid.device)
    gb = cache.get('k0_grad')
    if gb is not None and gb['key'] == key:
        return gb
    if not create or not FLAGS['brick_adam']:
        return None
    _, C, X, Y, Z = k0_grid.shape
    if k0_grid.stride() != (C * X * Y * Z, 1, Y * Z * C, Z * C, C) or C % 4 or k0_grid.dtype != F32 or min(X, Y, Z) < 2:
        return None           # not channel-last / channel count not float4-able: the plain path
    buf = _zeros_like_strided(k0_grid)
    # 64 bytes per 4x4x4-voxel brick: which of its voxels hold a trilinear corner of a survivor (fgs_brick_masks_pts)
    flags = torch.zeros(((X + 3) // 4) * ((Y + 3) // 4) * ((Z + 3) // 4) * 64, dtype=torch.uint8, device=k0_grid.device)
    gb = cache['k0_grad'] = dict(key=key, buf=buf, flags=flags, clean=True, dims=(C, X, Y, Z), base_users=None)
    gb['base_users'] = _storage_users(buf)
    return gb


def _grid_grad_idle(gb) -> bool:
    """Nobody but the cache holds the buffer (last step's p.grad has been dropped)."""
    return gb is not None and _storage_users(gb['buf']) <= gb['base_users']


def _take_grid_grad(cache, k0_grid):
    """(gradient tensor to scatter into -- all zero --, state or None).  With a state, the tensor aliases the persistent
    buffer (a detached alias: autograd's AccumulateGrad adopts it as p.grad without a copy)."""
    gb = _grid_grad_state(cache, k0_grid, create=True)
    if gb is None:
        return _zeros_like_strided(k0_grid), None
    owner = gb.get('pending')
    if owner is not None and not cache.get('sync_free'):
        # a captured step left its last k0 gradient in the buffer for the head of its next replay (CapturedFineStep, deferred
        # Adam pass): an eager backward pass in between would scatter on top of it
        raise RuntimeError("the persistent k0 gradient buffer holds the pending update of a captured step: call its flush() "
                           "(or check()) before an eager training step")
    if not _grid_grad_idle(gb):
        # somebody still holds the old buffer (last step's p.grad before zero_grad, a gradient being accumulated): it is
        # theirs now; a fresh zero-filled tensor becomes the persistent buffer
        gb['buf'] = _zeros_like_strided(k0_grid)
        gb['base_users'] = _storage_users(gb['buf'])
        if not gb['clean']:
            gb['flags'].zero_()
    elif not gb['clean']:
        gb['buf'].zero_()
        gb['flags'].zero_()
    gb['clean'] = False
    return gb['buf'].detach(), gb


def _publish_touched(gb, k0_grid, grad_k0, pts, M, g, st, exchange: bool, rows_dev=None):
    """Record which bricks `grad_k0` can be non-zero in and attach the record to the parameter for MaskedAdam
    (adam.MaskedAdam._bricks).  `exchange`: a gradient exchange follows (dist.GradAverager): the union over ranks then
    replaces the local occupancy, or invalidates the record if the exchange goes dense."""
    if gb is None:
        k0_grid._fgs_touched = None
        return
    C, X, Y, Z = gb['dims']
    call("fgs_brick_masks_pts", ptr(pts), M, g.lo_c, g.hi_c, X, Y, Z, ptr(gb['flags']), dyn(row_count=rows_dev), st)
    k0_grid._fgs_touched = dict(state=gb, grad_ptr=grad_k0.data_ptr(), version=gb['buf']._version, dims=gb['dims'],
                                flags=gb['flags'], idx=None, n=None, valid=True, exchange=exchange)


def _prefill_grid_grad(run, k0_grid):
    """Forward-time half of the k0.grad preparation (the slot behind the survivor-count copy, see _count_begin): a clean
    buffer needs nothing now (the backward pass takes it, and a forward pass that is never differentiated costs nothing);
    anything else is taken -- i.e. zero-filled -- here, where the fill is free."""
    gb = _grid_grad_state(run.cache, k0_grid, create=True)
    if gb is not None and gb['clean']:
        # (not necessarily idle yet: the reference's loop drops last step's gradients -- optimizer.zero_grad(set_to_none=True),
        # model/nerf_training.py:374 -- between this forward pass and backward)
        return None
    return _take_grid_grad(run.cache, k0_grid)


def reset_grid_grad(model, force: bool = False) -> None:
    """Bring the persistent feature-grid gradient buffer back to all-zero (after a backward pass whose gradient no
    optimizer step consumed, before capturing a step in a hipGraph).  `force`: also when the host-side record says "clean"
    (after a device-counted exchange overflowed inside a captured step, which the host-side record cannot know)."""
    gb = model.__dict__.get('_fused_cache', {}).get('k0_grad')
    if gb is not None and (force or not gb['clean']):
        gb['buf'].zero_()
        gb['flags'].zero_()
        gb['clean'] = True


def _head_scratch(width, dev):
    """Per-workgroup partial sums of fgs_head_bwd (4 MB at width 256); uninitialised, consumed inside the same call."""
    from ._lib import lib
    return torch.empty(int(lib().fgs_head_bwd_scratch_floats(int(width))), dtype=F32, device=dev)


def _seam(run, name, **tensors) -> None:
    """Stage seam of the fine-stage backward pass.  A test may install `model._fused_cache['bwd_probe'] = f(name, tensors)`: it is
    called with the tensors that cross the seam, right after the launches that produced them were issued, and may read them
    (clone) or overwrite them in place (copy_) -- e.g. with the CPU oracle's gradient at the same seam, so that the NEXT stage
    runs on exactly the oracle's upstream gradient (tests/test_stagewise_bwd_gpu.py).  No probe: nothing happens."""
    probe = run.cache.get('bwd_probe')
    if probe is not None:
        probe(name, tensors)


def _early_hooks(run):
    hook = run.cache.get('grad_hook') if _LINEAR_BWD_MODE in ("one", "split", "chain") else None   # dist.GradAverager.early
    return hook, run.cache.get('opt_hook')


def _workspace(model, n_rays: int, max_steps: int, dev) -> Dict[str, torch.Tensor]:
    """Per-(n_rays, max_steps) record arrays, cached on the model: no allocator traffic in the steady state.

    The records of a forward are re-read by its backward.  A forward whose backward is still pending OWNS the set it
    wrote (`ws['owner']`, a weak reference to its run): a second forward with the same ray count before that backward
    (a loss over two batches, gradient accumulation, a validation render while the graph is alive) gets a fresh set
    instead of overwriting it -- the reference's autograd graph owns its saved tensors the same way."""
    key = (n_rays, max_steps, str(dev))
    cache = model.__dict__.setdefault('_fused_ws', {})
    ws = cache.get(key)
    if ws is not None:
        owner = ws['owner']() if ws.get('owner') is not None else None
        if owner is not None and not owner.done:
            ws = None
    if ws is None:
        rec = n_rays * max_steps
        ws = dict(a_step=torch.empty(rec, dtype=I32, device=dev), a_alpha=torch.empty(rec, dtype=F32, device=dev),
                  a_T=torch.empty(rec, dtype=F32, device=dev), a_weight=torch.empty(rec, dtype=F32, device=dev),
                  a_sdf=torch.empty(rec, dtype=F32, device=dev), a_grad=torch.empty(rec * 3, dtype=F32, device=dev),
                  a_surv=torch.empty(rec, dtype=I32, device=dev), surv_slot=torch.empty(rec, dtype=I32, device=dev),
                  n_alive=torch.empty(n_rays, dtype=I64, device=dev), n_surv=torch.empty(n_rays, dtype=I64, device=dev),
                  n_inbbox=torch.empty(n_rays, dtype=I64, device=dev),
                  surv_off=torch.empty(n_rays + 1, dtype=I64, device=dev))
        ws['owner'], ws['gen'] = None, 0
        cache.clear()            # keep one shape resident
        cache[key] = ws
    ws['gen'] += 1               # one generation per forward: late readers (lazy 'mask') check they still see their own
    return ws


def _own_workspace(run, needs_grad: bool) -> None:
    """Called by the forward pass: the run keeps its record set until its backward has run (see _workspace)."""
    import weakref
    run.done = not needs_grad
    run.gen = run.workspace['gen']
    if needs_grad:
        run.workspace['owner'] = weakref.ref(run)


def enable_early_update(model, optimizer, averager=None, inline: bool = False) -> None:
    """Let `optimizer` (MaskedAdam) update the feature grid from inside the fused backward pass, right after the grid's
    gradient is final -- on several GPUs right after that gradient's exchange, on the exchange stream.  The ~45 us Adam
    pass over k0 (and the wait for its exchange) then leave the end of the step.  Only for steps in which nothing else
    writes into k0.grad (no TV on k0); `disable_early_update` turns it off again."""
    cache = model.__dict__.setdefault('_fused_cache', {})
    if averager is not None and (averager.world_size > 1 or averager.force):
        averager.after_early = lambda p, g: optimizer.early_update(p, g, on_stream=True)
    elif inline:
        # one GPU: issued in place, on the backward pass's own stream, right behind the feature-grid scatter and the voxel
        # marking -- i.e. beside the weight-gradient launch running on the side branch (_wgrad), instead of at the end of the
        # step behind it.  (On a stream of its own, high priority, the same pass made every kernel of a captured step slower.)
        cache['opt_hook'] = lambda p, g: optimizer.early_update(p, g, on_stream='inline')
    else:
        cache['opt_hook'] = optimizer.early_update


# FGS_FUSED_LOSS=1 (default): a training step that announces its loss (set_loss_spec) gets compositing, the loss terms, their
# gradients and the compositing backward in ONE launch (fgs_fine_render_loss) instead of five; 0: the separate launches
_FUSED_LOSS = os.environ.get("FGS_FUSED_LOSS", "1") == "1"


def set_loss_spec(model, target, loss_cfg, seed=None) -> None:
    """Announce the loss the NEXT fused fine-stage forward passes of `model` will be differentiated through: `target` [N,3] and
    the loss weights (the keys losses.fused_render_losses reads).  The forward pass then runs compositing + losses + their
    gradients + the compositing backward as one launch and hands the result to losses.fused_render_losses(res, target, cfg),
    which must be called with the same target tensor and weights (anything else: the ordinary path, nothing lost but the launch).
    `seed`: the device scalar the caller will pass to loss.backward(seed) (read by the kernel; None: 1).  None as target: off."""
    cache = model.__dict__.setdefault('_fused_cache', {})
    if target is None or not _FUSED_LOSS:
        cache.pop('loss_spec', None)
        return
    from .losses import _w5
    w5 = _w5(loss_cfg)
    cache['loss_spec'] = dict(target=target, w5=w5, w5_key=tuple(float(v) for v in w5), seed=seed)


def _composite(run, needs_grad, N, M, surv_off, weights, rgb, normal, step_id, alphainv_last, rgb_marched, sigmoid_rgb, pre_rgb,
               pre_sig, normal_marched, depth):
    """Compositing of a fused forward pass (both stages).  With an announced loss (set_loss_spec) whose target fits the batch: ONE
    launch for compositing + the loss terms + their gradients + the compositing backward (fgs_fine_render_loss), and the stash the
    loss node and the backward pass pick up; otherwise fgs_composite_fwd (returns None)."""
    dev, st = weights.device, stream()
    spec = run.cache.get('loss_spec')
    if (spec is not None and needs_grad and M > 0 and spec['target'].shape == (N, 3) and spec['target'].is_cuda
            and spec['target'].is_contiguous() and spec['target'].dtype == F32):
        from .losses import _loss_scratch
        fl = dict(loss=torch.empty((), dtype=F32, device=dev), d_out=torch.empty(M, 3, dtype=F32, device=dev),
                  d_w=torch.empty(M, dtype=F32, device=dev), g_normal=torch.empty(M, 3, dtype=F32, device=dev),
                  g_last=torch.empty(N, dtype=F32, device=dev), g_rm=torch.empty(N, 3, dtype=F32, device=dev),
                  target_ptr=spec['target'].data_ptr(), w5_key=spec['w5_key'], used=False,
                  seed_ptr=None if spec['seed'] is None else spec['seed'].data_ptr())
        scratch = _loss_scratch(dev, (N + 3) // 4 + 1)
        call("fgs_fine_render_loss", N, M, ptr(surv_off), ptr(weights), ptr(rgb), ptr(normal), ptr(step_id), run.bg,
             run.dist, ptr(run.viewdirs), ptr(spec['target']), ptr(alphainv_last), spec['w5'], ptr(spec['seed']),
             ptr(rgb_marched), ptr(sigmoid_rgb), ptr(pre_rgb), ptr(pre_sig), ptr(normal_marched), ptr(depth), ptr(fl['loss']),
             ptr(scratch), scratch.numel(), ptr(fl['d_out']), ptr(fl['d_w']), ptr(fl['g_normal']), ptr(fl['g_last']),
             ptr(fl['g_rm']), dyn(row_count=_rows(run)), st)
        return fl
    call("fgs_composite_fwd", N, ptr(surv_off), ptr(weights), ptr(rgb), ptr(normal), ptr(step_id), run.bg, run.dist,
         ptr(rgb_marched), ptr(sigmoid_rgb), ptr(pre_rgb), ptr(pre_sig), ptr(normal_marched), ptr(depth), st)
    return None


def _check_only_announced_loss(fl, g_rgb_marched, *others) -> None:
    """The backward pass is about to start from the stash of the one-launch loss: that is only right when the announced loss is the
    ONLY thing differentiated through this forward pass's outputs -- any other term (a loss on depth, on the weights ...) would hand
    its gradient to this node and be dropped.  Refuse instead."""
    if any(t is not None for t in others) or g_rgb_marched is None or g_rgb_marched.data_ptr() != fl['g_rm'].data_ptr():
        raise RuntimeError("fused.set_loss_spec: another term than the announced loss is differentiated through the render "
                           "result (only the announced loss may be: call set_loss_spec(model, None, cfg) to switch the "
                           "one-launch loss off for such a step)")


def disable_early_update(model, averager=None) -> None:
    model.__dict__.setdefault('_fused_cache', {}).pop('opt_hook', None)
    if averager is not None:
        averager.after_early = None


class LazyResult(dict):
    """ret_dict of forward_fine whose rarely used, expensive entries ('mask', 'mask_outbbox': per-sample masks over ALL
    emitted samples, which the fused kernels never materialise) are computed on first access."""

    def __init__(self, eager, lazy_fns):
        super().__init__(eager)
        self._lazy = dict(lazy_fns)
        for k in self._lazy:
            super().__setitem__(k, None)

    def __getitem__(self, k):
        if k in self._lazy:
            super().__setitem__(k, self._lazy.pop(k)())
        return super().__getitem__(k)

    def get(self, k, default=None):
        return self[k] if k in self else default


def _setup_run(model, rays_o, rays_d, viewdirs, global_step, render_kwargs, default_depth):
    """The per-call scalars both stages share; returns (run, s_val)."""
    run = _Run()
    run.cache = model.__dict__.setdefault('_fused_cache', {})    # per-model host-side constants (layouts, ...)
    run.geom = _geom(model)
    run.n_rays = N = len(rays_o)
    run.rays_o, run.rays_d = rays_o.contiguous().float(), rays_d.contiguous().float()
    run.viewdirs = viewdirs.contiguous().float()
    run.near = float(render_kwargs['near'])
    stepsize = render_kwargs['stepsize']
    # dist = stepsize * voxel_size in fp32 (model/nerf.py:795); stepdist (model/nerf.py:689) is the same value as a C float
    run.dist = float(np.float32(stepsize) * np.float32(run.geom.voxel_size))
    run.stepdist = run.dist
    run.bg = float(render_kwargs['bg'])
    run.thres = float(model.fast_color_thres)
    is_train = global_step is not None
    s_val = model._s_val_for(global_step, is_train)
    # inv_s = torch.ones(1) / self.s_val: one fp32 division (model/nerf.py:522); done on the host, no device read
    s32 = np.float32(s_val) if is_train else np.float32(getattr(model, '_s_val_host', model.s_start))
    model._s_val_host = float(s32)
    run.inv_s = float(np.float32(1.0) / s32)
    # s_learn (model/nerf.py:512-522): s_val is a trained parameter; the march backward accumulates d loss / d inv_s for it
    run.s_param = model.s_val if (getattr(model, 's_learn', False) and is_train and model.s_val.requires_grad) else None
    if run.s_param is not None and run.cache.get('sync_free') is not None:
        raise RuntimeError("the sync-free / captured step reads 1/s from a device-resident SCHEDULE; a learnable s_val (s_learn) is "
                           "served by the eager fused path only")
    run.max_steps = int(math.ceil(run.geom.diag / run.stepdist)) + 2
    run.workspace = _workspace(model, N, run.max_steps, rays_o.device)
    run.sync_free = run.cache.get('sync_free')
    run.count_ptr = None
    run.render_grad = bool(render_kwargs.get('render_grad', False))
    run.render_depth = bool(render_kwargs.get('render_depth', default_depth))
    return run, s_val


def roofline_report(pmc=None, flop_scale: float = 1.0, stamped=None):
    """Achieved fp32 FLOP/s of the dominant kernels -- the MLP matrix-core kernels: k_mlp_rc (register-resident forward chain
    and backward data-gradient chain, one launch each), k_mlp_wgrad (all weight / bias gradients, one launch), k_gemm (the two
    first-layer data gradients) -- from the HIP events recorded around every uninterrupted run of them, against the gfx950
    fp32 matrix-core peak (MI355X_MICROARCH.md: 157.3 TFLOP/s, v_mfma_f32_32x32x2_f32 at 256 FLOP/clk/CU, 2.4 GHz).
    `pmc`: bench.pmc_traffic_live()'s per-kernel HBM bytes (or None / {'error': ...}): `traffic` is then the mean over the
    launches of one step, measured in this run; without it `traffic` is null (never a number from another run)."""
    if stamped is not None:
        # fused_ops.stamps_read(): the launches of a captured step timed by their own workgroups (wall-clock readings), one
        # duration per replay of the timed region
        ev = [(None, None, None, label.replace(" (+ k_rc_pack)", "").replace(" (+ k_rc2_pack)", ""), len(durs), fl * len(durs), sum(durs) * 1e3)
              for label, fl, durs in stamped if durs]
    else:
        ev = [e + (None,) for e in PROFILE["gemm_events"] + fo.TIMING["events"]]
    if not ev:
        return None
    per = {}
    tot_ms, tot_fl, tot_n = 0.0, 0.0, 0
    for e0, e1, e1s, label, n, fl, ms in ev:
        fl = fl * flop_scale
        if ms is None:
            ms = e0.elapsed_time(e1)
            if e1s is not None:
                ms = max(ms, e0.elapsed_time(e1s))
        d = per.setdefault(label, [0, 0.0, 0.0])
        d[0] += n
        d[1] += ms
        d[2] += fl
        tot_ms += ms
        tot_fl += fl
        tot_n += n
    achieved = tot_fl / (tot_ms * 1e-3) / 1e12
    peak = 157.3
    traffic, detail = None, None
    if pmc and pmc.get("kernels"):
        K = pmc["kernels"]
        # launches of one step, from what this run timed: label prefix -> launches
        per_step = {}
        for label, (n, _ms, _fl) in per.items():
            key = next((k for k in K if label.startswith(k) or k in label), None)
            if key is not None:
                per_step[key] = per_step.get(key, 0) + n
        if per_step:
            traffic = round(sum(K[k]["bytes_per_launch"] * n for k, n in per_step.items()) / sum(per_step.values()))
        detail = {"by_kernel": K, "method": pmc.get("method")}
    elif pmc and pmc.get("error"):
        detail = {"error": pmc["error"]}
    out = {"bound": "mfma",
           "kernel": "MLP matrix-core kernels, fp32 v_mfma_f32_32x32x2_f32: k_mlp_rc2 (forward chain / backward data-gradient "
                     "chain incl. the two narrow first-layer products, one launch each: the waves split the features, a CU is "
                     "dealt whole 32-sample tiles; FGS_MLP_FORM=1 or other widths: k_mlp_rc + k_gemm), k_mlp_wgrad (all weight "
                     "and bias gradients, one launch; its duration here includes k_wgrad_reduce, the launch that adds the "
                     "per-workgroup partial blocks in a fixed order: FGS_WGRAD_STORE=1)",
           "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
           "traffic": traffic, "traffic_unit": "HBM bytes per launch, mean over the MLP launches of a step",
           "traffic_detail": detail,
           "launches": tot_n, "avg_launch_us": round(tot_ms * 1e3 / tot_n, 2),
           "algorithmic_gflop_per_launch": round(tot_fl / tot_n / 1e9, 3),
           "timing": "HIP events on the launch stream around each uninterrupted MLP chain in the timed region",
           "chains": {k: {"launches": v[0], "avg_us": round(v[1] * 1e3 / v[0], 2),
                            "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2)} for k, v in per.items()}}
    return out


__all__ = [n for n in dir() if not n.startswith('__')]
