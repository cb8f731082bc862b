"""Fused MI355X path for ``nerf.forward_fine`` / ``nerf.forward_coarse`` (model/nerf.py:776-941, 943-1075): one autograd
node per forward, ~20 HIP launches each way.

Forward:   march (1 wave / ray: sampling + SDF + 6-tap gradient + NeuS alpha + exact early-terminating scan; coarse: two
           trilinear lookups in the dense smoothed / gradient volumes and both Alphas2Weights passes)
           -> scan of per-ray survivor counts -> ONE device->host read (M_s, needed to size the result tensors the
           reference API returns) -> survivor compaction -> feature kernels (k0 trilerp, 24 SDF taps, encodings,
           reflection) writing straight into the MLP operand buffers -> the whole MLP chain in one persistent fp32-MFMA
           launch (k_mlp_fwd; per-layer k_gemm launches for widths other than 256) -> 3-wide head + sigmoid -> per-ray
           compositing.  The backward pass's big zero fills are issued here, at the end.
Backward:  compositing -> head -> per layer ONE launch with the data-gradient tiles and the split-K weight-gradient
           workgroups (k_linear_bwd; bias gradients in the epilogues) -> feature scatter (k0.grad) -> march backward
           (alpha2weight + NeuS alpha) -> all sdf.grad contributions combined per survivor in LDS bricks.
           With a dist.GradAverager attached, the MLP gradients and k0.grad are handed to the exchange from in here.

The reference touches the host ~10 times per forward (`.item()`, seven boolean-mask compactions, `unique`); this
path does it once.  Configurations outside ``supports`` / ``supports_coarse`` run the operator-at-a-time kernels.
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


def supports(model) -> bool:
    """Configurations the fused kernels cover (everything the shipped fine-stage configs use)."""
    from .nerf import mlp_layers
    if model.stage != 'fine' or model.rgbnet is None:
        return False
    if model.smooth_sdf and int(model.smooth_conv.weight.shape[-1]) > 7:      # dense.smooth3d covers kernel sides <= 7
        return False
    if not (model.fast_color_thres > 0) or not model.use_viewdir:
        return False
    if model.k_grad_feat != (1.0,) or len(model.k_sdf_feat) != 0:
        return False
    disp = sorted(set(model.grad_feat + model.k_grad_feat))
    if disp != sorted(set(model.sdf_feat + model.k_sdf_feat)) or len(disp) > 5:
        return False
    rl, fl = mlp_layers(model.rgbnet), mlp_layers(model.refnet)
    x0_cols = (model.k0_dim + (3 + 6 * len(model.posfreq)) + (3 + 6 * len(model.viewfreq)) + int(model.center_sdf)
               + 9 * len(disp) + 3)
    if x0_cols != rl[0].in_features or rl[-1].out_features + 3 + 6 * len(model.reffreq) != fl[0].in_features:
        return False
    rw, fw = rl[0].out_features, fl[0].out_features
    if rw % 4 or fw % 4 or fw > 256 or len(rl) < 2 or len(fl) < 2 or fl[-1].out_features != 3:
        return False
    if any(l.out_features != rw for l in rl) or any(l.out_features != fw for l in fl[:-1]):
        return False
    g = model.sdf.grid
    return g.is_cuda and g.is_contiguous() and model.k0.grid.is_cuda


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


def _layout(model, geom):
    """(layout_i ctypes int[11], displace ctypes float[K], ldx0, ldz, x0_cols) for csrc/features.hip fill_layout."""
    from .nerf import mlp_layers
    disp = sorted(set(model.grad_feat + model.k_grad_feat))   # model/nerf.py:843-851
    K = len(disp)
    rw = mlp_layers(model.rgbnet)[0].out_features
    x0_cols = mlp_layers(model.rgbnet)[0].in_features
    ldx0 = (x0_cols + 3) // 4 * 4
    z_cols = mlp_layers(model.refnet)[0].in_features
    ldz = (z_cols + 3) // 4 * 4
    li = [model.k0_dim, len(model.posfreq), len(model.viewfreq), len(model.reffreq), int(model.use_viewdir),
          int(model.center_sdf), int(model.use_grad_norm), K, ldx0, rw, ldz]
    expect = model.k0_dim + (3 + 6 * li[1]) + (3 + 6 * li[2]) + int(model.center_sdf) + 9 * K + 3
    assert expect == x0_cols and rw + 3 + 6 * li[3] == z_cols, (expect, x0_cols, z_cols)
    return (ctypes.c_int * 11)(*li), (ctypes.c_float * max(K, 1))(*(disp or [0.0])), ldx0, ldz, x0_cols


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



def _backward_chain(run, dY, M, rw, fw, ldz, ldx0, n_rgb, n_ref, rgb_w, ref_w, acts_rgb, acts_ref,
                    gw_rgb, gb_rgb, gw_ref, gb_ref, gW0p, gV0p, cs):
    """FGS_LINEAR_BWD=chain: every data gradient of the two MLPs in ONE persistent launch (fgs_mlp_chain_f32: ReLU masks
    and bias-gradient column sums in the epilogues, the intermediate dY tensors written out for the weight gradients),
    then the positional-encoding columns of dZ and the 7 weight-gradient products as plain GEMMs."""
    S = run.saved
    dev = dY.device
    WT = S['WT']                                   # transposed weights in chain order (built in forward)
    layers, k = [], 0
    spec = []                                      # (dY_in, a_in, dW, n_out, k_in, logical k_in) per chain layer
    cur = dY
    for i in range(n_ref - 2, 0, -1):
        out = torch.empty(M, fw, dtype=F32, device=dev)
        layers.append(dict(W=WT[k], K=fw, mask=acts_ref[i], colsum=gb_ref[i - 1], out=out)); k += 1
        spec.append((cur, acts_ref[i], gw_ref[i], fw, fw, fw))
        cur = out
    dZ = torch.empty(M, ldz, dtype=F32, device=dev)
    layers.append(dict(W=WT[k], K=fw, colsum=cs, out=dZ)); k += 1
    spec.append((cur, acts_ref[0], gV0p, fw, ldz, ref_w[0].shape[1]))
    dY_ref0 = cur
    cur = dZ[:, :rw]
    for i in range(n_rgb - 1, 0, -1):
        out = torch.empty(M, rw, dtype=F32, device=dev)
        layers.append(dict(W=WT[k], K=rw, mask=acts_rgb[i], colsum=gb_rgb[i - 1], out=out)); k += 1
        spec.append((cur, acts_rgb[i], gw_rgb[i], rw, rw, rw))
        cur = out
    dX0 = torch.empty(M, ldx0, dtype=F32, device=dev)
    layers.append(dict(W=WT[k], K=rw, n_rows=ldx0, n_store=ldx0, out=dX0))
    spec.append((cur, acts_rgb[0], gW0p, rw, ldx0, rgb_w[0].shape[1]))
    fo.mlp_chain(M, dY, fw, layers)
    grp = PROFILE.get("open")
    if grp is not None:
        grp[0] += 1
        grp[1] += sum(2.0 * M * n_out * min(lk, 256) for _, _, _, n_out, _, lk in spec)
    # encodings' columns of dZ (refnet layer 0 has ldz > 256 inputs)
    if ldz > 256:
        _gemm(fo.GEMM_NN, dY_ref0, S['V0p'][:, 256:], dZ[:, 256:], M, ldz - 256, fw,
              logical=(M, ref_w[0].shape[1] - 256, fw))
    gb_rgb[-1] = cs[:rw]
    gw_ref[0] = gV0p[:, :ref_w[0].shape[1]]
    for dy_in, a_in, dW, n_out, k_in, lk in spec:
        _gemm(fo.GEMM_TN, dy_in, a_in, dW, n_out, k_in, M, logical=(n_out, lk, M))
    return dZ, dX0


def _rc_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) -> bool:
    """Shapes the register-resident chains cover (a function of the model only: identical on every rank)."""
    return (_MLP_IMPL == "rc" and rw == fw and rw % 32 == 0 and rw <= 256 and ldx0 <= 256 and 0 < ldz - rw <= 64 and
            n_rgb + n_ref - 1 <= 8)


def _backward_rc(run, dY, M, rw, fw, ldz, ldx0, n_rgb, n_ref, rgb_w, ref_w, acts_rgb, acts_ref,
                 gw_rgb, gb_rgb, gw_ref, gb_ref, cs, gV0p=None, rgb_b=None):
    """FGS_MLP=rc: every 256-wide data gradient of the two MLPs in ONE register-resident launch (fgs_mlp_rc_chain on the
    transposed weight images, ReLU masks from the 16-byte-per-lane sign bits the forward chain saved), the two narrow
    products (the reflection-encoding columns of dZ, dX0) as plain NN GEMMs on the dY tensors the chain wrote out.  Returns
    (dZ, dX0, wgrad): `wgrad(fork)` issues every weight and bias gradient in ONE fgs_mlp_wgrad launch, written straight into
    the views of the flat gradient buffer -- the caller decides where in the backward pass (see _wgrad).
    With S['Wc_full'] (FGS_MLP_COLLAPSE): rgbnet's last layer and refnet's first are one layer here too (see _MLP_COLLAPSE);
    `gV0p` (a zero-filled [fw, ldz] slot of the flat buffer the rc path does not otherwise use) receives the collapsed weight's
    gradient, from which three small products behind the weight-gradient launch make dW3, dV0a and db3."""
    S = run.saved
    dev = dY.device
    bits = S['relu_bits']
    Wc_full = S.get('Wc_full')
    collapse = Wc_full is not None
    layers = []
    dY_ref = [None] * (n_ref - 1)            # dY_ref[i]: gradient w.r.t. the pre-activation output of refnet layer i
    dY_ref[n_ref - 2] = dY
    for i in range(n_ref - 2, 0, -1):        # g . W_i, masked by the ReLU of layer i - 1
        out = torch.empty(M, fw, dtype=F32, device=dev)
        layers.append(dict(W=ref_w[i], mask_bits=bits[n_rgb + i - 1], out=out, n_store=fw))
        dY_ref[i - 1] = out
    dZ = torch.empty(M, ldz, dtype=F32, device=dev)
    dY_rgb = [None] * n_rgb                  # dY_rgb[i]: gradient w.r.t. the output of rgbnet layer i
    if collapse:
        # dY_ref[0] . (V0a W3), masked by the ReLU of rgbnet layer n_rgb - 2: straight to that layer's output gradient
        out = torch.empty(M, rw, dtype=F32, device=dev)
        layers.append(dict(W=Wc_full[:, :rw], mask_bits=bits[n_rgb - 2], out=out, n_store=rw))
        dY_rgb[n_rgb - 2] = out
        first_rgb = n_rgb - 2
    else:
        layers.append(dict(W=ref_w[0][:, :rw], out=dZ, n_store=rw))      # no activation under refnet layer 0: no mask
        dY_rgb[n_rgb - 1] = dZ[:, :rw]
        first_rgb = n_rgb - 1
    for i in range(first_rgb, 0, -1):
        out = torch.empty(M, rw, dtype=F32, device=dev)
        layers.append(dict(W=rgb_w[i], mask_bits=bits[i - 1], out=out, n_store=rw))
        dY_rgb[i - 1] = out
    flop_chain = 2.0 * M * (fw * fw * (n_ref - 2) + fw * rw + rw * rw * (n_rgb - 1))
    if collapse:
        flop_chain -= 2.0 * M * rw * rw
    fo.rc_chain(True, M, dY, fw, layers, flop=flop_chain, rows_dev=_rows(run))
    # narrow products: the reflection-encoding columns of dZ (dY_ref[0] . V0[:, rw:]) and dX0 (dY_rgb[0] . W0).  (As one-layer
    # register-resident chains of 4 row tiles they measured 63 us each against 47 for the tiled GEMM: with 64 MFMAs per chunk
    # the chain's per-chunk barrier / DMA and its uncoalesced input load dominate.)
    z_cols, x_cols = ref_w[0].shape[1], rgb_w[0].shape[1]
    _gemm(fo.GEMM_NN, dY_ref[0], S['V0p'][:, rw:], dZ[:, rw:], M, ldz - rw, fw, logical=(M, z_cols - rw, fw), rows_dev=_rows(run))
    if S.get('W0c') is not None:
        # dX0 in compact form (fgs_dyn_t.dx0_compact): only the columns somebody differentiates through
        W0c = S['W0c']
        dX0 = torch.empty(M, W0c.shape[1], dtype=F32, device=dev)
        _gemm(fo.GEMM_NN, dY_rgb[0], W0c, dX0, M, W0c.shape[1], rw, logical=(M, run.dx0_cols[2], rw), rows_dev=_rows(run))
        run.dx0_compact = True
    else:
        dX0 = torch.empty(M, ldx0, dtype=F32, device=dev)
        _gemm(fo.GEMM_NN, dY_rgb[0], S['W0p'], dX0, M, ldx0, rw, logical=(M, x_cols, rw), rows_dev=_rows(run))
        run.dx0_compact = False
    # all weight / bias gradients (the bias gradient of the top refnet layer came out of the head kernel)
    items = []
    post = None
    for i in range(n_ref - 1):
        if collapse and i == 0:
            # dWc = dY_ref0^T h (h = the input of rgbnet's last layer) into the spare zero-filled slot, dbc straight into dc0's
            # slot (dc0 = dbc), and the reflection-encoding columns of dV0 where they belong
            items.append((dY_ref[0], acts_rgb[n_rgb - 1], gV0p[:, :rw], gb_ref[0], fw, rw))
            items.append((dY_ref[0], S['Z'][:, rw:], gw_ref[0][:, rw:], None, fw, z_cols - rw))
            continue
        items.append((dY_ref[i], acts_ref[i], gw_ref[i], None if i == n_ref - 2 else gb_ref[i], fw, ref_w[i].shape[1]))
    for i in range(n_rgb - 1 if collapse else n_rgb):
        items.append((dY_rgb[i], acts_rgb[i], gw_rgb[i], gb_rgb[i], rw, rgb_w[i].shape[1]))
    flop = 2.0 * M * (fw * sum(w.shape[1] for w in ref_w[:-1]) + rw * sum(w.shape[1] for w in rgb_w))
    if collapse:
        flop -= 2.0 * M * rw * rw
        V0a, W3, b3 = S['V0p'][:, :rw], rgb_w[n_rgb - 1], rgb_b[n_rgb - 1]
        dWc, dbc = gV0p[:, :rw], gb_ref[0]

        def post():
            # (on the stream of the weight-gradient launch, right behind it: three 256^3 products and three small vector ops)
            tmp = torch.empty(fw, rw, dtype=F32, device=dev)
            fo.gemm(fo.GEMM_TN, V0a, dWc, gw_rgb[n_rgb - 1], rw, rw, fw)                     # dW3 = V0a^T dWc
            fo.gemm(fo.GEMM_NT, dWc, W3.detach(), tmp, fw, rw, rw)                            # dWc W3^T
            gw_ref[0][:, :rw].copy_(torch.addcmul(tmp, dbc[:, None], b3.detach()[None, :]))   # dV0a = dWc W3^T + dbc b3^T
            gb_rgb[n_rgb - 1].copy_((V0a * dbc[:, None]).sum(0))                              # db3 = V0a^T dbc
    return dZ, dX0, lambda fork: _wgrad(dev, M, items, flop, fork, post, rows_dev=_rows(run))


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
    keep.append(items)                  # (allocated on the main stream: alive until the join)
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
_BRICK_ADAM = os.environ.get("FGS_BRICK_ADAM", "1") != "0"
_COARSE_VOL4 = os.environ.get("FGS_COARSE_VOL4", "1") != "0"


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
    if not create or not _BRICK_ADAM:
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


class _FusedFine(torch.autograd.Function):
    """inputs: sdf grid, k0 grid, then (weight, bias) of every rgbnet and refnet Linear; `run` carries the rest."""

    @staticmethod
    def forward(ctx, run, sdf_grid, k0_grid, *mlp):
        if run.s_param is not None:
            mlp = mlp[:-1]              # (the learnable s_val rides along as the last input: only its gradient matters here)
        # outputs the loss does not use arrive as None in backward (the kernels take NULL) instead of as zero tensors that
        # autograd would fill -- six launches of ~5 us at the head of the backward pass, one of them an int64 fill for ray_id
        ctx.set_materialize_grads(False)
        dev = sdf_grid.device
        g, N, st = run.geom, run.n_rays, stream()
        ms = run.max_steps
        rec = N * ms
        ws = run.workspace
        _own_workspace(run, any(ctx.needs_input_grad))
        # 1. march (alphainv_last is an output of this call: a fresh tensor per step, the other records live in `ws`)
        alphainv_last = torch.empty(N, dtype=F32, device=dev)
        call("fgs_march_fine_fwd", ptr(run.rays_o), ptr(run.rays_d), ptr(run.viewdirs), N, g.lo_c, g.hi_c, g.X, g.Y, g.Z,
             g.voxel_size, run.near, 1e9, run.stepdist, ptr(sdf_grid), run.dist, run.inv_s, run.thres,
             ptr(run.mask_grid), *(g.mask[:2] if g.mask else (None, None)), *(g.mask[2] if g.mask else (0, 0, 0)),
             g.mask[3] if g.mask else 0.0, ms, ptr(ws['a_step']), ptr(ws['a_alpha']), ptr(ws['a_T']), ptr(ws['a_weight']),
             ptr(ws['a_sdf']), ptr(ws['a_grad']), ptr(ws['a_surv']), ptr(ws['surv_slot']), ptr(ws['n_alive']),
             ptr(ws['n_surv']), ptr(ws['n_inbbox']), ptr(alphainv_last), dyn(inv_s=_inv_s(run)), st)
        sf = run.sync_free
        if sf:      # sync-free: offsets cut at the capacity and the overflow flags set by the scan launch itself
            call("fgs_exclusive_scan_guard_i64", ptr(ws['n_surv']), N, ptr(ws['surv_off']), sf['capacity'], ptr(sf['flags']),
                 ptr(sf['total']), st)
        else:
            call("fgs_exclusive_scan_i64", ptr(ws['n_surv']), N, ptr(ws['surv_off']), st)
        # everything that does not need the survivor count is issued BEFORE the host read, off the post-sync path:
        # first-layer weights are copied into K-padded operands (their row length is not a multiple of 4)
        n_rgb, n_ref = run.n_rgb, run.n_ref
        rgb_w = [mlp[2 * i] for i in range(n_rgb)]
        rgb_b = [mlp[2 * i + 1] for i in range(n_rgb)]
        ref_w = [mlp[2 * (n_rgb + i)] for i in range(n_ref)]
        ref_b = [mlp[2 * (n_rgb + i) + 1] for i in range(n_ref)]
        rw, fw = rgb_w[0].shape[0], ref_w[0].shape[0]
        ldx0, ldz = run.ldx0, run.ldz
        sf = run.sync_free
        token = None if sf else _count_begin(run, ws['surv_off'], N)
        # K-padded first-layer weights of both MLPs, one launch (F.pad: a fill + a copy launch per matrix)
        W0c = None
        if _DX0_COMPACT and _rc_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) and any(ctx.needs_input_grad):
            # ... and, in the same launch, the first rgbnet layer's weights WITHOUT the columns of the xyz / view-direction
            # encodings: the backward pass needs d loss / d X0 only for the k0, sdf, tap and gradient columns (12 + 40 of 106)
            k0d, gap, cw = run.dx0_cols
            W0 = rgb_w[0].detach()
            W0c = torch.empty(rw, (cw + 3) // 4 * 4, dtype=F32, device=dev)
            W0p, V0p, _, _ = fo.pad_cols_multi([W0, ref_w[0].detach(), W0[:, :k0d], W0[:, k0d + gap:]],
                                               [ldx0, ldz, k0d, cw - k0d], outs=[None, None, W0c[:, :k0d], W0c[:, k0d:cw]])
        else:
            W0p, V0p = fo.pad_cols_multi([rgb_w[0].detach(), ref_w[0].detach()], [ldx0, ldz])
        pre_k0 = _prefill_grid_grad(run, k0_grid) if (_PRE_FILL_AT_READ and any(ctx.needs_input_grad)) else None
        kC, kX, kY, kZ, ksC, ksX, ksY, ksZ = grid_strides(k0_grid)
        if sf:
            # sync-free: the count stays on the device (last entry of the survivor offsets); M is the CAPACITY from here on
            M = sf['capacity']
            run.count_ptr = ws['surv_off'].data_ptr() + 8 * N     # (handed to every per-survivor launch: fgs_dyn_t.row_count)
        else:
            M = _count_end(token)                  # the one host read of the step
        run.M = M
        # 2. survivors
        ray_id = torch.empty(M, dtype=I64, device=dev)
        step_id = torch.empty(M, dtype=I64, device=dev)
        rec_idx = torch.empty(M, dtype=I32, device=dev)
        weights = torch.empty(M, dtype=F32, device=dev)
        alpha = torch.empty(M, dtype=F32, device=dev)
        sdf = torch.empty(M, dtype=F32, device=dev)
        gradient = torch.empty(M, 3, dtype=F32, device=dev)
        pts = torch.empty(M, 3, dtype=F32, device=dev)
        call("fgs_surv_compact", N, M, ptr(ws['surv_off']), ms, ptr(ws['surv_slot']), ptr(ws['a_step']), ptr(ws['a_alpha']),
             ptr(ws['a_weight']), ptr(ws['a_sdf']), ptr(ws['a_grad']), ptr(run.rays_o), ptr(run.rays_d), g.lo_c, g.hi_c,
             g.X, g.Y, g.Z, run.near, 1e9, run.stepdist, ptr(ray_id), ptr(step_id), ptr(rec_idx), ptr(weights), ptr(alpha),
             ptr(sdf), ptr(gradient), ptr(pts), dyn(row_count=_rows(run)), st)
        # 3. features
        ldx0, ldz = run.ldx0, run.ldz
        X0 = torch.empty(M, ldx0, dtype=F32, device=dev)
        Z = torch.empty(M, ldz, dtype=F32, device=dev)
        normal = torch.empty(M, 3, dtype=F32, device=dev)
        kC, kX, kY, kZ, ksC, ksX, ksY, ksZ = grid_strides(k0_grid)
        call("fgs_feat_fine_fwd", M, ptr(ray_id), ptr(pts), ptr(sdf), ptr(gradient), ptr(run.viewdirs), g.lo_c, g.hi_c,
             g.X, g.Y, g.Z, g.voxel_size, run.layout_i, run.displace, ptr(sdf_grid), ptr(k0_grid), ksC, ksX, ksY, ksZ,
             ptr(X0), ptr(Z), ptr(normal), dyn(row_count=_rows(run)), st)
        # 4. MLPs
        use_rc = _rc_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) and M > 0
        one_launch = (not use_rc and _MLP_FWD_ONE_LAUNCH and rw == 256 and fw == 256 and ldx0 <= 128 and 0 < ldz - rw <= 64 and
                      n_rgb + n_ref - 1 <= 8)
        grp = _gemm_group("forward chain (" + ("k_mlp_rc: register-resident, all layers in one launch" if use_rc else
                                               "k_mlp_fwd: all layers in one launch" if one_launch
                                               else "NT: k_gemm<true,true,0>") + ")").__enter__()
        acts_rgb = [X0] + [torch.empty(M, rw, dtype=F32, device=dev) for _ in range(n_rgb - 1)]   # input of each rgbnet layer
        acts_ref = [Z] + [torch.empty(M, fw, dtype=F32, device=dev) for _ in range(n_ref - 1)]    # input of each refnet layer
        relu_bits = None
        if use_rc:
            # ReLU sign bits of every hidden layer, 16 bytes per lane of each 32-sample group, one buffer for all layers
            per = fo.rc_mask_bits(M, dev).numel()
            relu_bits = torch.empty(n_rgb + n_ref - 1, per, dtype=torch.int32, device=dev)
            layers = []
            collapse = _MLP_COLLAPSE and n_rgb >= 2 and n_ref >= 2
            Wc_full = bias_c = None
            if collapse:
                # Wc_full = [V0a W3 | V0b] (K-padded like V0p), bias_c = V0a b3 + c0: two small launches per step
                Wc_full = V0p.clone()
                fo.gemm(fo.GEMM_NN, V0p[:, :rw], rgb_w[-1].detach(), Wc_full[:, :rw], fw, rw, rw)
                bias_c = (V0p[:, :rw] * rgb_b[-1].detach()).sum(1) + ref_b[0].detach()
            for i in range(n_rgb - 1 if collapse else n_rgb):   # the last rgbnet layer writes Z[:, :rw] (no ReLU); Z[:, rw:] holds the reflect PE
                last = i == n_rgb - 1
                layers.append(dict(W=rgb_w[i].detach(), bias=rgb_b[i].detach(), relu=not last,
                                   mask_bits=None if last else relu_bits[i], out=Z if last else acts_rgb[i + 1], n_store=rw))
            for i in range(n_ref - 1):
                L = dict(W=ref_w[i].detach(), bias=ref_b[i].detach(), relu=True, mask_bits=relu_bits[n_rgb + i],
                         out=acts_ref[i + 1], n_store=fw)
                if i == 0:
                    L.update(ext=Z[:, rw:], ext_cols=ldz - rw)
                    if collapse:     # the carried input is rgbnet's last HIDDEN activation, the weight the pre-multiplied one
                        L.update(W=Wc_full[:, :ref_w[0].shape[1]], bias=bias_c)
                layers.append(L)
            flop_fwd = 2.0 * M * (rw * sum(w.shape[1] for w in rgb_w) + fw * sum(w.shape[1] for w in ref_w[:-1]))
            if collapse:
                flop_fwd -= 2.0 * M * rw * rgb_w[-1].shape[1]
            fo.rc_chain(False, M, X0, ldx0, layers, flop=flop_fwd, rows_dev=_rows(run))
        elif one_launch:
            layers = []
            for i in range(n_rgb):       # the last rgbnet layer writes Z[:, :rw] (no ReLU); Z[:, rw:] holds the reflect PE
                layers.append((W0p if i == 0 else rgb_w[i].detach(), ldx0 if i == 0 else rw, rgb_b[i].detach(),
                               i < n_rgb - 1, Z if i == n_rgb - 1 else acts_rgb[i + 1]))
            for i in range(n_ref - 1):
                layers.append((V0p if i == 0 else ref_w[i].detach(), ldz if i == 0 else fw, ref_b[i].detach(), True,
                               acts_ref[i + 1]))
            fo.mlp_fwd(M, X0, ldx0, Z[:, rw:], ldz - rw, layers)
            if PROFILE.get("open") is not None:
                PROFILE["open"][0] += 1
                PROFILE["open"][1] += 2.0 * M * (rw * sum(w.shape[1] for w in rgb_w) + fw * sum(w.shape[1] for w in ref_w[:-1]))
        else:
            a = X0
            for i in range(n_rgb):
                last = i == n_rgb - 1
                out = Z if last else acts_rgb[i + 1]
                B = W0p if i == 0 else rgb_w[i].detach()
                _gemm(fo.GEMM_NT, a, B, out, M, rw, a.shape[1] if i else ldx0, bias=rgb_b[i].detach(), relu=not last,
                      logical=(M, rw, rgb_w[i].shape[1]))
                a = out
            a = Z
            for i in range(n_ref - 1):
                out = acts_ref[i + 1]
                B = V0p if i == 0 else ref_w[i].detach()
                _gemm(fo.GEMM_NT, a, B, out, M, fw, ldz if i == 0 else fw, bias=ref_b[i].detach(), relu=True,
                      logical=(M, fw, ref_w[i].shape[1]))
                a = out
        a = acts_ref[n_ref - 1]
        grp.__exit__()
        rgb = torch.empty(M, 3, dtype=F32, device=dev)
        call("fgs_head_fwd", ptr(a), a.stride(0), fw, M, ptr(ref_w[-1].detach()), ptr(ref_b[-1].detach()), ptr(rgb),
             dyn(row_count=_rows(run)), st)
        # 5. compositing
        rgb_marched = torch.empty(N, 3, dtype=F32, device=dev)
        sigmoid_rgb = torch.empty(N, 3, dtype=F32, device=dev)
        pre_rgb = torch.empty(N, 3, dtype=F32, device=dev)
        pre_sig = torch.empty(N, 3, dtype=F32, device=dev)
        normal_marched = torch.empty(N, 3, dtype=F32, device=dev) if run.render_grad else None
        depth = torch.empty(N, dtype=F32, device=dev) if run.render_depth else None
        call("fgs_composite_fwd", N, ptr(ws['surv_off']), ptr(weights), ptr(rgb), ptr(normal), ptr(step_id), run.bg, run.dist,
             ptr(rgb_marched), ptr(sigmoid_rgb), ptr(pre_rgb), ptr(pre_sig), ptr(normal_marched), ptr(depth), st)
        # The big zero fills of the backward pass are issued HERE: when loss.backward() starts, the autograd engine needs
        # ~90 us of host time before its first launch and the GPU would sit idle; now it spends that gap on the fills.
        run.pre = None
        if any(ctx.needs_input_grad) and M > 0:        # all False under torch.no_grad() (rendering)
            # sdf.grad and the flat buffer of the MLP gradients (both accumulated into by the backward pass) share ONE zero
            # fill: [flat | sdf.grad], each 16-byte aligned
            _, n_flat = _fine_grad_layout(run, rgb_w, ref_w, rw, fw, ldx0, ldz)
            arena = torch.zeros(n_flat + sdf_grid.numel(), dtype=F32, device=dev)
            run.pre = (arena[n_flat:].view(sdf_grid.shape), pre_k0, arena[:n_flat])
        WT = None
        if run.pre is not None and _LINEAR_BWD_MODE == "chain" and rw == 256 and fw == 256 and n_ref - 1 + n_rgb <= 8:
            # transposed weights in the order the backward chain walks the layers (dX = dY . W as a forward-shaped product)
            WT = fo.transpose_multi([ref_w[i].detach() for i in range(n_ref - 2, 0, -1)] + [V0p[:, :rw]] +
                                    [rgb_w[i].detach() for i in range(n_rgb - 1, 0, -1)] + [W0p])

        # Tensors this function RETURNS must not be reachable from ctx through plain attributes: output -> grad_fn -> ctx
        # -> run -> output is a cycle through C++ that Python's collector cannot see (0.3 GB leaked per step).  Keep
        # detached aliases (same storage, no grad_fn) instead.
        run.saved = _detached(dict(ray_id=ray_id, pts=pts, sdf=sdf, gradient=gradient, weights=weights, rgb=rgb, X0=X0, Z=Z,
                                   acts_rgb=acts_rgb, acts_ref=acts_ref, W0p=W0p, V0p=V0p, W0c=W0c, WT=WT, relu_bits=relu_bits,
                                   Wc_full=(Wc_full if use_rc else None),
                                   pre_rgb=pre_rgb, pre_sig=pre_sig,
                                   alphainv_last=alphainv_last, k0_strides=(ksC, ksX, ksY, ksZ)))
        run.extras = dict(step_id=step_id, rec_idx=rec_idx, normal_marched=normal_marched, depth=depth,
                          n_inbbox=ws['n_inbbox'])
        ctx.run = run
        ctx.save_for_backward(sdf_grid, k0_grid, *mlp)
        ctx.mark_non_differentiable(ray_id, alpha, gradient)
        return rgb_marched, sigmoid_rgb, alphainv_last, weights, rgb, normal, ray_id, alpha, gradient

    @staticmethod
    def _backward_empty(run, sdf_grid, k0_grid, mlp, rgb_w, ref_w, rw, fw, ldx0, ldz):
        """No sample survived on THIS rank (every ray missed the volume): the local gradients are exactly zero, but the
        other ranks still exchange theirs from inside their backward passes -- issue the same hooks in the same order
        on the same buffers shapes, or the collectives of the early communicator would not match up (deadlock)."""
        dev = sdf_grid.device
        n_rgb, n_ref = run.n_rgb, run.n_ref
        items, total = _fine_grad_layout(run, rgb_w, ref_w, rw, fw, ldx0, ldz)
        flat = torch.zeros(total, dtype=F32, device=dev)
        views = [flat[off:off + n].view(sh) for sh, n, off in items]
        grad_sdf = torch.zeros_like(sdf_grid)
        grad_k0 = torch.empty_strided(k0_grid.shape, k0_grid.stride(), dtype=F32, device=dev).zero_()
        run.pre = None
        hook, opt_hook = _early_hooks(run)
        if hook is not None:
            hook('k0', [k0_grid], grad_k0)
            hook('mlp', mlp, flat)
            hook('join', None)
        elif opt_hook is not None:
            opt_hook(k0_grid, grad_k0)
        gw_rgb, gw_ref = views[:n_rgb], views[n_rgb:n_rgb + n_ref]
        gb_rgb = views[n_rgb + n_ref:2 * n_rgb + n_ref]
        gb_ref = views[2 * n_rgb + n_ref:2 * n_rgb + 2 * n_ref]
        if not _rc_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref):
            # the GEMM path keeps the first-layer weight gradients in K-padded slots and the last rgbnet bias gradient in the
            # column-sum slot: that is where the other ranks' contributions arrive
            gW0p, gV0p, cs = views[-3], views[-2], views[-1]
            gw_rgb[0], gw_ref[0], gb_rgb[-1] = gW0p[:, :rgb_w[0].shape[1]], gV0p[:, :ref_w[0].shape[1]], cs[:rw]
        grads = [None, grad_sdf, grad_k0]
        for i in range(n_rgb):
            grads += [gw_rgb[i].contiguous(), gb_rgb[i].contiguous()]
        for i in range(n_ref):
            grads += [gw_ref[i].contiguous(), gb_ref[i].contiguous()]
        if run.s_param is not None:
            grads.append(torch.zeros_like(run.s_param))
        return tuple(grads)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        return _FusedFine._backward_impl(ctx, *grads)

    @staticmethod
    def _backward_impl(ctx, g_rgb_marched, g_sigmoid_rgb, g_last, g_weights, g_raw_rgb, g_normal, *_unused):
        run = ctx.run
        if run.done:
            raise RuntimeError("fused forward_fine: backward called twice on the same forward (its march records are released "
                               "after the first backward; retain_graph is not supported by the fused path)")
        run.done = True          # releases the record set for the next forward (the kernels below are already ordered
        sdf_grid, k0_grid, *mlp = ctx.saved_tensors            # on the stream in front of anything that forward enqueues)
        S, g, N, M, st = run.saved, run.geom, run.n_rays, run.M, stream()
        ws = run.workspace
        dev = sdf_grid.device
        n_rgb, n_ref = run.n_rgb, run.n_ref
        rgb_w = [mlp[2 * i] for i in range(n_rgb)]
        ref_w = [mlp[2 * (n_rgb + i)] for i in range(n_ref)]
        rw, fw = rgb_w[0].shape[0], ref_w[0].shape[0]
        ldx0, ldz = run.ldx0, run.ldz

        def c(t):
            return None if t is None else t.contiguous()
        g_rgb_marched, g_sigmoid_rgb, g_last, g_weights, g_raw_rgb, g_normal = map(
            c, (g_rgb_marched, g_sigmoid_rgb, g_last, g_weights, g_raw_rgb, g_normal))

        if M == 0:
            return _FusedFine._backward_empty(run, sdf_grid, k0_grid, mlp, rgb_w, ref_w, rw, fw, ldx0, ldz)
        _seam(run, 'inputs', g_rgb_marched=g_rgb_marched, g_sigmoid_rgb=g_sigmoid_rgb, g_last=g_last, g_weights=g_weights,
              g_raw_rgb=g_raw_rgb, g_normal=g_normal)
        # 1. compositing
        d_out = torch.empty(M, 3, dtype=F32, device=dev)
        d_w = torch.empty(M, dtype=F32, device=dev)
        call("fgs_composite_bwd", M, ptr(S['ray_id']), ptr(S['weights']), ptr(S['rgb']), ptr(S['pre_rgb']), ptr(S['pre_sig']),
             ptr(g_rgb_marched), ptr(g_sigmoid_rgb), ptr(g_raw_rgb), ptr(g_weights), run.bg, ptr(d_out), ptr(d_w),
             dyn(row_count=_rows(run)), st)
        _seam(run, 'composite', d_out=d_out, d_w=d_w)

        # gradient buffers of the MLP parameters (weights via split-K atomics -> zero-initialised): one zero fill for all of
        # them, views of a flat buffer, each 16-byte aligned.  The layout is cached; only the three views the head kernel
        # needs are made before its launch, the rest while it runs (the GPU is idle at the start of a backward pass).
        items, total = _fine_grad_layout(run, rgb_w, ref_w, rw, fw, ldx0, ldz)
        flat = run.pre[2] if run.pre is not None else torch.zeros(total, dtype=F32, device=dev)   # (zero-filled in forward)

        def view(i):
            sh, n, off = items[i]
            return flat[off:off + n].view(sh)
        i_gw_ref, i_gb_rgb, i_gb_ref = n_rgb, n_rgb + n_ref, 2 * n_rgb + n_ref

        # 2. head: d_out -> dY of refnet layer n_ref-2 (masked), dV_last, dc_last, bias grad of layer n_ref-2
        acts_ref, acts_rgb = S['acts_ref'], S['acts_rgb']
        a_last = acts_ref[n_ref - 1]
        dY = torch.empty(M, fw, dtype=F32, device=dev)
        gw_last, gb_last, gb_prev = view(i_gw_ref + n_ref - 1), view(i_gb_ref + n_ref - 1), view(i_gb_ref + n_ref - 2)
        call("fgs_head_bwd", ptr(a_last), a_last.stride(0), fw, M, ptr(ref_w[-1]), ptr(d_out), ptr(dY), ptr(gw_last),
             ptr(gb_last), ptr(gb_prev), ptr(_head_scratch(fw, dev)), dyn(row_count=_rows(run)), st)
        _seam(run, 'head', dY=dY)
        views = [view(i) for i in range(len(items))]
        gw_rgb, gw_ref = views[:n_rgb], views[n_rgb:n_rgb + n_ref]
        gb_rgb = views[i_gb_rgb:i_gb_rgb + n_rgb]
        gb_ref = views[i_gb_ref:i_gb_ref + n_ref]
        gW0p, gV0p, cs = views[-3], views[-2], views[-1]
        # 3. refnet layers n_ref-2 .. 0   (dY is the gradient w.r.t. the pre-activation output of layer i)
        grp = _gemm_group("backward chain (" + _LINEAR_BWD_MODE + ")").__enter__()
        wgrad = None
        if S.get('relu_bits') is not None:
            dZ, dX0, wgrad = _backward_rc(run, dY, M, rw, fw, ldz, ldx0, n_rgb, n_ref, rgb_w, ref_w, acts_rgb, acts_ref,
                                          gw_rgb, gb_rgb, gw_ref, gb_ref, cs, gV0p=gV0p,
                                          rgb_b=[mlp[2 * i + 1] for i in range(n_rgb)])
        elif S.get('WT') is not None and ldx0 <= 256:
            dZ, dX0 = _backward_chain(run, dY, M, rw, fw, ldz, ldx0, n_rgb, n_ref, rgb_w, ref_w, acts_rgb, acts_ref,
                                      gw_rgb, gb_rgb, gw_ref, gb_ref, gW0p, gV0p, cs)
        else:
            for i in range(n_ref - 2, -1, -1):
                a_in = acts_ref[i]                      # input of layer i: Z for i == 0
                if i == 0:
                    dZ = torch.empty(M, ldz, dtype=F32, device=dev)
                    # no activation between the rgbnet output / encodings and refnet layer 0: no mask;
                    # column sums of dZ[:, :rw] are the bias gradient of the last rgbnet layer
                    _linear_bwd(dY, S['V0p'], a_in, dZ, gV0p, M, fw, ldz, colsum=cs, logical_k_in=ref_w[0].shape[1])
                    gb_rgb[-1] = cs[:rw]
                else:
                    d_in = torch.empty(M, fw, dtype=F32, device=dev)
                    _linear_bwd(dY, ref_w[i], a_in, d_in, gw_ref[i], M, fw, fw, mask=a_in, colsum=gb_ref[i - 1])
                    dY = d_in
            gw_ref[0] = gV0p[:, :ref_w[0].shape[1]]
            # 4. rgbnet layers n_rgb-1 .. 0 ; dY of the last layer is dZ[:, :rw] (a strided view, ld = ldz)
            dY = dZ[:, :rw]
            for i in range(n_rgb - 1, -1, -1):
                a_in = acts_rgb[i]                      # X0 for i == 0
                if i == 0:
                    dX0 = torch.empty(M, ldx0, dtype=F32, device=dev)
                    _linear_bwd(dY, S['W0p'], a_in, dX0, gW0p, M, rw, ldx0, logical_k_in=rgb_w[0].shape[1])
                else:
                    d_in = torch.empty(M, rw, dtype=F32, device=dev)
                    _linear_bwd(dY, rgb_w[i], a_in, d_in, gw_rgb[i], M, rw, rw, mask=a_in, colsum=gb_rgb[i - 1])
                    dY = d_in
        if S.get('relu_bits') is None:
            gw_rgb[0] = gW0p[:, :rgb_w[0].shape[1]]
        grp.__exit__()
        _flush_tn(dev)
        _seam(run, 'mlp', dX0=dX0, dZ=dZ, compact=bool(getattr(run, 'dx0_compact', False)), saved=S)
        hook, opt_hook = _early_hooks(run)
        # One GPU: the weight-gradient launch is forked off here and everything below runs beside it (_MARCH_FIRST: the two
        # vector-bound kernels of the sdf path first, see there).
        march_first = wgrad is not None and hook is None and _MARCH_FIRST
        forked = False
        if wgrad is not None and not march_first and (hook is None or (_WGRAD_FORK and _WGRAD_FORK_DIST)):
            wgrad(True)                          # on a side stream, beside everything below (with an exchange attached too:
            wgrad = None                         # the MLP gradients' exchange is then issued from that stream, see below)
            forked = hook is not None

        # 5. features -> grids
        if run.pre is not None:
            grad_sdf, pre_k0 = run.pre[:2]        # zero-filled at the end of the forward pass
            run.pre = None
        else:
            grad_sdf, pre_k0 = torch.zeros_like(sdf_grid), None
        grad_k0, k0_state = pre_k0 if pre_k0 is not None else _take_grid_grad(run.cache, k0_grid)
        g_sdf_s = torch.empty(M, dtype=F32, device=dev)
        g_grad_s = torch.empty(M, 3, dtype=F32, device=dev)
        tot_sdf = torch.empty(M, dtype=F32, device=dev)
        tot_grad = torch.empty(M, 3, dtype=F32, device=dev)
        ksC, ksX, ksY, ksZ = S['k0_strides']

        compact = bool(getattr(run, 'dx0_compact', False))

        def feat_bwd(k0_part: bool, enc_part: bool):
            call("fgs_feat_fine_bwd", M, ptr(S['ray_id']), ptr(S['pts']), ptr(S['sdf']), ptr(S['gradient']), ptr(run.viewdirs),
                 g.lo_c, g.hi_c, g.X, g.Y, g.Z, g.voxel_size, run.layout_i, run.displace, ptr(S['X0']), ptr(S['Z']), ptr(dX0),
                 ptr(dZ), ptr(g_normal), ptr(grad_sdf), ptr(grad_k0) if k0_part else None, ksC, ksX, ksY, ksZ,
                 ptr(g_sdf_s) if enc_part else None, ptr(g_grad_s) if enc_part else None,
                 dyn(row_count=_rows(run), compact=compact), st)

        g_inv_s = torch.zeros(1, dtype=F32, device=dev) if run.s_param is not None else None

        def march_bwd():
            call("fgs_march_fine_bwd", ptr(run.rays_o), ptr(run.rays_d), ptr(run.viewdirs), N, g.lo_c, g.hi_c, g.X, g.Y, g.Z,
                 g.voxel_size, run.near, 1e9, run.stepdist, run.dist, run.inv_s, run.max_steps, ptr(ws['a_step']),
                 ptr(ws['a_surv']), ptr(ws['a_alpha']), ptr(ws['a_T']), ptr(ws['a_weight']), ptr(ws['a_sdf']), ptr(ws['a_grad']),
                 ptr(ws['n_alive']), ptr(ws['surv_off']), ptr(S['alphainv_last']), ptr(d_w), ptr(g_last), ptr(g_sdf_s),
                 ptr(g_grad_s), ptr(grad_sdf), ptr(tot_sdf), ptr(tot_grad), ptr(g_inv_s), dyn(inv_s=_inv_s(run)), st)

        if march_first:
            feat_bwd(False, True)
            march_bwd()
            wgrad(True)
            wgrad = None
            feat_bwd(True, False)
        else:
            feat_bwd(True, True)
        _seam(run, 'features', g_sdf_s=g_sdf_s, g_grad_s=g_grad_s, grad_k0=grad_k0)
        _publish_touched(k0_state, k0_grid, grad_k0, S['pts'], M, g, st, exchange=hook is not None, rows_dev=_rows(run))
        if hook is not None:
            # the exchanges, in the order EVERY path of every rank issues them (k0, mlp, join: _backward_empty too): k0's is
            # the long one (tens of MB at 8 ranks) and starts first, under the weight-gradient launch and the sdf scatter
            # kernels; the MLP gradients -- views of `flat`, final after that launch -- follow
            hook('k0', [k0_grid], grad_k0)
            _exchange_mlp(dev, wgrad, forked, hook, mlp, flat)
        elif opt_hook is not None and not _K0_ADAM_LATE:
            opt_hook(k0_grid, grad_k0)           # MaskedAdam.early_update: k0's Adam pass runs beside them too
        # 6. march backward
        if not march_first:
            march_bwd()
        _seam(run, 'march', tot_sdf=tot_sdf, tot_grad=tot_grad)
        # 7. every sdf.grad contribution of the survivors (24 taps + centre + six +/-1 taps), combined on chip
        call("fgs_sdf_scatter_surv", M, ptr(S['pts']), g.lo_c, g.hi_c, g.X, g.Y, g.Z, g.voxel_size, run.layout_i,
             run.displace, ptr(S['X0']), ptr(dX0), ptr(tot_sdf), ptr(tot_grad), ptr(grad_sdf),
             dyn(row_count=_rows(run), compact=compact), st)
        if hook is None and opt_hook is not None and _K0_ADAM_LATE:
            opt_hook(k0_grid, grad_k0)           # ... as the LAST kernel of this branch (see _K0_ADAM_LATE)

        _join_side(dev)
        if hook is not None:
            hook('join', None)
        grads: List[Optional[torch.Tensor]] = [None, grad_sdf, grad_k0]
        for i in range(n_rgb):
            grads += [gw_rgb[i].contiguous(), gb_rgb[i].contiguous()]
        for i in range(n_ref):
            grads += [gw_ref[i].contiguous(), gb_ref[i].contiguous()]
        if run.s_param is not None:     # inv_s = 1 / s_val  =>  d s_val = -d inv_s / s_val^2
            grads.append((-g_inv_s / run.s_param.detach().to(dev).float() ** 2).reshape(run.s_param.shape))
        return tuple(grads)


def _seam(run, name, **tensors) -> None:
    """Stage seam of the fine-stage backward pass.  A test may install `model._fused_cache['bwd_probe'] = f(name, tensors)`: it is
    called with the tensors that cross the seam, right after the launches that produced them were issued, and may read them
    (clone) or overwrite them in place (copy_) -- e.g. with the CPU oracle's gradient at the same seam, so that the NEXT stage
    runs on exactly the oracle's upstream gradient (tests/test_stagewise_bwd_gpu.py).  No probe: nothing happens."""
    probe = run.cache.get('bwd_probe')
    if probe is not None:
        probe(name, tensors)


def _fine_grad_layout(run, rgb_w, ref_w, rw, fw, ldx0, ldz):
    """(items, total): every MLP gradient of the fine stage as a 16-byte aligned view of one flat buffer (cached)."""
    lay = run.cache.get('grad_layout')
    if lay is None:
        shapes = ([tuple(w.shape) for w in rgb_w] + [tuple(w.shape) for w in ref_w] + [(w.shape[0],) for w in rgb_w] +
                  [(w.shape[0],) for w in ref_w] + [(rw, ldx0), (fw, ldz), (ldz,)])
        items, off = [], 0
        for sh in shapes:
            n = int(np.prod(sh))
            items.append((sh, n, off))
            off += (n + 3) // 4 * 4
        lay = run.cache['grad_layout'] = (items, off)
    return lay


def _early_hooks(run):
    hook = run.cache.get('grad_hook') if _LINEAR_BWD_MODE in ("one", "split", "chain") else None   # dist.GradAverager.early
    return hook, run.cache.get('opt_hook')


def supports_coarse(model) -> bool:
    """Coarse-stage configurations ('coarse', 'geometry_searching') the fused kernels cover."""
    from .nerf import mlp_layers
    if model.stage not in ('coarse', 'geometry_searching') or not (model.fast_color_thres > 0):
        return False
    if getattr(model, 'grad_mode', 'interpolate') not in ('interpolate', 'raw', 'grad_conv'):
        return False
    if model.smooth_sdf and int(model.smooth_conv.weight.shape[-1]) > 7:
        return False
    fl = mlp_layers(model.refnet)
    cols = (model.k0_dim + (3 + 6 * len(model.posfreq)) + (3 + 6 * len(model.reffreq)) + 3 +
            ((3 + 6 * len(model.viewfreq)) if model.use_viewdir else 0))
    fw = fl[0].out_features
    if cols != fl[0].in_features or fw % 4 or fw > 256 or len(fl) < 2 or fl[-1].out_features != 3:
        return False
    if any(l.out_features != fw for l in fl[:-1]):
        return False
    g = model.sdf.grid
    return g.is_cuda and g.is_contiguous() and model.k0.grid.is_cuda


class _FusedCoarse(torch.autograd.Function):
    """inputs: smoothed SDF grid [1,1,X,Y,Z], gradient volume [1,3,X,Y,Z] (both autograd nodes of dense.py over
    sdf.grid), k0 grid, then (weight, bias) of every refnet Linear."""

    @staticmethod
    def forward(ctx, run, sdf_smooth, gradvol, k0_grid, *mlp):
        if run.s_param is not None:
            mlp = mlp[:-1]              # (see _FusedFine.forward)
        ctx.set_materialize_grads(False)              # see _FusedFine.forward
        dev = sdf_smooth.device
        g, N, st, ms, ws = run.geom, run.n_rays, stream(), run.max_steps, run.workspace
        _own_workspace(run, any(ctx.needs_input_grad))
        sdf_smooth, gradvol = sdf_smooth.contiguous(), gradvol.contiguous()
        use_mc = run.mask_grid is not None
        inc = run.inc
        alphainv_last = torch.empty(N, dtype=F32, device=dev)   # an output of the march: a fresh tensor per step
        sf = run.sync_free
        call("fgs_march_coarse_fwd", ptr(run.rays_o), ptr(run.rays_d), ptr(run.viewdirs), N, g.lo_c, g.hi_c, g.X, g.Y, g.Z,
             run.near, 1e9, run.stepdist, ptr(sdf_smooth), ptr(gradvol), ptr(getattr(run, 'vol4', None)), run.dist, run.inv_s,
             run.thres, ptr(run.mask_grid), *(g.mask[:2] if use_mc else (None, None)), *(g.mask[2] if use_mc else (0, 0, 0)),
             g.mask[3] if use_mc else 0.0, ptr(inc[0]) if inc else None, *(inc[1] if inc else (0, 0, 0)),
             inc[2] if inc else None, inc[3] if inc else None, ms, ptr(ws['a_step']), ptr(ws['a_alpha']), ptr(ws['a_T']),
             ptr(ws['a_weight']), ptr(ws['a_sdf']), ptr(ws['a_grad']), ptr(ws['a_surv']), ptr(ws['surv_slot']),
             ptr(ws['n_alive']), ptr(ws['n_surv']), ptr(ws['n_inbbox']), ptr(alphainv_last), dyn(inv_s=_inv_s(run)), st)
        if sf:
            call("fgs_exclusive_scan_guard_i64", ptr(ws['n_surv']), N, ptr(ws['surv_off']), sf['capacity'], ptr(sf['flags']),
                 ptr(sf['total']), st)
        else:
            call("fgs_exclusive_scan_i64", ptr(ws['n_surv']), N, ptr(ws['surv_off']), st)
        token = None if sf else _count_begin(run, ws['surv_off'], N)
        n_ref = run.n_ref                           # queued behind the count copy: K-padded first-layer weights, k0.grad fill
        ref_w = [mlp[2 * i] for i in range(n_ref)]
        ref_b = [mlp[2 * i + 1] for i in range(n_ref)]
        fw, ldx0 = ref_w[0].shape[0], run.ldx0
        V0c = None
        rc_shapes = _MLP_IMPL == "rc" and fw % 32 == 0 and fw <= 256 and ldx0 <= 256 and n_ref - 1 <= 8
        if _DX0_COMPACT and rc_shapes and any(ctx.needs_input_grad):
            # (as in the fine stage: the first layer's weights without the xyz / view-direction encodings' columns, gathered in
            # the launch that makes the padded copy -- dX0 is computed and read as [k0 | reflect_emb | normal])
            k0d, gap, cw = run.dx0_cols
            V0 = ref_w[0].detach()
            V0c = torch.empty(fw, (cw + 3) // 4 * 4, dtype=F32, device=dev)
            V0p, _, _ = fo.pad_cols_multi([V0, V0[:, :k0d], V0[:, k0d + gap:k0d + gap + cw - k0d]], [ldx0, k0d, cw - k0d],
                                          outs=[None, V0c[:, :k0d], V0c[:, k0d:cw]])
        else:
            (V0p,) = fo.pad_cols_multi([ref_w[0].detach()], [ldx0])
        pre_k0 = _prefill_grid_grad(run, k0_grid) if (_PRE_FILL_AT_READ and any(ctx.needs_input_grad)) else None
        kC, kX, kY, kZ, ksC, ksX, ksY, ksZ = grid_strides(k0_grid)
        if sf:       # sync-free (see _FusedFine.forward): the count stays on the device, M is the CAPACITY from here on
            M = sf['capacity']
            run.count_ptr = ws['surv_off'].data_ptr() + 8 * N
        else:
            M = _count_end(token)                  # the one host read of the step
        run.M = M
        ray_id = torch.empty(M, dtype=I64, device=dev)
        step_id = torch.empty(M, dtype=I64, device=dev)
        rec_idx = torch.empty(M, dtype=I32, device=dev)
        weights = torch.empty(M, dtype=F32, device=dev)
        alpha = torch.empty(M, dtype=F32, device=dev)
        sdf = torch.empty(M, dtype=F32, device=dev)
        gradient = torch.empty(M, 3, dtype=F32, device=dev)
        pts = torch.empty(M, 3, dtype=F32, device=dev)
        call("fgs_surv_compact", N, M, ptr(ws['surv_off']), ms, ptr(ws['surv_slot']), ptr(ws['a_step']), ptr(ws['a_alpha']),
             ptr(ws['a_weight']), ptr(ws['a_sdf']), ptr(ws['a_grad']), ptr(run.rays_o), ptr(run.rays_d), g.lo_c, g.hi_c,
             g.X, g.Y, g.Z, run.near, 1e9, run.stepdist, ptr(ray_id), ptr(step_id), ptr(rec_idx), ptr(weights), ptr(alpha),
             ptr(sdf), ptr(gradient), ptr(pts), dyn(row_count=_rows(run)), st)
        ldx0 = run.ldx0
        X0 = torch.empty(M, ldx0, dtype=F32, device=dev)
        normal = torch.empty(M, 3, dtype=F32, device=dev)
        kC, kX, kY, kZ, ksC, ksX, ksY, ksZ = grid_strides(k0_grid)
        call("fgs_feat_coarse_fwd", M, ptr(ray_id), ptr(pts), ptr(gradient), ptr(run.viewdirs), g.lo_c, g.hi_c, g.X, g.Y,
             g.Z, run.layout_i, ptr(k0_grid), ksC, ksX, ksY, ksZ, ptr(X0), ptr(normal), dyn(row_count=_rows(run)), st)
        use_rc = _MLP_IMPL == "rc" and fw % 32 == 0 and fw <= 256 and ldx0 <= 256 and n_ref - 1 <= 8 and M > 0
        grp = _gemm_group("forward chain (" + ("k_mlp_rc: register-resident, all layers in one launch" if use_rc
                                               else "NT: k_gemm<true,true,0>") + ")").__enter__()
        acts = [X0]
        a = X0
        relu_bits = None
        if use_rc:       # widths 192 (coarse) and 128 (geometry_searching): the same register-resident chain as the fine stage
            relu_bits = torch.empty(n_ref - 1, fo.rc_mask_bits(M, dev).numel(), dtype=torch.int32, device=dev)
            acts += [torch.empty(M, fw, dtype=F32, device=dev) for _ in range(n_ref - 1)]
            fo.rc_chain(False, M, X0, ldx0, [dict(W=ref_w[i].detach(), bias=ref_b[i].detach(), relu=True, mask_bits=relu_bits[i],
                                                   out=acts[i + 1], n_store=fw) for i in range(n_ref - 1)],
                        flop=2.0 * M * fw * sum(w.shape[1] for w in ref_w[:-1]), rows_dev=_rows(run))
            a = acts[-1]
        else:
            for i in range(n_ref - 1):
                out = torch.empty(M, fw, dtype=F32, device=dev)
                _gemm(fo.GEMM_NT, a, V0p if i == 0 else ref_w[i].detach(), out, M, fw, ldx0 if i == 0 else fw,
                      bias=ref_b[i].detach(), relu=True, logical=(M, fw, ref_w[i].shape[1]))
                a = out
                acts.append(out)
        grp.__exit__()
        rgb = torch.empty(M, 3, dtype=F32, device=dev)
        call("fgs_head_fwd", ptr(a), a.stride(0), fw, M, ptr(ref_w[-1].detach()), ptr(ref_b[-1].detach()), ptr(rgb),
             dyn(row_count=_rows(run)), st)
        rgb_marched = torch.empty(N, 3, dtype=F32, device=dev)
        sigmoid_rgb = torch.empty(N, 3, dtype=F32, device=dev)
        pre_rgb = torch.empty(N, 3, dtype=F32, device=dev)
        pre_sig = torch.empty(N, 3, dtype=F32, device=dev)
        normal_marched = torch.empty(N, 3, dtype=F32, device=dev) if run.render_grad else None
        depth = torch.empty(N, dtype=F32, device=dev) if run.render_depth else None
        call("fgs_composite_fwd", N, ptr(ws['surv_off']), ptr(weights), ptr(rgb), ptr(normal), ptr(step_id), run.bg, run.dist,
             ptr(rgb_marched), ptr(sigmoid_rgb), ptr(pre_rgb), ptr(pre_sig), ptr(normal_marched), ptr(depth), st)
        run.pre = None                                 # backward's big zero fills, issued here (see _FusedFine.forward)
        if any(ctx.needs_input_grad) and M > 0:
            run.pre = (torch.zeros(g.X, g.Y, g.Z, 4, dtype=F32, device=dev), pre_k0)
        run.saved = _detached(dict(ray_id=ray_id, pts=pts, gradient=gradient, weights=weights, rgb=rgb, X0=X0, acts=acts,
                                   V0p=V0p, V0c=V0c, relu_bits=relu_bits, pre_rgb=pre_rgb, pre_sig=pre_sig, alphainv_last=alphainv_last,
                                   k0_strides=(ksC, ksX, ksY, ksZ)))
        run.extras = dict(step_id=step_id, rec_idx=rec_idx, normal_marched=normal_marched, depth=depth,
                          n_inbbox=ws['n_inbbox'])
        ctx.run = run
        ctx.save_for_backward(k0_grid, *mlp)
        ctx.mark_non_differentiable(ray_id, alpha, gradient)
        return rgb_marched, sigmoid_rgb, alphainv_last, weights, rgb, normal, ray_id, alpha, gradient

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        return _FusedCoarse._backward_impl(ctx, *grads)

    @staticmethod
    def _backward_impl(ctx, g_rgb_marched, g_sigmoid_rgb, g_last, g_weights, g_raw_rgb, g_normal, *_unused):
        run = ctx.run
        if run.done:
            raise RuntimeError("fused forward_coarse: backward called twice on the same forward (retain_graph is not "
                               "supported by the fused path)")
        run.done = True          # see _FusedFine.backward
        k0_grid, *mlp = ctx.saved_tensors
        S, g, N, M, st, ws = run.saved, run.geom, run.n_rays, run.M, stream(), run.workspace
        dev = k0_grid.device
        n_ref = run.n_ref
        ref_w = [mlp[2 * i] for i in range(n_ref)]
        fw, ldx0 = ref_w[0].shape[0], run.ldx0

        def c(t):
            return None if t is None else t.contiguous()
        g_rgb_marched, g_sigmoid_rgb, g_last, g_weights, g_raw_rgb, g_normal = map(
            c, (g_rgb_marched, g_sigmoid_rgb, g_last, g_weights, g_raw_rgb, g_normal))
        shapes = [tuple(w.shape) for w in ref_w] + [(w.shape[0],) for w in ref_w] + [(fw, ldx0)]
        sizes = [(int(np.prod(s)) + 3) // 4 * 4 for s in shapes]
        flat = torch.zeros(sum(sizes), dtype=F32, device=dev)
        views, off = [], 0
        for s, n in zip(shapes, sizes):
            views.append(flat[off:off + int(np.prod(s))].view(*s))
            off += n
        gw, gb, gV0p = views[:n_ref], views[n_ref:2 * n_ref], views[-1]
        if M == 0:   # no kept sample on this rank: zero local gradients, but the same hooks as every other rank (see
            run.pre = None                                  # _FusedFine._backward_empty)
            grad_k0 = torch.empty_strided(k0_grid.shape, k0_grid.stride(), dtype=F32, device=dev).zero_()
            hook, opt_hook = _early_hooks(run)
            if hook is not None:
                hook('k0', [k0_grid], grad_k0)
                hook('mlp', mlp, flat)
                hook('join', None)
            elif opt_hook is not None:
                opt_hook(k0_grid, grad_k0)
            if not (_MLP_IMPL == "rc" and fw % 32 == 0 and fw <= 256 and ldx0 <= 256 and n_ref - 1 <= 8):
                gw[0] = gV0p[:, :ref_w[0].shape[1]]          # (the GEMM path keeps dW0 in the K-padded slot)
            grads = [None, torch.zeros(1, 1, g.X, g.Y, g.Z, dtype=F32, device=dev),
                     torch.zeros(1, 3, g.X, g.Y, g.Z, dtype=F32, device=dev), grad_k0]
            for i in range(n_ref):
                grads += [gw[i].contiguous(), gb[i].contiguous()]
            if run.s_param is not None:
                grads.append(torch.zeros_like(run.s_param))
            return tuple(grads)
        d_out = torch.empty(M, 3, dtype=F32, device=dev)
        d_w = torch.empty(M, dtype=F32, device=dev)
        call("fgs_composite_bwd", M, ptr(S['ray_id']), ptr(S['weights']), ptr(S['rgb']), ptr(S['pre_rgb']), ptr(S['pre_sig']),
             ptr(g_rgb_marched), ptr(g_sigmoid_rgb), ptr(g_raw_rgb), ptr(g_weights), run.bg, ptr(d_out), ptr(d_w),
             dyn(row_count=_rows(run)), st)
        acts = S['acts']
        a_last = acts[n_ref - 1]
        dY = torch.empty(M, fw, dtype=F32, device=dev)
        call("fgs_head_bwd", ptr(a_last), a_last.stride(0), fw, M, ptr(ref_w[-1]), ptr(d_out), ptr(dY), ptr(gw[-1]),
             ptr(gb[-1]), ptr(gb[n_ref - 2]), ptr(_head_scratch(fw, dev)), dyn(row_count=_rows(run)), st)
        dX0 = None
        wgrad = None
        dx0_compact = False
        grp = _gemm_group("backward chain (" + ("rc" if S.get('relu_bits') is not None else _LINEAR_BWD_MODE) + ")").__enter__()
        if S.get('relu_bits') is not None:
            # register-resident data-gradient chain (layers n_ref-2 .. 1), dX0 as one narrow NN product, every weight / bias
            # gradient in one fgs_mlp_wgrad launch straight into the views of the flat buffer
            bits = S['relu_bits']
            dYs = [None] * (n_ref - 1)
            dYs[n_ref - 2] = dY
            layers = []
            for i in range(n_ref - 2, 0, -1):
                out = torch.empty(M, fw, dtype=F32, device=dev)
                layers.append(dict(W=ref_w[i], mask_bits=bits[i - 1], out=out, n_store=fw))
                dYs[i - 1] = out
            if layers:
                fo.rc_chain(True, M, dY, fw, layers, flop=2.0 * M * fw * fw * len(layers), rows_dev=_rows(run))
            if S.get('V0c') is not None:     # compact dX0 (fgs_dyn_t.dx0_compact)
                V0c = S['V0c']
                dX0 = torch.empty(M, V0c.shape[1], dtype=F32, device=dev)
                _gemm(fo.GEMM_NN, dYs[0], V0c, dX0, M, V0c.shape[1], fw, logical=(M, run.dx0_cols[2], fw), rows_dev=_rows(run))
                dx0_compact = True
            else:
                dX0 = torch.empty(M, ldx0, dtype=F32, device=dev)
                _gemm(fo.GEMM_NN, dYs[0], S['V0p'], dX0, M, ldx0, fw, logical=(M, ref_w[0].shape[1], fw), rows_dev=_rows(run))
            wg_items = [(dYs[i], acts[i], gw[i], None if i == n_ref - 2 else gb[i], fw, ref_w[i].shape[1])
                        for i in range(n_ref - 1)]
            wgrad = lambda fork: _wgrad(dev, M, wg_items, 2.0 * M * fw * sum(w.shape[1] for w in ref_w[:-1]), fork,
                                        rows_dev=_rows(run))
        else:
            for i in range(n_ref - 2, -1, -1):
                a_in = acts[i]
                if i == 0:
                    dX0 = torch.empty(M, ldx0, dtype=F32, device=dev)
                    _linear_bwd(dY, S['V0p'], a_in, dX0, gV0p, M, fw, ldx0, logical_k_in=ref_w[0].shape[1])
                else:
                    d_in = torch.empty(M, fw, dtype=F32, device=dev)
                    _linear_bwd(dY, ref_w[i], a_in, d_in, gw[i], M, fw, fw, mask=a_in, colsum=gb[i - 1])
                    dY = d_in
            gw[0] = gV0p[:, :ref_w[0].shape[1]]
        grp.__exit__()
        _flush_tn(dev)
        hook, opt_hook = _early_hooks(run)
        forked = False
        if wgrad is not None and (hook is None or (_WGRAD_FORK and _WGRAD_FORK_DIST)):
            wgrad(True)
            wgrad = None
            forked = hook is not None
        if run.pre is not None:
            d4, pre_k0 = run.pre
            run.pre = None
        else:
            d4, pre_k0 = torch.zeros(g.X, g.Y, g.Z, 4, dtype=F32, device=dev), None
        grad_k0, k0_state = pre_k0 if pre_k0 is not None else _take_grid_grad(run.cache, k0_grid)
        g_grad_s = torch.empty(M, 3, dtype=F32, device=dev)
        ksC, ksX, ksY, ksZ = S['k0_strides']
        call("fgs_feat_coarse_bwd", M, ptr(S['ray_id']), ptr(S['pts']), ptr(S['gradient']), ptr(run.viewdirs), g.lo_c,
             g.hi_c, g.X, g.Y, g.Z, run.layout_i, ptr(S['X0']), ptr(dX0), ptr(g_normal), ptr(grad_k0), ksC, ksX, ksY, ksZ,
             ptr(g_grad_s), dyn(row_count=_rows(run), compact=dx0_compact), st)
        _publish_touched(k0_state, k0_grid, grad_k0, S['pts'], M, g, st, exchange=hook is not None, rows_dev=_rows(run))
        if hook is not None:                     # (k0, mlp, join: the order of every path, see _FusedFine)
            hook('k0', [k0_grid], grad_k0)
            _exchange_mlp(dev, wgrad, forked, hook, mlp, flat)
        elif opt_hook is not None:
            opt_hook(k0_grid, grad_k0)
        # d4: voxel-interleaved accumulation buffer [X,Y,Z,4]; the two dense adjoints (dense.py) read their channel(s) of
        # it in place through element strides
        g_inv_s = torch.zeros(1, dtype=F32, device=dev) if run.s_param is not None else None
        call("fgs_march_coarse_bwd", ptr(run.rays_o), ptr(run.rays_d), ptr(run.viewdirs), N, g.lo_c, g.hi_c, g.X, g.Y, g.Z,
             run.near, 1e9, run.stepdist, run.dist, run.inv_s, run.max_steps, ptr(ws['a_step']), ptr(ws['a_alpha']),
             ptr(ws['a_T']), ptr(ws['a_weight']), ptr(ws['a_sdf']), ptr(ws['a_grad']), ptr(ws['n_alive']), ptr(ws['n_surv']),
             ptr(ws['surv_off']), ptr(S['alphainv_last']), ptr(d_w), ptr(g_last), ptr(g_grad_s), ptr(d4), ptr(g_inv_s),
             dyn(inv_s=_inv_s(run)), st)
        d_smooth = d4[..., 0][None, None]                       # [1,1,X,Y,Z], element stride 4
        d_gradvol = d4[..., 1:4].permute(3, 0, 1, 2)[None]      # [1,3,X,Y,Z], channel stride 1, voxel stride 4
        _join_side(dev)
        if hook is not None:
            hook('join', None)
        grads: List[Optional[torch.Tensor]] = [None, d_smooth, d_gradvol, grad_k0]
        for i in range(n_ref):
            grads += [gw[i].contiguous(), gb[i].contiguous()]
        if run.s_param is not None:
            grads.append((-g_inv_s / run.s_param.detach().to(dev).float() ** 2).reshape(run.s_param.shape))
        return tuple(grads)


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


def forward_coarse(model, rays_o, rays_d, viewdirs, global_step=20000, **render_kwargs):
    """nerf.forward_coarse (model/nerf.py:943-1075) through the fused kernels; same ret_dict."""
    from . import dense
    from .nerf import mlp_layers
    dev = rays_o.device
    run, s_val = _setup_run(model, rays_o, rays_d, viewdirs, global_step, render_kwargs, default_depth=True)
    N, g = run.n_rays, run.geom
    fl = mlp_layers(model.refnet)
    run.n_ref = len(fl)
    cols = fl[0].in_features
    run.ldx0 = (cols + 3) // 4 * 4
    run.layout_i = (ctypes.c_int * 6)(model.k0_dim, len(model.posfreq), len(model.viewfreq), len(model.reffreq),
                                      int(model.use_viewdir), run.ldx0)
    # compact dX0 (_DX0_COMPACT): columns [k0 | reflect_emb | normal] of [k0, xyz_emb, reflect_emb, normal, viewdirs_emb]:
    # (k0 columns, width of the xyz block behind them, compact width)
    gap_c = 3 + 6 * len(model.posfreq)
    run.dx0_cols = (int(model.k0_dim), gap_c, int(model.k0_dim) + (3 + 6 * len(model.reffreq)) + 3)
    # the mask cache only prunes in stage 'coarse' (model/nerf.py:951)
    run.mask_grid = model.mask_cache.sdf_mask if (model.stage == 'coarse' and model.mask_cache is not None) else None
    run.inc = None
    if model.inc_mask is not None:
        im = model.inc_mask
        key = im                                   # the module itself: keeps it alive, so no id() reuse
        cached = model.__dict__.get('_fused_inc')
        if cached is None or cached[0] is not key:
            world = im.mask.to(torch.uint8).contiguous()
            sc = im.xyz2ijk_scale.detach().cpu().float().tolist()
            sh = im.xyz2ijk_shift.detach().cpu().float().tolist()
            cached = (key, (world, tuple(int(s) for s in world.shape), (ctypes.c_float * 3)(*sc), (ctypes.c_float * 3)(*sh)))
            model.__dict__['_fused_inc'] = cached   # plain attribute, not a registered sub-module
        run.inc = cached[1]
    # dense per-iteration volumes (row a6): smoothed SDF grid and central-difference gradient volume, both autograd
    # nodes over sdf.grid; model.gradient stays differentiable for density_total_variation (model/nerf.py:440-446)
    if model.smooth_sdf:
        taps = getattr(model, '_fused_taps', None)
        if taps is None or taps[0] is not model.smooth_conv:
            taps = (model.smooth_conv, dense._taps_c(model.smooth_conv.weight))
            model._fused_taps = taps
        sdf_smooth = dense.smooth3d(model.sdf.grid, model.smooth_conv.weight, taps[1])
    else:
        sdf_smooth = model.sdf.grid
    # (the gradient-volume pass also leaves the voxel-interleaved copy {smoothed sdf, g_x, g_y, g_z} the march samples with
    # one 16-byte load per trilinear corner: FGS_COARSE_VOL4=0 switches it off)
    holder = {}
    gmode = getattr(model, 'grad_mode', 'interpolate')
    model.gradient = dense.sdf_gradient_volume(model.sdf.grid, g.voxel_size, sdf_smooth if (_COARSE_VOL4 and gmode != 'grad_conv') else None,
                                               holder, mode=gmode,
                                               grad_conv_weight=model.grad_conv.weight if gmode == 'grad_conv' else None)
    run.vol4 = holder.get('vol4')
    mlp = []
    for layer in fl:
        mlp += [layer.weight, layer.bias]
    fw_ = fl[0].out_features
    if run.sync_free and not (_MLP_IMPL == "rc" and fw_ % 32 == 0 and fw_ <= 256 and run.ldx0 <= 256 and len(fl) - 1 <= 8):
        raise RuntimeError("the sync-free coarse-stage path needs the register-resident MLP kernels (FGS_MLP=rc, refnet width "
                           "a multiple of 32, <= 256)")
    if run.s_param is not None:
        mlp = mlp + [run.s_param]
    (rgb_marched, sigmoid_rgb, alphainv_last, weights, rgb, normal, ray_id, alpha, gradient) = _FusedCoarse.apply(
        run, sdf_smooth, model.gradient, model.k0.grid, *mlp)
    ex = run.extras
    depth = ex['depth']

    def lazy_outbbox():
        with torch.no_grad():
            pts, _, _, mask_outbbox, _ = model.sample_ray(rays_o=rays_o, rays_d=rays_d, **render_kwargs)
            if run.mask_grid is not None:
                mask_outbbox[~mask_outbbox] |= ~model.mask_cache(pts)
        return mask_outbbox

    def lazy_mask():
        """`weights > thres` of the first Alphas2Weights over the (mask-cache / inc-mask filtered) sample list
        (model/nerf.py:982), rebuilt with the operator-at-a-time path only when somebody reads it."""
        keep = model.gradient
        with torch.no_grad():
            mask = model._forward_coarse_composed(rays_o, rays_d, viewdirs, global_step, **render_kwargs)['mask']
        model.gradient = keep
        return mask

    eager = {'alphainv_cum': alphainv_last, 'weights': weights, 'ray_id': ray_id,
             'rgb_marched': rgb_marched, 'sigmoid_rgb': sigmoid_rgb, 'normal_marched': ex['normal_marched'],
             'normal': normal, 'raw_alpha': alpha, 'raw_rgb': rgb, 'depth': depth,
             'disp': None if depth is None else 1 / depth, 'gradient': gradient, 's_val': s_val,
             'step_id': ex['step_id'], 'n_inbbox_visited': ex['n_inbbox'], 'ray_viewdirs': run.viewdirs,
             'survivor_pts': run.saved['pts'],
             'survivor_count_ptr': run.count_ptr}       # sync-free mode: see forward_fine
    return LazyResult(eager, {'mask': lazy_mask, 'mask_outbbox': lazy_outbbox,
                              'viewdirs': lambda: run.viewdirs[ray_id]})     # per-sample gather only when somebody reads it


def forward_fine(model, rays_o, rays_d, viewdirs, global_step=20000, **render_kwargs):
    from .nerf import mlp_layers
    dev = rays_o.device
    run, s_val = _setup_run(model, rays_o, rays_d, viewdirs, global_step, render_kwargs, default_depth=False)
    N = run.n_rays
    run.layout_i, run.displace, run.ldx0, run.ldz, x0_cols = _layout(model, run.geom)
    # (k0 columns, width of the xyz + view-direction encodings behind them, columns of X0 without those: see _DX0_COMPACT)
    gap = (3 + 6 * len(model.posfreq)) + ((3 + 6 * len(model.viewfreq)) if model.use_viewdir else 0)
    run.dx0_cols = (int(model.k0_dim), gap, x0_cols - gap)
    run.mask_grid = model.mask_cache.sdf_mask if model.mask_cache is not None else None
    rl, fl = mlp_layers(model.rgbnet), mlp_layers(model.refnet)
    run.n_rgb, run.n_ref = len(rl), len(fl)
    mlp = []
    for layer in rl + fl:
        mlp += [layer.weight, layer.bias]
    if run.sync_free and not _rc_eligible(rl[0].out_features, fl[0].out_features, run.ldx0, run.ldz, len(rl), len(fl)):
        raise RuntimeError("the sync-free fine-stage path needs the register-resident MLP kernels (FGS_MLP=rc and equal "
                           "rgbnet / refnet widths that are multiples of 32, <= 256)")
    # model/nerf.py:791: every lookup of the fine stage samples the smoothed grid when smooth_sdf is set (an autograd node over
    # sdf.grid, csrc/dense.hip); model.gradient stays the gradient volume of the RAW grid (model/nerf.py:856)
    sdf_in = model.sdf.grid
    if model.smooth_sdf:
        from . import dense
        taps = getattr(model, '_fused_taps', None)
        if taps is None or taps[0] is not model.smooth_conv:
            taps = (model.smooth_conv, dense._taps_c(model.smooth_conv.weight))
            model._fused_taps = taps
        sdf_in = dense.smooth3d(model.sdf.grid, model.smooth_conv.weight, taps[1])
    run.sdf_in = sdf_in.detach()
    if run.s_param is not None:
        mlp = mlp + [run.s_param]
    (rgb_marched, sigmoid_rgb, alphainv_last, weights, rgb, normal, ray_id, alpha, gradient) = _FusedFine.apply(
        run, sdf_in, model.k0.grid, *mlp)
    ex = run.extras
    depth = ex['depth']

    def lazy_masks():
        """The reference's per-sample masks, recomputed with the operator-at-a-time kernels only when asked for."""
        with torch.no_grad():
            _, _, _, mask_outbbox, _ = model.sample_ray(rays_o=rays_o, rays_d=rays_d, **render_kwargs)
        return mask_outbbox

    def _current_sdf_in():
        """the grid the lookups sample NOW (the training loop reads 'mask' after optimizer.step(): on the updated grid)"""
        if not model.smooth_sdf:
            return model.sdf.grid
        from . import dense
        with torch.no_grad():
            return dense.smooth3d(model.sdf.grid.detach(), model.smooth_conv.weight, model._fused_taps[1])

    def lazy_mask():
        """`weights > thres` over the reference's alpha-compacted list (model/nerf.py:825): per ray the alive records come
        first (survivors flagged), the samples behind the terminating one follow (all False)."""
        g, ws = run.geom, run.workspace
        if ws['gen'] != run.gen:
            raise RuntimeError("result['mask'] of the fused forward_fine must be read before the next forward with the same "
                               "ray count (its per-ray records have been overwritten)")
        n_m1 = torch.empty(N, dtype=I64, device=dev)
        n_in = torch.empty(N, dtype=I64, device=dev)
        call("fgs_march_count", ptr(run.rays_o), ptr(run.rays_d), ptr(run.viewdirs), N, g.lo_c, g.hi_c, g.X, g.Y, g.Z,
             g.voxel_size, run.near, 1e9, run.stepdist, ptr(_current_sdf_in()), run.dist, run.inv_s, run.thres,
             ptr(run.mask_grid), *(g.mask[:2] if g.mask else (None, None)), *(g.mask[2] if g.mask else (0, 0, 0)),
             g.mask[3] if g.mask else 0.0, run.max_steps, ptr(n_m1), ptr(n_in), None, stream())
        # the training loop reads 'mask' after optimizer.step() (nerf_training.py:373-381), i.e. on an updated sdf grid:
        # never let a ray's list be shorter than its alive segment of this forward (valid until the next forward)
        n_m1 = torch.maximum(n_m1, ws['n_alive'])
        off = torch.cumsum(n_m1, 0) - n_m1
        mask = torch.zeros(int(n_m1.sum().item()), dtype=torch.bool, device=dev)
        mask[off[ray_id] + ex['rec_idx'].long()] = True
        return mask

    eager = {'alphainv_cum': alphainv_last, 'weights': weights, 'ray_id': ray_id,
             'rgb_marched': rgb_marched, 'sigmoid_rgb': sigmoid_rgb, 'normal_marched': ex['normal_marched'],
             'normal': normal, 'raw_alpha': alpha, 'raw_rgb': rgb, 'depth': depth,
             'disp': None if depth is None else 1 / depth, 'gradient': gradient, 's_val': s_val,
             'step_id': ex['step_id'], 'n_inbbox_visited': ex['n_inbbox'], 'ray_viewdirs': run.viewdirs,
             'survivor_pts': run.saved['pts'],
             # sync-free mode: the per-survivor entries above have CAPACITY rows; the rows that count are the first
             # *survivor_count_ptr (a device int64), which consumers pass on as fgs_dyn_t.row_count
             'survivor_count_ptr': run.count_ptr}
    return LazyResult(eager, {'mask': lazy_mask, 'mask_outbbox': lazy_masks, 'viewdirs': lambda: run.viewdirs[ray_id]})


def roofline_report(pmc=None, flop_scale: float = 1.0):
    """Achieved fp32 FLOP/s of the dominant kernels -- the MLP matrix-core kernels: k_mlp_rc (register-resident forward chain
    and backward data-gradient chain, one launch each), k_mlp_wgrad (all weight / bias gradients, one launch), k_gemm (the two
    first-layer data gradients) -- from the HIP events recorded around every uninterrupted run of them, against the gfx950
    fp32 matrix-core peak (MI355X_MICROARCH.md: 157.3 TFLOP/s, v_mfma_f32_32x32x2_f32 at 256 FLOP/clk/CU, 2.4 GHz).
    `pmc`: bench.pmc_traffic_live()'s per-kernel HBM bytes (or None / {'error': ...}): `traffic` is then the mean over the
    launches of one step, measured in this run; without it `traffic` is null (never a number from another run)."""
    ev = PROFILE["gemm_events"] + fo.TIMING["events"]
    if not ev:
        return None
    per = {}
    tot_ms, tot_fl, tot_n = 0.0, 0.0, 0
    for e0, e1, e1s, label, n, fl in ev:
        fl = fl * flop_scale
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
           "kernel": "MLP matrix-core kernels, fp32 v_mfma_f32_32x32x2_f32: k_mlp_rc (forward chain / backward data-gradient "
                     "chain, activations resident in registers, one launch each), k_mlp_wgrad (all weight and bias gradients, "
                     "one launch), k_gemm (the two first-layer data gradients)",
           "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
           "traffic": traffic, "traffic_unit": "HBM bytes per launch, mean over the MLP launches of a step",
           "traffic_detail": detail,
           "launches": tot_n, "avg_launch_us": round(tot_ms * 1e3 / tot_n, 2),
           "algorithmic_gflop_per_launch": round(tot_fl / tot_n / 1e9, 3),
           "timing": "HIP events on the launch stream around each uninterrupted MLP chain in the timed region",
           "chains": {k: {"launches": v[0], "avg_us": round(v[1] * 1e3 / v[0], 2),
                            "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2)} for k, v in per.items()}}
    return out
