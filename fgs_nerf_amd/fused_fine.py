"""Fused MI355X path for ``nerf.forward_fine`` (model/nerf.py:776-941): one autograd node per forward, ~20 HIP launches each way
(module docstring of fused.py has the launch sequence)."""
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
from .fused_common import *      # noqa: F401,F403  (every shared helper, underscore names included: fused_common.__all__)


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


def _rc2_bwd_weights(n_rgb, n_ref, rgb_w, ref_w, rw, V0p, W0c, W0c_views=None):
    """The layer list of the form-2 backward chain as far as its WEIGHTS go (the order _backward_rc issues: refnet main layers,
    the reflection-encoding side layer, refnet layer 0, rgbnet main layers, the compact dX0 side layer) -- what the combined pack
    launch of the forward pass needs (fused_ops.rc2_pack)."""
    L = [dict(W=ref_w[i].detach()) for i in range(n_ref - 2, 0, -1)]
    L.append(dict(W=(V0p if V0p is not None else ref_w[0].detach())[:, rw:], side=True))
    L.append(dict(W=ref_w[0].detach()[:, :rw]))
    L += [dict(W=rgb_w[i].detach()) for i in range(n_rgb - 1, 0, -1)]
    if W0c_views is not None:
        L += [dict(W=v, side=True) for v in W0c_views]
    elif W0c is not None and W0c.shape[1] <= 64:
        L.append(dict(W=W0c, side=True))
    return L


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
    form = S.get('rc_form') or 1
    z_cols, x_cols = ref_w[0].shape[1], rgb_w[0].shape[1]
    layers = []
    dY_ref = [None] * (n_ref - 1)            # dY_ref[i]: gradient w.r.t. the pre-activation output of refnet layer i
    dY_ref[n_ref - 2] = dY
    for i in range(n_ref - 2, 0, -1):        # g . W_i, masked by the ReLU of layer i - 1
        out = torch.empty(M, fw, dtype=F32, device=dev)
        layers.append(dict(W=ref_w[i], mask_bits=bits[n_rgb + i - 1], out=out, n_store=fw))
        dY_ref[i - 1] = out
    dZ = torch.empty(M, ldz, dtype=F32, device=dev)
    flop_side = 0.0
    if form == 2:
        # side layer: the reflection-encoding columns of dZ = dY_ref[0] . V0[:, rw:], from the carried gradient BEFORE the next
        # main layer replaces it (the first form leaves this product and dX0 to k_gemm launches behind the chain)
        V0_enc = (S['V0p'] if S.get('V0p') is not None else ref_w[0].detach())[:, rw:]
        layers.append(dict(W=V0_enc, out=dZ[:, rw:], n_store=ldz - rw, side=True))
        flop_side += 2.0 * M * fw * (z_cols - rw)
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
    if form == 2 and S.get('W0c_views') is not None:
        # ... and dX0 in compact form (fgs_dyn_t.dx0_compact) as the chain's last two side layers: the k0 columns and the columns
        # behind the encodings of the fixed ray inputs, straight from the first rgbnet layer's weight
        va, vb = S['W0c_views']
        dX0 = torch.empty(M, (va.shape[1] + vb.shape[1] + 3) // 4 * 4, dtype=F32, device=dev)
        layers.append(dict(W=va, out=dX0, n_store=va.shape[1], side=True))
        layers.append(dict(W=vb, out=dX0[:, va.shape[1]:], n_store=vb.shape[1], side=True))
        flop_side += 2.0 * M * rw * run.dx0_cols[2]
        run.dx0_compact = True
        fo.rc_chain(True, M, dY, fw, layers, flop=flop_chain + flop_side, rows_dev=_rows(run), form=2, prepacked=S.get('rc2_token'))
    elif form == 2 and S.get('W0c') is not None and S['W0c'].shape[1] <= 64:
        # ... and dX0 in compact form (fgs_dyn_t.dx0_compact) as the chain's last (side) layer
        W0c = S['W0c']
        dX0 = torch.empty(M, W0c.shape[1], dtype=F32, device=dev)
        layers.append(dict(W=W0c, out=dX0, n_store=W0c.shape[1], side=True))
        flop_side += 2.0 * M * rw * run.dx0_cols[2]
        run.dx0_compact = True
        fo.rc_chain(True, M, dY, fw, layers, flop=flop_chain + flop_side, rows_dev=_rows(run), form=2, prepacked=S.get('rc2_token'))
    else:
        fo.rc_chain(True, M, dY, fw, layers, flop=flop_chain + flop_side, rows_dev=_rows(run), form=form)
        dX0 = None
    # narrow products: the reflection-encoding columns of dZ (dY_ref[0] . V0[:, rw:]) and dX0 (dY_rgb[0] . W0).  (As one-layer
    # register-resident chains of 4 row tiles they measured 63 us each against 47 for the tiled GEMM: with 64 MFMAs per chunk
    # the chain's per-chunk barrier / DMA and its uncoalesced input load dominate.)
    if form != 2:
        _gemm(fo.GEMM_NN, dY_ref[0], S['V0p'][:, rw:], dZ[:, rw:], M, ldz - rw, fw, logical=(M, z_cols - rw, fw), rows_dev=_rows(run))
    if dX0 is not None:
        pass
    elif S.get('W0c') is not None:
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
        W0c_views = None
        want_rc2 = _rc2_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) and not (_MLP_COLLAPSE and n_rgb >= 2 and n_ref >= 2)
        if want_rc2 and _DX0_COMPACT and run.dx0_cols[0] % 4 == 0 and (run.dx0_cols[2] - run.dx0_cols[0]) % 4 == 0:
            # the feature-split chains read the parameters themselves (their pack launch gathers element by element: no pitch or
            # alignment to provide), the compact dX0 as two side layers over column ranges of the first rgbnet layer: no copies
            k0d, gap, cw = run.dx0_cols
            W0 = rgb_w[0].detach()
            W0c_views = (W0[:, :k0d], W0[:, k0d + gap:])
            W0p = V0p = None
        elif _DX0_COMPACT and _rc_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) and any(ctx.needs_input_grad):
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
        k0_join = run.cache.pop('pre_k0_read', None)      # (CapturedFineStep: the previous iteration's k0 update, issued beside the
        if k0_join is not None:                           #  march kernel on a side branch, joins here: first reader of k0)
            k0_join()
        call("fgs_feat_fine_fwd", M, ptr(ray_id), ptr(pts), ptr(sdf), ptr(gradient), ptr(run.viewdirs), g.lo_c, g.hi_c,
             g.X, g.Y, g.Z, g.voxel_size, run.layout_i, run.displace, ptr(sdf_grid), ptr(k0_grid), ksC, ksX, ksY, ksZ,
             ptr(X0), ptr(Z), ptr(normal), dyn(row_count=_rows(run)), st)
        # 4. MLPs
        use_rc = _rc_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) and M > 0
        if V0p is None and not use_rc:       # (no survivors at all: the fallback products below take the padded copies)
            W0p, V0p = fo.pad_cols_multi([rgb_w[0].detach(), ref_w[0].detach()], [ldx0, ldz])
            W0c_views = None
        one_launch = (not use_rc and _MLP_FWD_ONE_LAUNCH and rw == 256 and fw == 256 and ldx0 <= 128 and 0 < ldz - rw <= 64 and
                      n_rgb + n_ref - 1 <= 8)
        grp = _gemm_group("forward chain (" + ("k_mlp_rc: register-resident, all layers in one launch" if use_rc else
                                               "k_mlp_fwd: all layers in one launch" if one_launch
                                               else "NT: k_gemm<true,true,0>") + ")").__enter__()
        acts_rgb = [X0] + [torch.empty(M, rw, dtype=F32, device=dev) for _ in range(n_rgb - 1)]   # input of each rgbnet layer
        acts_ref = [Z] + [torch.empty(M, fw, dtype=F32, device=dev) for _ in range(n_ref - 1)]    # input of each refnet layer
        relu_bits = None
        rgb = None
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
            rc_form = 2 if (_rc2_eligible(rw, fw, ldx0, ldz, n_rgb, n_ref) and not collapse) else 1
            rc2_token = None
            if rc_form == 2 and _RC2_HEAD and ref_w[-1].shape[0] <= 4:
                # the 256 -> 3 output head + sigmoid as the chain's last (side) layer: no k_head_fwd launch, no re-read of the
                # last hidden activation
                rgb = torch.empty(M, ref_w[-1].shape[0], dtype=F32, device=dev)
                layers.append(dict(W=ref_w[-1].detach(), bias=ref_b[-1].detach(), out=rgb, n_store=ref_w[-1].shape[0], side=2))
                flop_fwd += 2.0 * M * fw * ref_w[-1].shape[0]
            if rc_form == 2 and any(ctx.needs_input_grad):
                # the weight images of BOTH chains of this step in one launch, here (the backward chain then starts without one)
                bwd_w = _rc2_bwd_weights(n_rgb, n_ref, rgb_w, ref_w, rw, V0p, W0c, W0c_views)
                rc2_token = fo.rc2_pack(layers, False, ldx0, bwd_w, True, fw, device=dev)
            fo.rc_chain(False, M, X0, ldx0, layers, flop=flop_fwd, rows_dev=_rows(run), form=rc_form, prepacked=rc2_token)
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
        if rgb is None:
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
        run.fused_loss = _composite(run, any(ctx.needs_input_grad), N, M, ws['surv_off'], weights, rgb, normal, step_id, alphainv_last,
                                    rgb_marched, sigmoid_rgb, pre_rgb, pre_sig, normal_marched, depth)
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
                                   acts_rgb=acts_rgb, acts_ref=acts_ref, W0p=W0p, V0p=V0p, W0c=W0c, W0c_views=W0c_views, WT=WT, relu_bits=relu_bits,
                                   Wc_full=(Wc_full if use_rc else None), rc_form=(rc_form if use_rc else None),
                                   rc2_token=(rc2_token if use_rc else None),
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
        fl = getattr(run, 'fused_loss', None)
        if fl is not None and fl['used']:
            # the forward pass already ran the loss and the compositing backward (fgs_fine_render_loss): the gradients that arrive
            # here are the loss node's placeholders; the real ones are in the stash
            _check_only_announced_loss(fl, g_rgb_marched, g_sigmoid_rgb, g_last, g_weights, g_raw_rgb, g_normal)
            d_out, d_w, g_normal, g_last = fl['d_out'], fl['d_w'], fl['g_normal'], fl['g_last']
        else:
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

        _join_side_or_defer(run, dev, allowed=hook is None)
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
             '_fused_loss': getattr(run, 'fused_loss', None),
             'survivor_pts': run.saved['pts'],
             # sync-free mode: the per-survivor entries above have CAPACITY rows; the rows that count are the first
             # *survivor_count_ptr (a device int64), which consumers pass on as fgs_dyn_t.row_count
             'survivor_count_ptr': run.count_ptr}
    return LazyResult(eager, {'mask': lazy_mask, 'mask_outbbox': lazy_masks, 'viewdirs': lambda: run.viewdirs[ray_id]})


