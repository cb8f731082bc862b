"""Fused MI355X path for ``nerf.forward_coarse`` (model/nerf.py:943-1075), stages 'coarse' and 'geometry_searching': the same
survivor / MLP / compositing kernels as the fine stage behind a march kernel that samples the dense smoothed-SDF and gradient
volumes (csrc/march_coarse.hip, csrc/dense.hip)."""
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
from .fused_common import *      # noqa: F401,F403


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
            # (FGS_MLP_FORM_COARSE=2: the feature-split form, csrc/mlp_rc2.hip -- 6 / 4 feature tiles at widths 192 / 128.  Measured,
            # round 4, bench.py --stage coarse on one box: forward chain 159 us against 130, backward chain with the compact dX0 as a
            # side layer 131 against 100 + 48, step 1.1245 against 1.1142 ms -- with 12 sample tiles per CU neither form is
            # quantised here, and 16 + 8 MFMAs per k-group (one tile per wave + the dealt remainder) carry the same per-group and
            # per-layer fixed costs as the fine stage's 32: the register-resident form stays the default)
            rc_form = 2 if (os.environ.get("FGS_MLP_FORM_COARSE", "1") == "2" and fw in (128, 192, 256) and n_ref - 1 <= 8) else 1
            fo.rc_chain(False, M, X0, ldx0, [dict(W=ref_w[i].detach(), bias=ref_b[i].detach(), relu=True, mask_bits=relu_bits[i],
                                                   out=acts[i + 1], n_store=fw) for i in range(n_ref - 1)],
                        flop=2.0 * M * fw * sum(w.shape[1] for w in ref_w[:-1]), rows_dev=_rows(run), form=rc_form)
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
        run.fused_loss = _composite(run, any(ctx.needs_input_grad), N, M, ws['surv_off'], weights, rgb, normal, step_id, alphainv_last,
                                    rgb_marched, sigmoid_rgb, pre_rgb, pre_sig, normal_marched, depth)
        run.pre = None                                 # backward's big zero fills, issued here (see _FusedFine.forward)
        if any(ctx.needs_input_grad) and M > 0:
            run.pre = (torch.zeros(g.X, g.Y, g.Z, 4, dtype=F32, device=dev), pre_k0)
        run.saved = _detached(dict(ray_id=ray_id, pts=pts, gradient=gradient, weights=weights, rgb=rgb, X0=X0, acts=acts,
                                   V0p=V0p, V0c=V0c, relu_bits=relu_bits, rc_form=(rc_form if use_rc else None), pre_rgb=pre_rgb, pre_sig=pre_sig, alphainv_last=alphainv_last,
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
        fl = getattr(run, 'fused_loss', None)
        if fl is not None and fl['used']:
            # (the forward pass ran the loss and the compositing backward already: fused_common._composite)
            _check_only_announced_loss(fl, g_rgb_marched, g_sigmoid_rgb, g_last, g_weights, g_raw_rgb, g_normal)
            d_out, d_w, g_normal, g_last = fl['d_out'], fl['d_w'], fl['g_normal'], fl['g_last']
        else:
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
            form = S.get('rc_form') or 1
            flop_b = 2.0 * M * fw * fw * len(layers)
            if form == 2 and layers and S.get('V0c') is not None and S['V0c'].shape[1] <= 64:
                # the compact dX0 rides in the chain as its last (side) layer
                V0c = S['V0c']
                dX0 = torch.empty(M, V0c.shape[1], dtype=F32, device=dev)
                layers.append(dict(W=V0c, out=dX0, n_store=V0c.shape[1], side=True))
                flop_b += 2.0 * M * fw * run.dx0_cols[2]
                dx0_compact = True
            if layers:
                fo.rc_chain(True, M, dY, fw, layers, flop=flop_b, rows_dev=_rows(run), form=form)
            if dX0 is not None:
                pass
            elif S.get('V0c') is not None:     # compact dX0 (fgs_dyn_t.dx0_compact)
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
        _join_side_or_defer(run, dev, allowed=hook is None)
        if hook is not None:
            hook('join', None)
        grads: List[Optional[torch.Tensor]] = [None, d_smooth, d_gradvol, grad_k0]
        for i in range(n_ref):
            grads += [gw[i].contiguous(), gb[i].contiguous()]
        if run.s_param is not None:
            grads.append((-g_inv_s / run.s_param.detach().to(dev).float() ** 2).reshape(run.s_param.shape))
        return tuple(grads)


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
            # (a contiguous bool tensor is one byte per voxel already: share it, so that a captured iteration of the
            # voxel-increment phase, which rewrites the mask in place -- fgs_box_mask_fill -- is seen here)
            world = im.mask.view(torch.uint8) if (im.mask.dtype == torch.bool and im.mask.is_contiguous()) \
                else im.mask.to(torch.uint8).contiguous()
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
    model.gradient = dense.sdf_gradient_volume(model.sdf.grid, g.voxel_size, sdf_smooth if (FLAGS['coarse_vol4'] and gmode != 'grad_conv') else None,
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
             '_fused_loss': getattr(run, 'fused_loss', None),
             'survivor_pts': run.saved['pts'],
             'survivor_count_ptr': run.count_ptr}       # sync-free mode: see forward_fine
    return LazyResult(eager, {'mask': lazy_mask, 'mask_outbbox': lazy_outbbox,
                              'viewdirs': lambda: run.viewdirs[ray_id]})     # per-sample gather only when somebody reads it


