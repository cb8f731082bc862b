"""fgs_fine_render_loss (csrc/losses.hip k_render_loss: compositing + loss terms + their gradients + compositing backward in one
launch) against the four launches it replaces -- fgs_composite_fwd, fgs_fine_loss_fwd, fgs_fine_loss_bwd, fgs_composite_bwd -- on
random survivors with empty rays, clamped pixels (pre-clamp values outside [0, 1]), back-facing normals and a non-unit seed:
every per-ray and per-survivor output BIT-IDENTICAL (same expressions, -ffp-contract=off), the scalar to 1e-6 (another summation
order).  The four launches themselves are held to the oracle elsewhere (tests/test_fused_gpu.py, test_stagewise_bwd_gpu.py)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_rays,seed_val,w5", [(300, 1.0, (1.0, 0.05, 0.001, 1e-4, 0.02)), (4096, 0.37, (1.0, 0.0, 0.001, 1e-4, 0.02)),
                                                (7, 1.0, (1.0, 0.2, 0.0, 0.0, 0.0))])
def test_fused_render_loss_matches_the_four_launches(dev, n_rays, seed_val, w5):
    from fgs_nerf_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(n_rays)
    N = n_rays
    counts = torch.randint(0, 40, (N,), generator=g)
    counts[::7] = 0                                           # empty rays
    off = torch.zeros(N + 1, dtype=torch.int64)
    off[1:] = counts.cumsum(0)
    M = int(off[-1])
    ray_id = torch.repeat_interleave(torch.arange(N), counts)
    weights = torch.rand(M, generator=g) * 0.2
    weights[::5] *= 8.0                                       # some rays composite beyond 1: the clamp gate closes
    rgb = torch.sigmoid(torch.randn(M, 3, generator=g) * 2)
    normal = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
    step_id = torch.randint(0, 500, (M,), generator=g)
    viewdirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    target = torch.rand(N, 3, generator=g)
    last = torch.rand(N, generator=g)
    t = {k: v.to(dev).contiguous() for k, v in dict(off=off, ray_id=ray_id, weights=weights, rgb=rgb, normal=normal, step_id=step_id,
                                                    viewdirs=viewdirs, target=target, last=last).items()}
    bg, dist = 1.0, 0.00625
    W5 = (ctypes.c_float * 5)(*w5)
    seed = torch.tensor(seed_val, device=dev)
    st = stream()

    def e(*shape):
        return torch.full(shape, float('nan'), device=dev)
    # ---- the four launches
    A = dict(rm=e(N, 3), sr=e(N, 3), pre=e(N, 3), pres=e(N, 3), nm=e(N, 3), dep=e(N))
    call("fgs_composite_fwd", N, ptr(t['off']), ptr(t['weights']), ptr(t['rgb']), ptr(t['normal']), ptr(t['step_id']), bg, dist,
         ptr(A['rm']), ptr(A['sr']), ptr(A['pre']), ptr(A['pres']), ptr(A['nm']), ptr(A['dep']), st)
    loss_a = torch.zeros((), device=dev)
    scratch = torch.zeros(4096, device=dev)
    call("fgs_fine_loss_fwd", N, M, ptr(A['rm']), ptr(A['sr']), ptr(t['target']), ptr(t['last']), ptr(t['weights']), ptr(t['normal']),
         ptr(t['rgb']), ptr(t['ray_id']), ptr(t['viewdirs']), W5, ptr(loss_a), ptr(scratch), scratch.numel(), None, st)
    G = dict(g_rm=e(N, 3), g_sr=e(N, 3), g_last=e(N), g_normal=e(M, 3), g_raw=e(M, 3))
    call("fgs_fine_loss_bwd", N, M, ptr(A['rm']), ptr(A['sr']), ptr(t['target']), ptr(t['last']), ptr(t['weights']), ptr(t['normal']),
         ptr(t['rgb']), ptr(t['ray_id']), ptr(t['viewdirs']), W5, ptr(seed), ptr(G['g_rm']), ptr(G['g_sr']), ptr(G['g_last']),
         ptr(G['g_normal']), ptr(G['g_raw']) if w5[1] > 0 else None, None, st)
    d_out_a, d_w_a = e(M, 3), e(M)
    call("fgs_composite_bwd", M, ptr(t['ray_id']), ptr(t['weights']), ptr(t['rgb']), ptr(A['pre']), ptr(A['pres']), ptr(G['g_rm']),
         ptr(G['g_sr']), ptr(G['g_raw']) if w5[1] > 0 else None, None, bg, ptr(d_out_a), ptr(d_w_a), None, st)
    # ---- the one launch
    B = dict(rm=e(N, 3), sr=e(N, 3), pre=e(N, 3), pres=e(N, 3), nm=e(N, 3), dep=e(N))
    loss_b = torch.zeros((), device=dev)
    scratch_b = torch.zeros(4096, device=dev)
    d_out_b, d_w_b, gn_b, gl_b, grm_b = e(M, 3), e(M), e(M, 3), e(N), e(N, 3)
    call("fgs_fine_render_loss", N, M, ptr(t['off']), ptr(t['weights']), ptr(t['rgb']), ptr(t['normal']), ptr(t['step_id']), bg, dist,
         ptr(t['viewdirs']), ptr(t['target']), ptr(t['last']), W5, ptr(seed), ptr(B['rm']), ptr(B['sr']), ptr(B['pre']),
         ptr(B['pres']), ptr(B['nm']), ptr(B['dep']), ptr(loss_b), ptr(scratch_b), scratch_b.numel(), ptr(d_out_b), ptr(d_w_b),
         ptr(gn_b), ptr(gl_b), ptr(grm_b), None, st)
    torch.cuda.synchronize()
    for k in A:
        assert torch.equal(A[k], B[k]), k
    assert bool(((A['pre'] < 0) | (A['pre'] > 1)).any())                    # the clamp gate is exercised
    assert torch.equal(d_out_a, d_out_b) and torch.equal(d_w_a, d_w_b)
    assert torch.equal(G['g_normal'], gn_b) and torch.equal(G['g_last'], gl_b) and torch.equal(G['g_rm'], grm_b)
    assert bool((gn_b != 0).any()) or w5[3] == 0
    assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_a))
    assert float(scratch_b[0]) == 0.0                                        # the arrival counter is left as found


def test_training_step_with_announced_loss_matches_the_separate_launches(dev):
    """fused.set_loss_spec: the fine-stage forward runs the one launch, fused_render_losses hands its scalar out and the backward
    pass starts from its stash.  Against the same step through the separate launches: the loss to 1e-6 (summation order), the MLP
    and grid gradients to the run-to-run level of their atomics (the per-survivor inputs of everything downstream are
    bit-identical, test above).  A mismatching target falls back to the separate launches."""
    from conftest import rel_l2
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.fused import set_loss_spec
    from fgs_nerf_amd.losses import fused_render_losses, register_unit_seed
    from fgs_nerf_amd.nerf import mlp_layers
    n_rays = 1024
    model = synth.build_model(160, synth.FINE_MODEL, device=dev)
    ro, rd, vd = (t[:n_rays].contiguous().to(dev) for t in synth.random_rays(4096, seed=synth.SEED))
    target = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(12)).to(dev)
    lossw = dict(synth.FINE_LOSS, weight_rgbper=0.05)
    params = [model.sdf.grid, model.k0.grid] + [p for net in (model.rgbnet, model.refnet) for L in mlp_layers(net) for p in (L.weight, L.bias)]
    seed = register_unit_seed(torch.ones((), device=dev))

    def step(announce, tgt=target, unit=True):
        for p in params:
            p.grad = None
        set_loss_spec(model, target if announce else None, lossw)
        res = model(ro, rd, vd, global_step=1000, **synth.RENDER_KWARGS)
        used = res.get('_fused_loss') is not None
        loss = fused_render_losses(res, tgt, lossw, model)
        taken = used and res.get('_fused_loss')['used']
        loss.backward(seed if unit else torch.full((), 0.5, device=dev))
        set_loss_spec(model, None, lossw)
        return float(loss), [p.grad.detach().clone() for p in params], used, taken

    la, ga, used_a, taken_a = step(False)
    lb, gb, used_b, taken_b = step(True)
    assert not used_a and used_b and taken_b
    assert abs(la - lb) <= 1e-6 * abs(la)
    for a, b, p in zip(ga, gb, params):
        assert rel_l2(b, a) < 2e-6, (tuple(p.shape), rel_l2(b, a))
    # a seed that is not known to be 1: the stash is scaled
    lc, gc, _, taken_c = step(True, unit=False)
    assert taken_c
    for a, c, p in zip(ga, gc, params):
        assert rel_l2(2.0 * c, a) < 2e-6, tuple(p.shape)
    # another target than announced: the ordinary path, results of THAT target
    other = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(13)).to(dev)
    ld, gd, used_d, taken_d = step(True, tgt=other)
    le, ge, _, _ = step(False, tgt=other)
    assert used_d and not taken_d
    assert abs(ld - le) <= 1e-6 * abs(le)
    for d, e in zip(gd, ge):
        assert rel_l2(d, e) < 2e-6


def test_coarse_step_with_announced_loss_matches_the_separate_launches(dev):
    """The same hand-over in the coarse stage's fused forward / backward pass (fused_coarse.py through fused_common._composite)."""
    from conftest import rel_l2
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.fused import set_loss_spec
    from fgs_nerf_amd.losses import fused_render_losses, register_unit_seed
    n_rays = 1024
    model = synth.build_model(96, synth.COARSE_MODEL, device=dev)
    ro, rd, vd = (t[:n_rays].contiguous().to(dev) for t in synth.random_rays(4096, seed=synth.SEED))
    target = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(12)).to(dev)
    lossw = dict(synth.COARSE_LOSS, weight_rgbper=0.2, weight_orientation=1e-4)
    params = [p for p in model.parameters() if p.requires_grad]
    seed = register_unit_seed(torch.ones((), device=dev))

    def step(announce):
        for p in params:
            p.grad = None
        set_loss_spec(model, target if announce else None, lossw)
        res = model(ro, rd, vd, global_step=700, **synth.RENDER_KWARGS)
        loss = fused_render_losses(res, target, lossw, model)
        taken = res.get('_fused_loss') is not None and res.get('_fused_loss')['used']
        loss.backward(seed)
        set_loss_spec(model, None, lossw)
        return float(loss.detach()), [None if p.grad is None else p.grad.detach().clone() for p in params], taken

    la, ga, taken_a = step(False)
    lb, gb, taken_b = step(True)
    assert taken_b and not taken_a
    assert abs(la - lb) <= 1e-6 * abs(la)
    n_cmp = 0
    for a, b, p in zip(ga, gb, params):
        assert (a is None) == (b is None)
        if a is not None and float(a.norm()) > 0:
            assert rel_l2(b, a) < 2e-6, (tuple(p.shape), rel_l2(b, a))
            n_cmp += 1
    assert n_cmp >= 4


def test_fixed_order_scalars_are_bit_reproducible_over_many_launches(dev):
    """The "last workgroup to arrive sums the partials" scalars (fgs_common.h fgs_arrive_is_last: agent-scope stores, a wait, the
    arrival; no agent-scope fence) under repetition: 1500 launches of each kernel on the same inputs, with unrelated traffic on
    the device in between, must give ONE bit pattern -- a partial that was not yet visible to the last workgroup, or a stale
    copy of it, would show as a different sum."""
    import ctypes
    from fgs_nerf_amd import dense
    from fgs_nerf_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(9)
    N = 4096
    counts = torch.randint(0, 30, (N,), generator=g)
    off = torch.zeros(N + 1, dtype=torch.int64)
    off[1:] = counts.cumsum(0)
    M = int(off[-1])
    t = dict(off=off, weights=torch.rand(M, generator=g) * 0.2, rgb=torch.rand(M, 3, generator=g),
             normal=torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1), step_id=torch.randint(0, 500, (M,), generator=g),
             viewdirs=torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1), target=torch.rand(N, 3, generator=g),
             last=torch.rand(N, generator=g))
    t = {k: v.to(dev).contiguous() for k, v in t.items()}
    W5 = (ctypes.c_float * 5)(1.0, 0.05, 0.001, 1e-4, 0.02)
    o = {k: torch.empty(N, 3, device=dev) for k in ('rm', 'sr', 'pre', 'pres', 'nm')}
    dep, gl, grm = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
    d_out, d_w, gn = torch.empty(M, 3, device=dev), torch.empty(M, device=dev), torch.empty(M, 3, device=dev)
    scratch = torch.zeros(4096, device=dev)
    reps = 1500
    losses = torch.empty(reps, device=dev)
    noise = torch.randn(1 << 22, device=dev)
    vol = (torch.rand(1, 1, 96, 96, 96, device=dev) + 0.1).requires_grad_(False)
    tv = torch.empty(reps, device=dev)
    for i in range(reps):
        call("fgs_fine_render_loss", N, M, ptr(t['off']), ptr(t['weights']), ptr(t['rgb']), ptr(t['normal']), ptr(t['step_id']), 1.0,
             0.00625, ptr(t['viewdirs']), ptr(t['target']), ptr(t['last']), W5, None, ptr(o['rm']), ptr(o['sr']), ptr(o['pre']),
             ptr(o['pres']), ptr(o['nm']), ptr(dep), ptr(losses[i:]), ptr(scratch), scratch.numel(), ptr(d_out), ptr(d_w), ptr(gn),
             ptr(gl), ptr(grm), None, stream())
        tv[i] = dense.grid_tv_loss(vol, None, scale=0.3).detach()
        if i % 3 == 0:
            noise.mul_(1.0001)                       # dirty lines in every L2 between the launches
    torch.cuda.synchronize()
    assert bool((losses.view(torch.int32) == losses.view(torch.int32)[0]).all()), losses.unique()
    assert bool((tv.view(torch.int32) == tv.view(torch.int32)[0]).all()), tv.unique()
    assert float(scratch[0]) == 0.0


def test_another_term_on_the_render_result_is_refused(dev):
    """With an announced loss the backward pass starts from the kernel's stash: a second term differentiated through the same
    render result (here: on the weights) would be dropped silently -- it must raise instead."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.fused import set_loss_spec
    from fgs_nerf_amd.losses import fused_render_losses
    model = synth.build_model(96, synth.FINE_MODEL, device=dev)
    ro, rd, vd = (t[:512].contiguous().to(dev) for t in synth.random_rays(4096, seed=synth.SEED))
    target = torch.rand(512, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    set_loss_spec(model, target, synth.FINE_LOSS)
    res = model(ro, rd, vd, global_step=1000, **synth.RENDER_KWARGS)
    loss = fused_render_losses(res, target, synth.FINE_LOSS, model) + 0.1 * res['weights'].sum()
    with pytest.raises(RuntimeError, match="another term"):
        loss.backward()
    set_loss_spec(model, None, synth.FINE_LOSS)
