"""The captured (hipGraph) training iteration against the eager fused step: same batches, same schedule.
The render is deterministic and the survivor counts must agree exactly; the loss scalar is an atomic sum over blocks and the
gradients go through fp32 atomics in both forms (order dependent), so losses are compared to an ulp and parameters in norm
after a few Adam steps.  Also: the survivor count never reaches the
host, an overflowing capacity is flagged and leaves the parameters untouched."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, G=48, N=512):
    import bench
    from fgs_nerf_amd import synth
    model = synth.build_model(G, synth.FINE_MODEL, device=dev)
    opt = bench.make_optimizer(model)
    batches = []
    for b in range(3):
        ro, rd, vd = synth.random_rays(N, seed=50 + b)
        target = torch.rand(N, 3, generator=torch.Generator().manual_seed(b))
        batches.append(tuple(t.to(dev).contiguous() for t in (ro, rd, vd, target)))
    return model, opt, batches


def _lr_schedule(it, group):
    return group['lr'] * (0.9 ** it)           # a schedule that really changes every iteration


def test_captured_step_matches_eager_steps(dev):
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.graph_step import CapturedFineStep
    from fgs_nerf_amd.losses import fused_render_losses
    N, ITERS, TV = 512, 4, (0.01 * 0.1 / 512, True)
    gsteps = [1000, 1500, 2500, 4000]
    # ---- eager reference: host scalars, survivor count read by the host every step
    model, opt, batches = _setup(dev)
    base = [g['lr'] for g in opt.param_groups]
    losses_e, surv_e = [], 0
    for it in range(ITERS):
        for g, b in zip(opt.param_groups, base):
            g['lr'] = _lr_schedule(it, dict(g, lr=b))
        ro, rd, vd, target = batches[it % 3]
        res = model(ro, rd, vd, global_step=gsteps[it], **synth.RENDER_KWARGS)
        surv_e += int(res['weights'].shape[0])
        loss = fused_render_losses(res, target, synth.FINE_LOSS, model)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        model.sdf_total_variation_add_grad(*TV)
        opt.step()
        losses_e.append(float(loss))
    # ---- captured
    model2, opt2, _ = _setup(dev)
    base2 = {id(g): g['lr'] for g in opt2.param_groups}
    step = CapturedFineStep(model2, opt2, synth.FINE_LOSS, synth.RENDER_KWARGS, N, n_iters=ITERS,
                            global_step_of=lambda it: gsteps[it], lr_of=lambda it, g: _lr_schedule(it, dict(g, lr=base2[id(g)])),
                            tv=TV, capacity=8192)
    step.capture(batches[0])
    p0 = model2.k0.grid.detach().clone()
    losses_g = []
    for it in range(ITERS):
        losses_g.append(step.replay(batches[it % 3]).clone())
    overflow, surv_g = step.check()
    losses_g = [float(x) for x in losses_g]
    assert not overflow and surv_g == surv_e, (overflow, surv_g, surv_e)
    # same survivors, same kernels: the first loss differs only by the order of the loss kernel's atomic block sums
    assert abs(losses_g[0] - losses_e[0]) < 2e-7 * abs(losses_e[0]) + 1e-9, (losses_g, losses_e)
    for a, b in zip(losses_g, losses_e):
        assert abs(a - b) < 2e-4 * abs(b), (losses_g, losses_e)
    assert not torch.equal(model2.k0.grid, p0)
    # element by element (conftest.adam_drift_report): both runs are the same kernels, apart only in the order of float atomics
    from conftest import adam_drift_report
    lr_of = {id(p): _lr_schedule(0, dict(g, lr=b)) for g, b in zip(opt.param_groups, base) for p in g['params']}
    print()
    for (name, pa), pb in zip(model.named_parameters(), model2.parameters()):
        if pa not in opt.state:
            continue
        # (how many elements differ beyond rounding varies from run to run with the order of the float atomics -- up to 6.6e-2 of a
        #  256-element bias seen -- and so does the worst element: a weight whose gradient is at the noise level took 0.22 of an
        #  update in one of five runs of this test on one box.  Bounded: at most 2e-3 of a tensor beyond a tenth of an update, no
        #  element a whole update apart after the four steps; the norm-level statement below is the tight one.)
        adam_drift_report(name, pb, pa, lr_of[id(pa)], ITERS, tight=1e-4, max_frac_tight=1.0, max_frac_tenth=2e-3, max_frac_lr=0.0,
                          max_worst_lr=1.0)
        da = float((pa.detach() - pb.detach()).norm() / pa.detach().norm().clamp_min(1e-30))
        assert da < 1e-4, da
    assert all(st['step'] == ITERS for st in opt2.state.values())


def test_capacity_overflow_is_flagged_and_skips_the_update(dev):
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.graph_step import CapturedFineStep
    model, opt, batches = _setup(dev)
    step = CapturedFineStep(model, opt, synth.FINE_LOSS, synth.RENDER_KWARGS, 512, n_iters=4,
                            global_step_of=lambda it: 1000, lr_of=lambda it, g: g['lr'], tv=None, capacity=256)
    step.capture(batches[0])
    before = [p.detach().clone() for p in model.parameters()]
    step.replay(batches[1])
    overflow, _ = step.check()
    assert overflow                                                           # 512 rays keep far more than 256 samples
    for p, q in zip(model.parameters(), before):
        assert torch.equal(p.detach(), q)                                     # the optimizer kernels skipped the step


def test_stepper_captured_window_matches_step_by_step(dev):
    """TrainStepper.run_captured (a window of the reference's loop body as hipGraph replays: learning-rate decay, s_val
    schedule and the sdf TV add-grad as rows of the device table) against TrainStepper.step() for the same iterations, on
    twin models with the same batch sequence."""
    from fgs_nerf_amd import nerf_training as nt
    from fgs_nerf_amd import synth
    cfg = dict(N_iters=20000, N_rand=512, lrate_k0=0.1, lrate_sdf=0.005, lrate_rgbnet=1e-3, lrate_refnet=1e-3, lrate_decay=20,
               ray_sampler='flatten', weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.0, weight_tv_density=0.01,
               weight_tv_k0=0.0, sigmoid_rgb_loss=0.02, weight_orientation=1e-4, tv_every=1, tv_from=0, tv_end=30000,
               voxel_inc=False, pg_scale=[], reset_iter=[], tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.0),
               tv_dense_before=20000, cosine_lr=True,
               cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0), decay_step_module={},
               skip_zero_grad_fields=['density', 'k0', 'k1'])
    R, FIRST, N = 2048, 900, 5
    rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=31))
    target = torch.rand(R, 3, generator=torch.Generator().manual_seed(7)).to(dev)
    runs = {}
    for mode in ("steps", "captured"):
        model = synth.build_model(48, synth.FINE_MODEL, device=dev)
        st = nt.TrainStepper(model, cfg, {}, synth.RENDER_KWARGS, target, *rays, stage='fine', seed=13)
        if mode == "steps":
            losses = torch.stack([st.step(g).detach() for g in range(FIRST, FIRST + N)])
        else:
            losses, overflow = st.run_captured(FIRST, N)
            assert not overflow
        torch.cuda.synchronize()
        runs[mode] = (losses.cpu(), [p.detach().clone() for p in model.parameters()],
                      [g['lr'] for g in st.optimizer.param_groups], [s['step'] for s in st.optimizer.state.values()])
    la, lb = runs["steps"][0], runs["captured"][0]
    assert abs(float(la[0]) - float(lb[0])) < 2e-6 * abs(float(la[0])), (la, lb)          # same parameters, same batch
    assert float(((la - lb) / la).abs().max()) < 5e-4, (la, lb)
    # (five Adam steps of two runs whose gradients are summed with float atomics: an entry whose tiny gradient flips sign moves
    # by 2 lr; the one-step agreement of the early k0 update is pinned in test_inline_early_k0_update_matches_the_update_in_step)
    for pa, pb in zip(runs["steps"][1], runs["captured"][1]):
        assert float((pa - pb).norm() / pa.norm().clamp_min(1e-30)) < 5e-3
    for a, b in zip(runs["steps"][2], runs["captured"][2]):
        assert abs(a - b) <= 1e-12 * abs(a)                                                  # the host's lr copy caught up
    assert runs["steps"][3] == runs["captured"][3] and set(runs["captured"][3]) == {N}
    # a captured window after eager steps on the SAME stepper (their autograd leftovers must not reach the capture)
    model = synth.build_model(32, synth.FINE_MODEL, device=dev)
    st = nt.TrainStepper(model, cfg, {}, synth.RENDER_KWARGS, target, *rays, stage='fine', seed=2)
    for g in range(1, 3):
        st.step(g)
    losses, overflow = st.run_captured(3, 3)
    assert not overflow and bool(torch.isfinite(losses).all())
    # windows the captured form does not cover are refused, not silently mis-run
    model = synth.build_model(32, synth.FINE_MODEL, device=dev)
    st = nt.TrainStepper(model, dict(cfg, weight_tv_k0=0.1), {}, synth.RENDER_KWARGS, target, *rays, stage='fine', seed=1)
    with pytest.raises(RuntimeError, match="k0"):
        st.run_captured(1, 6)


def test_captured_window_across_a_pg_scale_boundary_matches_step_by_step(dev):
    """A progressive-growing iteration inside the window (model/nerf_training.py:244-253: new grids, new optimizer): run_captured
    cuts the window there, rescales exactly as step() does at the head of that iteration and captures the rest anew -- same
    losses and parameters as step() iteration by iteration, the grid grown, the optimizer restarted."""
    from fgs_nerf_amd import nerf_training as nt
    from fgs_nerf_amd import synth
    cfg = dict(N_iters=20000, N_rand=512, lrate_k0=0.1, lrate_sdf=0.005, lrate_rgbnet=1e-3, lrate_refnet=1e-3, lrate_decay=20,
               ray_sampler='flatten', weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.0, weight_tv_density=0.01,
               weight_tv_k0=0.0, sigmoid_rgb_loss=0.02, weight_orientation=1e-4, tv_every=1, tv_from=0, tv_end=30000,
               voxel_inc=False, pg_scale=[903], scale_ratio=2.0, reset_iter=[], tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.0),
               tv_dense_before=20000, cosine_lr=True,
               cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0), decay_step_module={},
               skip_zero_grad_fields=['density', 'k0', 'k1'])
    R, FIRST, N = 2048, 900, 6
    rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=31))
    target = torch.rand(R, 3, generator=torch.Generator().manual_seed(7)).to(dev)
    runs = {}
    for mode in ("steps", "captured"):
        model = synth.build_model(32, synth.FINE_MODEL, device=dev)
        st = nt.TrainStepper(model, cfg, {}, synth.RENDER_KWARGS, target, *rays, stage='fine', seed=13)
        if mode == "steps":
            losses = torch.stack([st.step(g).detach() for g in range(FIRST, FIRST + N)])
        else:
            losses, overflow = st.run_captured(FIRST, N)
            assert not overflow
        torch.cuda.synchronize()
        runs[mode] = (losses.cpu(), [p.detach().clone() for p in model.parameters()], tuple(model.sdf.grid.shape),
                      sorted({s['step'] for s in st.optimizer.state.values()}))
    assert runs["steps"][2] == runs["captured"][2] and runs["captured"][2][2] > 32            # the grid grew, identically
    assert runs["steps"][3] == runs["captured"][3] == [N - 3]                                # optimizer restarted at the cut
    la, lb = runs["steps"][0], runs["captured"][0]
    assert la.shape == lb.shape == (N,)
    assert float(((la - lb) / la).abs().max()) < 1e-3, (la, lb)
    # (Adam restarts at the cut: in its first steps an entry moves by ~lr whatever the size of its gradient, so an entry whose
    # tiny atomically-summed gradient comes out with the other sign differs by 2 lr per step -- a few of the 256 entries of a bias
    # vector (|b| ~ 0.03) are enough for 6e-3 in norm; bounded entry-wise by that mechanism, in norm by 2e-2 / 5e-3)
    for pa, pb in zip(runs["steps"][1], runs["captured"][1]):
        assert pa.shape == pb.shape
        assert float((pa - pb).abs().max()) <= 2 * 0.1 * (N - 3) + 1e-6
        assert float((pa - pb).norm() / pa.norm().clamp_min(1e-30)) < (2e-2 if pa.dim() == 1 else 5e-3)


def test_stepper_captured_window_with_the_shipped_tv_schedule(dev):
    """The shipped fine-stage schedule (config/shiny_blender.py: tv_every = 3, sdf TV add-grad + the autograd smooth-gradient TV
    term in the iterations where it is active) as a captured window: two graphs over the same state, chosen per iteration
    -- against TrainStepper.step() for the same iterations on a twin model."""
    from fgs_nerf_amd import nerf_training as nt
    from fgs_nerf_amd import synth
    cfg = dict(N_iters=20000, N_rand=512, lrate_k0=0.1, lrate_sdf=0.005, lrate_rgbnet=1e-3, lrate_refnet=1e-3, lrate_decay=20,
               ray_sampler='flatten', weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.0, weight_tv_density=0.01,
               weight_tv_k0=0.0, sigmoid_rgb_loss=0.02, weight_orientation=1e-4, tv_every=3, tv_from=0, tv_end=30000,
               voxel_inc=False, pg_scale=[], reset_iter=[], tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.05),
               tv_dense_before=20000, cosine_lr=True,
               cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0), decay_step_module={},
               skip_zero_grad_fields=['density', 'k0', 'k1'])
    R, FIRST, N = 2048, 898, 7                       # iterations 898 .. 904: the TV schedule is active in 900 and 903
    rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=31))
    target = torch.rand(R, 3, generator=torch.Generator().manual_seed(7)).to(dev)
    runs = {}
    for mode in ("steps", "captured"):
        model = synth.build_model(48, synth.FINE_MODEL, device=dev)
        st = nt.TrainStepper(model, cfg, {}, synth.RENDER_KWARGS, target, *rays, stage='fine', seed=13)
        if mode == "steps":
            losses = torch.stack([st.step(g).detach() for g in range(FIRST, FIRST + N)])
        else:
            losses, overflow = st.run_captured(FIRST, N)
            assert not overflow
        torch.cuda.synchronize()
        runs[mode] = (losses.cpu(), [p.detach().clone() for p in model.parameters()],
                      [s['step'] for s in st.optimizer.state.values()])
    la, lb = runs["steps"][0], runs["captured"][0]
    assert float(((la - lb) / la).abs().max()) < 5e-4, (la, lb)
    # the TV iterations carry the extra loss term: their losses stand out from the plain ones in BOTH runs the same way
    assert float((la - lb).abs()[[2, 5]].max()) < 5e-4 * float(la.abs().max())
    for pa, pb in zip(runs["steps"][1], runs["captured"][1]):
        assert float((pa - pb).norm() / pa.norm().clamp_min(1e-30)) < 2e-3
    assert runs["steps"][2] == runs["captured"][2] and set(runs["captured"][2]) == {N}


def test_captured_coarse_step_matches_eager_steps(dev):
    """The coarse stage (5^3 smoothing + gradient volume + forward_coarse + losses + backward + TV + MaskedAdam) captured the
    same way: same survivor totals, first loss equal to an ulp, parameters equal in norm after a few Adam steps."""
    import bench
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.graph_step import CapturedStep
    from fgs_nerf_amd.losses import fused_render_losses
    N, ITERS, TV = 512, 4, (0.01 * 0.1 / 512, True)

    def setup():
        model = synth.build_model(48, synth.COARSE_MODEL, device=dev)
        opt = bench.make_optimizer(model)
        batches = []
        for b in range(3):
            ro, rd, vd = synth.random_rays(N, seed=70 + b)
            target = torch.rand(N, 3, generator=torch.Generator().manual_seed(b))
            batches.append(tuple(t.to(dev).contiguous() for t in (ro, rd, vd, target)))
        return model, opt, batches

    model, opt, batches = setup()
    losses_e, surv_e = [], 0
    for it in range(ITERS):
        ro, rd, vd, target = batches[it % 3]
        res = model(ro, rd, vd, global_step=300 + 50 * it, **synth.RENDER_KWARGS)
        surv_e += int(res['weights'].shape[0])
        loss = fused_render_losses(res, target, synth.COARSE_LOSS, model)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        model.sdf_total_variation_add_grad(*TV)
        opt.step()
        losses_e.append(float(loss.detach()))
    model2, opt2, _ = setup()
    step = CapturedStep(model2, opt2, synth.COARSE_LOSS, synth.RENDER_KWARGS, N, n_iters=ITERS,
                        global_step_of=lambda it: 300 + 50 * it, lr_of=lambda it, g: g['lr'], tv=TV, capacity=16384)
    step.capture(batches[0])
    losses_g = [float(step.replay(batches[it % 3]).clone()) for it in range(ITERS)]
    overflow, surv_g = step.check()
    assert not overflow and surv_g == surv_e, (overflow, surv_g, surv_e)
    assert abs(losses_g[0] - losses_e[0]) < 2e-7 * abs(losses_e[0]) + 1e-9, (losses_g, losses_e)
    for a, b in zip(losses_g, losses_e):
        assert abs(a - b) < 5e-4 * abs(b), (losses_g, losses_e)
    for pa, pb in zip(model.parameters(), model2.parameters()):
        assert float((pa.detach() - pb.detach()).norm() / pa.detach().norm().clamp_min(1e-30)) < 3e-3


@pytest.mark.parametrize("n,cap", [(1, 10), (4096, 10 ** 9), (4096, 30000), (5000, 0), (70001, 123456)])
def test_scan_with_capacity_guard_equals_scan_then_guard(dev, n, cap):
    """fgs_exclusive_scan_guard_i64 (one launch) == fgs_exclusive_scan_i64 followed by fgs_count_guard, bit for bit: offsets
    cut at the capacity, flags[1] = overflow of this call, flags[0] sticky, total += min(count, capacity)."""
    from fgs_nerf_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(n)
    counts = torch.randint(0, 30, (n,), generator=g, dtype=torch.int64).to(dev)
    a, b = torch.empty(n + 1, dtype=torch.int64, device=dev), torch.empty(n + 1, dtype=torch.int64, device=dev)
    fa = torch.tensor([0, 7], dtype=torch.int32, device=dev)
    fb = fa.clone()
    ta, tb = torch.tensor([11], dtype=torch.int64, device=dev), torch.tensor([11], dtype=torch.int64, device=dev)
    for _ in range(2):                                   # twice: the sticky flag and the running total accumulate
        call("fgs_exclusive_scan_i64", ptr(counts), n, ptr(a), stream())
        call("fgs_count_guard", ptr(a), n + 1, cap, ptr(fa), ptr(ta), stream())
        call("fgs_exclusive_scan_guard_i64", ptr(counts), n, ptr(b), cap, ptr(fb), ptr(tb), stream())
        assert torch.equal(a, b) and torch.equal(fa, fb) and torch.equal(ta, tb)
    total = int(counts.sum())
    assert int(fb[1]) == int(total > cap) and int(b[-1]) == min(total, cap)


@pytest.mark.parametrize("captured", [False, True])
def test_inline_early_k0_update_matches_the_update_in_step(dev, captured):
    """fused.enable_early_update(inline=True) -- k0's Adam pass issued from inside the backward pass, right behind the
    feature-grid scatter, beside the weight-gradient launch -- against the same iterations with every update in opt.step():
    eager (host schedule) and captured (device schedule: step size and skip flag read from device memory).  One step from the
    same state: k0 and its moments agree to the float-atomics order of k0.grad; several steps: parameters in norm; the optimizer's
    step counters advance once per iteration either way."""
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.graph_step import CapturedFineStep
    from fgs_nerf_amd.losses import fused_render_losses
    N, ITERS, TV = 512, 3, (0.01 * 0.1 / 512, True)
    runs = {}
    for early in (False, True):
        model, opt, batches = _setup(dev)
        if early:
            fused.enable_early_update(model, opt, None, inline=True)
        if captured:
            base = {id(g): g['lr'] for g in opt.param_groups}
            step = CapturedFineStep(model, opt, synth.FINE_LOSS, synth.RENDER_KWARGS, N, n_iters=ITERS,
                                    global_step_of=lambda it: 1000 + it, lr_of=lambda it, g: base[id(g)], tv=TV, capacity=8192)
            trained = [p for g in opt.param_groups for p in g['params']]     # (s_val is a schedule mirror the tick kernel writes)
            before = [p.detach().clone() for p in trained]
            step.capture(batches[1])                    # (a different batch than the first replay's)
            torch.cuda.synchronize()
            for a, b in zip(before, trained):           # the warm-up pass of capture() applies NO update, early or not
                assert torch.equal(a, b.detach())
            snaps, prev = [], model.k0.grid.detach().clone()
            for it in range(ITERS):
                step.replay(batches[it % 3])
                torch.cuda.synchronize()
                assert not torch.equal(prev, model.k0.grid.detach())     # ... and every replay does update k0
                prev = model.k0.grid.detach().clone()
                if it == 0:
                    snaps = [model.k0.grid.detach().clone(), opt.state[model.k0.grid]['exp_avg'].clone()]
            assert not step.check()[0]
        else:
            snaps = []
            for it in range(ITERS):
                ro, rd, vd, target = batches[it % 3]
                res = model(ro, rd, vd, global_step=1000 + it, **synth.RENDER_KWARGS)
                loss = fused_render_losses(res, target, synth.FINE_LOSS, model)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                model.sdf_total_variation_add_grad(*TV)
                opt.step()
                if it == 0:
                    snaps = [model.k0.grid.detach().clone(), opt.state[model.k0.grid]['exp_avg'].clone()]
        torch.cuda.synchronize()
        runs[early] = (snaps, [p.detach().clone() for p in model.parameters()], [st['step'] for st in opt.state.values()])
        fused.disable_early_update(model)
    for a, b in zip(runs[False][0], runs[True][0]):          # after ONE step from the same state
        assert float((a - b).norm() / a.norm().clamp_min(1e-30)) < 2e-5
    # (several steps of two runs whose gradients are summed with float atomics: Adam's first updates are ~lr * sign(g), so a
    # near-zero bias gradient that flips sign moves that entry by 2 lr -- 2.3e-3 of a bias vector's norm has been seen)
    for pa, pb in zip(runs[False][1], runs[True][1]):
        assert float((pa - pb).norm() / pa.norm().clamp_min(1e-30)) < 5e-3
    assert runs[False][2] == runs[True][2] and set(runs[True][2]) == {ITERS}


def test_captured_coarse_window_with_ori_tv_and_table_updates_matches_step_by_step(dev):
    """The shipped COARSE-stage loop body (config/shiny_blender.py:105-146): `ori_tv` autograd TV terms every iteration, a
    `tv_updates` entry, a `decay_step_module` entry and a `pg_scale` rescale inside the window.  run_captured cuts the window at
    each of them, applies what step() applies, and captures the rest anew: same losses, parameters, learning rates and TV terms
    as step() iteration by iteration."""
    from fgs_nerf_amd import nerf_training as nt
    from fgs_nerf_amd import synth
    cfg = dict(N_iters=15000, N_rand=512, lrate_k0=0.1, lrate_sdf=0.005, lrate_refnet=1e-3, lrate_decay=20,
               ray_sampler='flatten', weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.2, weight_tv_density=0.01,
               weight_tv_k0=0.0, sigmoid_rgb_loss=0.1, weight_orientation=1e-4, tv_every=1, tv_from=0, tv_end=40000,
               voxel_inc=False, pg_scale=[904], scale_ratio=2.0, reset_iter=[], ori_tv=True,
               tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.05), tv_updates={901: dict(sdf_tv=0.2)},
               decay_step_module={902: dict(sdf=0.5, k0=0.3)}, tv_dense_before=20000, cosine_lr=True,
               cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0),
               skip_zero_grad_fields=['density', 'k0', 'k1'])
    R, FIRST, N = 2048, 900, 7
    rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=41))
    target = torch.rand(R, 3, generator=torch.Generator().manual_seed(9)).to(dev)
    runs = {}
    for mode in ("steps", "captured"):
        model = synth.build_model(32, synth.COARSE_MODEL, device=dev)
        st = nt.TrainStepper(model, dict(cfg), {}, synth.RENDER_KWARGS, target, *rays, stage='coarse', seed=17)
        if mode == "steps":
            losses = torch.stack([st.step(g).detach() for g in range(FIRST, FIRST + N)])
        else:
            losses, overflow = st.run_captured(FIRST, N)
            assert not overflow
        torch.cuda.synchronize()
        runs[mode] = (losses.cpu(), [p.detach().clone() for p in model.parameters()], tuple(model.sdf.grid.shape),
                      {g['name']: g['lr'] for g in st.optimizer.param_groups}, dict(st.cfg_train['tv_terms']))
    assert runs["steps"][2] == runs["captured"][2] and runs["captured"][2][2] > 32            # rescaled at 904, identically
    assert runs["steps"][4] == runs["captured"][4] and runs["captured"][4]['sdf_tv'] == 0.2  # the tv_updates entry was applied
    for k, v in runs["steps"][3].items():
        assert abs(v - runs["captured"][3][k]) <= 1e-12 * abs(v), k                          # lr schedule incl. decay_step_module
    la, lb = runs["steps"][0], runs["captured"][0]
    assert la.shape == lb.shape == (N,)
    assert float(((la - lb) / la).abs().max()) < 2e-3, (la, lb)
    for pa, pb in zip(runs["steps"][1], runs["captured"][1]):
        assert pa.shape == pb.shape and float((pa - pb).norm() / pa.norm().clamp_min(1e-30)) < 1e-2


def test_captured_window_in_the_voxel_increment_phase_matches_step_by_step(dev):
    """Coarse stage with `voxel_inc` (config/shiny_blender.py:51-58, model/nerf_training.py:286-291): every iteration up to
    inc_steps renders through a larger increment mask.  The captured iteration rebuilds the mask on the device from index bounds in
    its schedule table (fgs_box_mask_fill): the masks equal step()'s set_inc_mask voxel for voxel, the survivor totals, losses
    and parameters follow; the window runs past inc_steps (the mask then stays what iteration inc_steps left)."""
    from fgs_nerf_amd import nerf_training as nt
    from fgs_nerf_amd import synth
    from fgs_nerf_amd._lib import call, ptr, stream
    cfg = dict(N_iters=15000, N_rand=512, lrate_k0=0.1, lrate_sdf=0.005, lrate_refnet=1e-3, lrate_decay=20,
               ray_sampler='flatten', weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.2, weight_tv_density=0.01,
               weight_tv_k0=0.0, sigmoid_rgb_loss=0.1, weight_orientation=1e-4, tv_every=1, tv_from=0, tv_end=40000,
               voxel_inc=True, inc_steps=8, x_mid=0.5, y_mid=0.45, z_mid=0.55, x_init_ratio=0.3, y_init_ratio=0.4,
               z_init_ratio=0.35, pg_scale=[], scale_ratio=2.0, reset_iter=[], ori_tv=True,
               tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.05), tv_dense_before=20000, cosine_lr=True,
               cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0), decay_step_module={},
               skip_zero_grad_fields=['density', 'k0', 'k1'])
    R, FIRST, N = 2048, 3, 8                       # iterations 3 .. 10: six inside the phase, two behind it
    rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=43))
    target = torch.rand(R, 3, generator=torch.Generator().manual_seed(11)).to(dev)
    runs = {}
    for mode in ("steps", "captured"):
        model = synth.build_model(32, synth.COARSE_MODEL, device=dev)
        st = nt.TrainStepper(model, dict(cfg), {}, synth.RENDER_KWARGS, target, *rays, stage='coarse', seed=19)
        if mode == "steps":
            losses, masks = [], []
            for g in range(FIRST, FIRST + N):
                losses.append(st.step(g).detach())
                masks.append(model.inc_mask.mask.clone())
            losses = torch.stack(losses)
            # the device-side rebuild sets the voxels set_inc_mask sets, iteration by iteration
            for g, m in zip(range(FIRST, FIRST + N), masks):
                w = min(min(g, cfg['inc_steps']) * 1.0 / cfg['inc_steps'], 1.0)
                b = model.inc_index_bounds(st.inc_lower_init - w * st.inc_lower_init, st.inc_upper_init + w * (1 - st.inc_upper_init))
                out = torch.full(m.shape, 7, dtype=torch.uint8, device=dev)
                call("fgs_box_mask_fill", ptr(out), *m.shape, ptr(torch.tensor(b, dtype=torch.float32, device=dev)), stream())
                assert torch.equal(out.bool(), m) and int(out.max()) <= 1, g
            assert int(masks[0].sum()) < int(masks[3].sum()) < int(masks[5].sum()) == int(masks[-1].sum()) == masks[-1].numel()
        else:
            losses, overflow = st.run_captured(FIRST, N)
            assert not overflow
            assert bool(model.inc_mask.mask.all())               # the last iterations' mask: the whole grid
        torch.cuda.synchronize()
        runs[mode] = (losses.cpu(), [p.detach().clone() for p in model.parameters()])
    la, lb = runs["steps"][0], runs["captured"][0]
    assert abs(float(la[0]) - float(lb[0])) < 2e-6 * abs(float(la[0])), (la, lb)          # same mask, same parameters, same batch
    assert float(((la - lb) / la).abs().max()) < 2e-3, (la, lb)
    for pa, pb in zip(runs["steps"][1], runs["captured"][1]):
        assert float((pa - pb).abs().max()) <= 2 * 0.1 * N + 1e-6
        assert float((pa - pb).norm() / pa.norm().clamp_min(1e-30)) < (2e-2 if pa.dim() == 1 else 5e-3)
