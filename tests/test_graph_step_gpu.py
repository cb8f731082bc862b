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
    for pa, pb in zip(model.parameters(), model2.parameters()):
        da = float((pa.detach() - pb.detach()).norm() / pa.detach().norm().clamp_min(1e-30))
        assert da < 2e-3, da
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
