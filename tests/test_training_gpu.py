"""GPU tests of the per-iteration training body (fgs-nerf_amd/nerf_training.py) against an explicit
oracle-side loop: CPU autograd through oracle.forward_fine, the oracle's TV-add-grad and Adam kernels, the reference's
schedule arithmetic written out literally (model/nerf_training.py:300-456)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import adam_drift_report, rel_l2

pytestmark = pytest.mark.gpu

TRAIN = dict(N_iters=20, N_rand=256, lrate_k0=0.1, lrate_sdf=0.005, lrate_rgbnet=1e-3, lrate_refnet=1e-3, lrate_decay=20,
             ray_sampler='flatten', weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.05, weight_tv_density=0.01,
             weight_tv_k0=0.0, sigmoid_rgb_loss=0.02, weight_orientation=1e-4, tv_every=3, tv_from=0, tv_end=30000,
             voxel_inc=False, pg_scale=[], reset_iter=[], tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.05),
             tv_dense_before=20000, cosine_lr=True, cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0),
             decay_step_module={2: dict(sdf=0.1)}, skip_zero_grad_fields=['density', 'k0', 'k1'])


# element-wise parameter bars of the multi-step comparisons (conftest.adam_drift_report has the argument): bounds on the fraction
# of elements that differ beyond rounding / by a tenth of an update / by a whole update after 6 steps at 24^3, 256 rays
# (measured, round 4: sdf 8.6e-2 / 6.6e-3 / 0, worst 0.98 lr; k0 7.8e-2 / 5.4e-5 / 0, worst 0.22 lr.  Across a rescale -- two
#  "first Adam steps" phases, and the trilinear resampling spreads every difference over the finer grid: sdf 0.25 / 8.7e-2 / 1.6e-2,
#  worst 5.3 lr; k0 7.7e-2 / 1.1e-4 / 0, worst 0.27 lr.)
DRIFT_SDF = dict(tight=1e-4, max_frac_tight=0.17, max_frac_tenth=1.5e-2, max_frac_lr=1e-3)
DRIFT_K0 = dict(tight=1e-4, max_frac_tight=0.15, max_frac_tenth=5e-4, max_frac_lr=0.0, max_worst_lr=1.0)
DRIFT_SDF_RESCALE = dict(tight=1e-4, max_frac_tight=0.4, max_frac_tenth=0.15, max_frac_lr=3.5e-2)


def test_schedule_arithmetic_matches_reference_formulas():
    from fgs_nerf_amd import nerf_training as nt
    cfg = dict(TRAIN, N_iters=20000)
    # cosine: factor = cos_lr(g-1) / cos_lr(g-2) with the reference's warm-up branch for negative iterations
    def cos(it):
        return 1.0 if it < 0 else (1 + math.cos(it / 20000 * math.pi)) * 0.5
    for g in (1, 2, 3, 777, 19999):
        assert abs(nt.lr_decay_factor(cfg, g) - cos(g - 1) / cos(g - 2)) < 1e-15
    assert abs(nt.lr_decay_factor(dict(cfg, cosine_lr=False), 5) - 0.1 ** (1 / 20000)) < 1e-15


def _oracle_loop(oracle, P, rays_c, target_c, cfg, iters, sampler, tvk, out_state=None):
    """The reference's per-iteration body written out literally on the CPU (model/nerf_training.py:243-253, 300-456):
    progressive growing (trilinear rescale of both grids, model/grid.py:101-106, new optimizer with the base learning
    rates), oracle.forward_fine, the loss terms, the smooth-gradient TV term, TV add-grad, the C restatement of the Adam
    kernels, the schedule arithmetic.  Mutates P; returns (losses, lr)."""
    from fgs_nerf_amd.losses import render_losses
    R = len(target_c)
    base_lr = {'k0': cfg['lrate_k0'], 'sdf': cfg['lrate_sdf'], 'rgbnet': cfg['lrate_rgbnet'], 'refnet': cfg['lrate_refnet']}
    lr = dict(base_lr)
    n_iters = cfg['N_iters']

    def leaves_of():
        names = ['sdf', 'k0'] + [f'rgbnet.{i}.{k}' for i in range(len(P['rgbnet'])) for k in ('w', 'b')] + \
                [f'refnet.{i}.{k}' for i in range(len(P['refnet'])) for k in ('w', 'b')]
        leaves = [P['sdf'], P['k0']] + [t for wb in P['rgbnet'] for t in wb] + [t for wb in P['refnet'] for t in wb]
        return names, leaves
    names, leaves = leaves_of()
    state = [(np.zeros(t.numel(), np.float32), np.zeros(t.numel(), np.float32)) for t in leaves]
    adam_step = 0
    num_voxels = int(P['sdf'].shape[2]) ** 3
    losses = []
    for gs in range(1, iters + 1):
        if gs in cfg.get('pg_scale', []):
            num_voxels = num_voxels * cfg['scale_ratio']
            voxel_size, world = oracle.grid_resolution(P['xyz_min'], P['xyz_max'], num_voxels)
            size = tuple(int(w) for w in world)
            with torch.no_grad():
                P['sdf'] = F.interpolate(P['sdf'].detach(), size=size, mode='trilinear', align_corners=True)
                P['k0'] = F.interpolate(P['k0'].detach(), size=size, mode='trilinear', align_corners=True)
            P['voxel_size'] = voxel_size
            names, leaves = leaves_of()
            state = [(np.zeros(t.numel(), np.float32), np.zeros(t.numel(), np.float32)) for t in leaves]
            adam_step, lr = 0, dict(base_lr)               # create_optimizer_or_freeze_model(..., global_step=0)
        G = int(P['sdf'].shape[2])
        w_tv = float(cfg['weight_tv_density'] * cfg['tv_terms']['sdf_tv'] / R * G / 128)
        sel = sampler().cpu()
        for t in leaves:
            t.requires_grad_(True)
            t.grad = None
        res = oracle.forward_fine(P, rays_c[0][sel], rays_c[1][sel], rays_c[2][sel], global_step=gs, near=2.0, stepsize=0.5, bg=1)
        loss = render_losses(res, target_c[sel], cfg)
        tv_now = gs % cfg['tv_every'] == 0
        if tv_now:
            gv = oracle.neus_sdf_gradient(P['sdf'], P['voxel_size']).permute(1, 0, 2, 3, 4)
            sm = F.conv3d(F.pad(gv, (1,) * 6, mode='replicate'), tvk)
            loss = loss + cfg['weight_tv_density'] * ((sm.detach() - gv) ** 2).mean() * cfg['tv_terms']['smooth_grad_tv']
        loss.backward()
        losses.append(float(loss))
        adam_step += 1
        with torch.no_grad():
            if tv_now:
                g = np.ascontiguousarray(P['sdf'].grad.numpy())
                oracle.K.total_variation_add_grad(np.ascontiguousarray(P['sdf'].detach().numpy()), g, w_tv, w_tv, w_tv, True)
                P['sdf'].grad.copy_(torch.from_numpy(g))
            for t, name, (m, v) in zip(leaves, names, state):
                grp = name.split('.')[0]
                p = np.ascontiguousarray(t.detach().numpy()).reshape(-1)
                oracle.K.adam_upd(p, np.ascontiguousarray(t.grad.numpy()).reshape(-1), m, v, adam_step, 0.9, 0.99, lr[grp], 1e-8,
                                  mode=1 if grp == 'k0' else 0)
                t.detach().copy_(torch.from_numpy(p).view_as(t))
        f = (lambda it: 1.0 if it < 0 else (1 + math.cos(it / n_iters * math.pi)) * 0.5)
        fac = f(gs - 1) / f(gs - 2)
        for k in lr:
            lr[k] *= fac
        for name, mul in cfg.get('decay_step_module', {}).get(gs - 1, {}).items():
            lr[name] *= mul
    if out_state is not None:          # name -> (exp_avg, exp_avg_sq) of the oracle's optimizer, flat
        out_state.update({n: st for n, st in zip(names, state)})
    return losses, lr


def test_stepper_matches_oracle_loop(dev, oracle):
    from fgs_nerf_amd import nerf_training as nt
    from fgs_nerf_amd import synth
    G, R, ITERS = 24, 256, 6
    model = synth.build_model(G, synth.FINE_MODEL, device=dev)
    rays_c = synth.random_rays(R, seed=11)
    target_c = torch.rand(R, 3, generator=torch.Generator().manual_seed(12))
    rays = tuple(r.to(dev) for r in rays_c)
    P = synth.oracle_params(model)                       # snapshot of the initial parameters on the CPU
    stepper = nt.TrainStepper(model, TRAIN, {}, synth.RENDER_KWARGS, target_c.to(dev), *rays, stage='fine', seed=5)
    twin = nt.DeviceBatchSampler(R, R, dev, seed=5)      # the same permutations the stepper will draw
    ostate = {}
    losses_o, lr = _oracle_loop(oracle, P, rays_c, target_c, TRAIN, ITERS, twin, model.tv_smooth_conv.weight.detach().cpu(), ostate)

    # ---- the stepper
    losses_g = [float(stepper.step(gs)) for gs in range(1, ITERS + 1)]
    for a, b in zip(losses_g, losses_o):
        assert abs(a - b) < 2e-4 * abs(b), (losses_g, losses_o)
    lrs = {g['name']: g['lr'] for g in stepper.optimizer.param_groups}
    for k in lr:
        assert abs(lrs[k] - lr[k]) < 1e-12 * max(1.0, lr[k]), (k, lrs[k], lr[k])
    # parameters after 6 Adam steps, element by element (conftest.adam_drift_report): elements with a real gradient tight, the
    # noise-level ones within 2 lr per step; the norms of round 3 stay as a summary line
    print()
    adam_drift_report('sdf', model.sdf.grid, P['sdf'], TRAIN['lrate_sdf'], ITERS, **DRIFT_SDF)
    adam_drift_report('k0', model.k0.grid, P['k0'], TRAIN['lrate_k0'], ITERS, **DRIFT_K0)
    assert rel_l2(model.sdf.grid, P['sdf']) < 2e-3
    assert rel_l2(model.k0.grid, P['k0']) < 5e-2
    st = stepper.stats()
    assert set(st) == {'psnr', 'wmax', 'wsum', 'wnonzero', 's_val'} and st['psnr'] > 0


@pytest.mark.parametrize("forced_averager", [False, True])
def test_stepper_across_a_pg_scale_boundary_matches_oracle_loop(dev, oracle, forced_averager):
    """Progressive growing in the middle of a run (model/nerf_training.py:243-253): at iteration 4 both grids are resampled
    to 2x the voxels, a new optimizer starts from the base learning rates, the fused path's cached geometry / workspaces
    and -- with an averager -- the gradient exchange and the in-backward k0 update are re-bound to the NEW parameters
    (GradAverager.rebind).  Three iterations before and three after, against the literal oracle loop; the forced averager
    runs the whole exchange path in a single-rank RCCL group (an identity that still exercises every re-bound hook)."""
    import os
    import socket
    import torch.distributed as dist
    from fgs_nerf_amd import nerf_training as nt
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.dist import GradAverager
    G, R, ITERS = 24, 256, 6
    cfg = dict(TRAIN, pg_scale=[4], scale_ratio=2.0, decay_step_module={})
    if forced_averager:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        model = synth.build_model(G, synth.FINE_MODEL, device=dev)
        rays_c = synth.random_rays(R, seed=11)
        target_c = torch.rand(R, 3, generator=torch.Generator().manual_seed(12))
        rays = tuple(r.to(dev) for r in rays_c)
        P = synth.oracle_params(model)
        avg = GradAverager(model.parameters(), force=True, sparse_min_numel=1 << 12) if forced_averager else None
        stepper = nt.TrainStepper(model, cfg, {}, synth.RENDER_KWARGS, target_c.to(dev), *rays, stage='fine', seed=5,
                                  averager=avg)
        twin = nt.DeviceBatchSampler(R, R, dev, seed=5)
        ostate = {}
        losses_o, lr = _oracle_loop(oracle, P, rays_c, target_c, cfg, ITERS, twin, model.tv_smooth_conv.weight.detach().cpu(), ostate)
        k0_before = model.k0.grid
        losses_g = [float(stepper.step(gs)) for gs in range(1, ITERS + 1)]
        assert model.k0.grid is not k0_before and tuple(model.k0.grid.shape[2:]) == tuple(P['k0'].shape[2:]) != (G, G, G)
        assert torch.equal(model.voxel_size.cpu(), P['voxel_size'])
        for a, b in zip(losses_g, losses_o):
            assert abs(a - b) < 2e-4 * abs(b), (losses_g, losses_o)
        lrs = {g['name']: g['lr'] for g in stepper.optimizer.param_groups}
        for k in lr:
            assert abs(lrs[k] - lr[k]) < 1e-12 * max(1.0, lr[k]), (k, lrs[k], lr[k])
        assert all(st['step'] == 3 for st in stepper.optimizer.state.values())       # the new optimizer took 3 steps
        # two "first Adam steps" phases (before and after the rescale) instead of one: the ~lr * sign(g) updates of voxels
        # whose gradient is rounding noise accumulate twice (measured 2.8e-3; 1.4e-3 in the run without a rescale)
        print()
        adam_drift_report('sdf', model.sdf.grid, P['sdf'], cfg['lrate_sdf'], ITERS, **DRIFT_SDF_RESCALE)
        adam_drift_report('k0', model.k0.grid, P['k0'], cfg['lrate_k0'], ITERS, **DRIFT_K0)
        assert rel_l2(model.sdf.grid, P['sdf']) < 5e-3
        assert rel_l2(model.k0.grid, P['k0']) < 5e-2
        if forced_averager:       # the exchange follows the NEW grids
            assert any(p is model.k0.grid for p in avg.params) and not any(p is k0_before for p in avg.params)
    finally:
        if forced_averager:
            dist.destroy_process_group()


def test_stepper_fused_and_composed_agree_and_coarse_runs(dev):
    from fgs_nerf_amd import nerf_training as nt
    from fgs_nerf_amd import synth
    R = 512
    rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=3))
    target = torch.rand(R, 3, device=dev)
    out = {}
    for fused in (True, False):
        model = synth.build_model(32, synth.FINE_MODEL, device=dev, fused=fused)
        st = nt.TrainStepper(model, dict(TRAIN, N_rand=R), {}, synth.RENDER_KWARGS, target, *rays, stage='fine', seed=9)
        out[fused] = [float(st.step(g)) for g in range(1, 5)]
    for a, b in zip(out[True], out[False]):
        assert abs(a - b) < 1e-3 * abs(b), out
    # coarse stage with the voxel-increment schedule and the exponential decay
    cfg = dict(TRAIN, N_rand=R, cosine_lr=False, decay_step_module={}, lrate_rgbnet=0, voxel_inc=True, inc_steps=3,
               x_mid=0.5, y_mid=0.5, z_mid=0.5, x_init_ratio=0.6, y_init_ratio=0.6, z_init_ratio=0.6,
               weight_rgbper=0.2, sigmoid_rgb_loss=0.1)
    model = synth.build_model(32, synth.COARSE_MODEL, device=dev)
    st = nt.TrainStepper(model, cfg, {}, synth.RENDER_KWARGS, target, *rays, stage='coarse', seed=9)
    ls = [float(st.step(g)) for g in range(1, 7)]
    assert all(np.isfinite(ls)) and model.inc_mask is not None
    assert abs(st.optimizer.param_groups[0]['lr'] - 0.1 * (0.1 ** (1 / 20000)) ** 6) < 1e-9


@pytest.mark.parametrize("stage", ["coarse", "fine"])
def test_training_reduces_the_loss_on_a_synthetic_scene(dev, stage):
    """End-to-end sanity of the whole update path (render, losses, TV schedule, exchange-free averaging, MaskedAdam, LR
    schedule): a student model trained for 150 iterations on pixels rendered by a differently initialised teacher must
    fit them markedly better than at the start.  Catches sign / scale errors no single-step parity test sees."""
    from fgs_nerf_amd import nerf_training as nt
    from fgs_nerf_amd import synth
    cfg = synth.FINE_MODEL if stage == "fine" else synth.COARSE_MODEL
    torch.manual_seed(0)
    teacher = synth.build_model(48, cfg, device=dev)
    with torch.no_grad():                                   # a smaller, coloured ball
        teacher.sdf.grid += 0.25
        teacher.k0.grid.normal_(0.0, 0.5)
    R = 8192
    rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=17))
    with torch.no_grad():
        target = torch.cat([teacher(*(r[i:i + 2048] for r in rays), global_step=None, **synth.RENDER_KWARGS)['rgb_marched']
                            for i in range(0, R, 2048)])
    student = synth.build_model(48, cfg, device=dev)
    train = dict(TRAIN, N_iters=150, N_rand=2048, decay_step_module={}, weight_rgbper=0.0 if stage == "fine" else 0.2)
    if stage == "coarse":
        train['lrate_rgbnet'] = 0
    st = nt.TrainStepper(student, train, {}, synth.RENDER_KWARGS, target, *rays, stage=stage, seed=3)

    def mse():
        with torch.no_grad():
            out = torch.cat([student(*(r[i:i + 2048] for r in rays), global_step=None, **synth.RENDER_KWARGS)['rgb_marched']
                             for i in range(0, R, 2048)])
        return float(((out - target) ** 2).mean())
    before = mse()
    for g in range(1, 151):
        loss = st.step(g)
    assert bool(torch.isfinite(loss))
    after = mse()
    assert after < 0.5 * before, (before, after)
    assert st.stats()['psnr'] > 0


def test_early_update_of_the_feature_grid_matches_a_plain_step(dev):
    """MaskedAdam.early_update (k0's Adam pass issued from inside the fused backward pass, on a side stream) against the
    same step with the update in optimizer.step(): identical parameters and optimizer state afterwards."""
    import bench
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.losses import fused_render_losses
    rays = tuple(t.to(dev) for t in synth.random_rays(600, seed=3))
    target = torch.rand(600, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    outs = []
    for early in (False, True):
        model = synth.build_model(48, synth.FINE_MODEL, device=dev)
        opt = bench.make_optimizer(model)
        if early:
            fused.enable_early_update(model, opt)
        for step in range(3):
            opt.zero_grad(set_to_none=True)
            res = model(*rays, global_step=1000 + step, **synth.RENDER_KWARGS)
            fused_render_losses(res, target, synth.FINE_LOSS, model).backward()
            opt.step()
        torch.cuda.synchronize()
        st = opt.state[model.k0.grid]
        outs.append((model.k0.grid.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone(), st['step'],
                     model.sdf.grid.detach().clone()))
        assert not opt._early
    # the scatter kernels' float atomics are order dependent: compare in norm (Adam's first steps amplify tiny gradients;
    # exp_avg_sq carries the gradient's relative noise twice: seen up to 1.2e-3 between two identical runs)
    assert outs[0][3] == outs[1][3] == 3
    for a, b in zip(outs[0][:3] + outs[0][4:], outs[1][:3] + outs[1][4:]):
        assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < 3e-3
