"""GPU parity tests of the coarse stages (BASELINE config 3): the dense per-iteration volume
kernels (csrc/dense.hip) and the fused forward_coarse path (csrc/march_coarse.hip + fused._FusedCoarse) against
torch restatements, the committed 16^3 golden outputs and the CPU oracle.

Tolerances: rendered pixels / per-sample outputs <= 1e-5 rel-L2 (north_star); gradients <= 1e-3 rel-L2 vs the oracle
at mid size (fp32 atomics, split-K order), <= 2e-4 on the 16^3 golden scene; kept-sample indices bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import match_survivors, rel_l2

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("shape,k", [((19, 23, 17), 5), ((8, 9, 33), 3), ((5, 4, 6), 7), ((40, 40, 40), 5)])
def test_smooth3d_matches_conv3d(dev, shape, k):
    """fgs_smooth3d_fwd/bwd == nn.Conv3d(replicate padding) with the reference's Gaussian taps (model/nerf.py:260-272)."""
    from fgs_nerf_amd import dense
    gen = torch.Generator().manual_seed(5)
    ax = np.arange(-(k // 2), k // 2 + 1, 1)
    xx, yy, zz = np.meshgrid(ax, ax, ax)
    kern = torch.from_numpy(np.exp(-(xx ** 2 + yy ** 2 + zz ** 2) / (2 * 0.8 ** 2))).float()
    kern = (kern / kern.sum()) * (1 + 0.1 * torch.rand(k, k, k, generator=gen))     # asymmetric: catches tap-order errors
    x = torch.randn(1, 1, *shape, generator=gen)
    gy = torch.randn(1, 1, *shape, generator=gen)
    xr = x.clone().double().requires_grad_(True)
    ref = F.conv3d(F.pad(xr, (k // 2,) * 6, mode='replicate'), kern.double()[None, None])
    ref.backward(gy.double())
    xd = x.to(dev).requires_grad_(True)
    out = dense.smooth3d(xd, kern.to(dev))
    out.backward(gy.to(dev))
    assert rel_l2(out, ref.float()) < 1e-6
    assert rel_l2(xd.grad, xr.grad.float()) < 1e-6


@pytest.mark.parametrize("shape", [(19, 23, 17), (3, 4, 5), (2, 2, 2), (40, 40, 40)])
def test_gradient_volume_matches_reference_slicing(dev, shape):
    """fgs_sdf_gradvol_fwd is bit-exact with the slicing arithmetic of model/nerf.py:485-494; bwd is its adjoint."""
    from fgs_nerf_amd import dense
    gen = torch.Generator().manual_seed(6)
    vs = float(np.float32(2.0 / 37))
    s = torch.randn(1, 1, *shape, generator=gen)
    sr = s.clone().requires_grad_(True)
    g = torch.zeros(1, 3, *shape)
    g = g.clone()
    gx = (sr[:, 0, 2:, :, :] - sr[:, 0, :-2, :, :]) / 2 / vs
    gy = (sr[:, 0, :, 2:, :] - sr[:, 0, :, :-2, :]) / 2 / vs
    gz = (sr[:, 0, :, :, 2:] - sr[:, 0, :, :, :-2]) / 2 / vs
    ref = torch.stack([F.pad(gx, (0, 0, 0, 0, 1, 1)), F.pad(gy, (0, 0, 1, 1, 0, 0)), F.pad(gz, (1, 1, 0, 0, 0, 0))], 1)
    cot = torch.randn(1, 3, *shape, generator=gen)
    ref.backward(cot)
    sd = s.to(dev).requires_grad_(True)
    out = dense.sdf_gradient_volume(sd, vs)
    out.backward(cot.to(dev))
    assert torch.equal(out.cpu(), ref.detach())
    assert rel_l2(sd.grad, sr.grad) < 1e-6


def grads_of(model):
    from fgs_nerf_amd.nerf import mlp_layers
    out = {'sdf': model.sdf.grid.grad, 'k0': model.k0.grid.grad}
    for i, l in enumerate(mlp_layers(model.refnet)):
        out[f'refnet.{i}.weight'], out[f'refnet.{i}.bias'] = l.weight.grad, l.bias.grad
    return out


def test_fused_coarse_is_selected_and_matches_golden(dev, golden):
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.losses import render_losses
    g = golden("e2e_coarse.npz")
    model = synth.build_model(16, synth.COARSE_MODEL, device=dev)
    assert fused.supports_coarse(model)
    rays = (T(g["rays_o"], dev), T(g["rays_d"], dev), T(g["viewdirs"], dev))
    res = model(*rays, global_step=int(g["global_step"]), **synth.RENDER_KWARGS)
    assert isinstance(res, fused.LazyResult)
    assert np.array_equal(res["ray_id"].cpu().numpy(), g["ray_id"])
    for key, tol in (("rgb_marched", 1e-5), ("sigmoid_rgb", 1e-5), ("weights", 1e-5), ("raw_rgb", 1e-5), ("normal", 1e-5),
                     ("alphainv_cum", 1e-6)):
        assert rel_l2(res[key], g[key]) < tol, key
    loss = render_losses(res, T(g["target"], dev), synth.COARSE_LOSS, model)
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    loss.backward()
    for k, v in grads_of(model).items():
        assert rel_l2(v, g["grad_" + k]) < 2e-4, k


@pytest.mark.parametrize("G,N,stage,extra", [
    (48, 512, "coarse", {}),
    (40, 300, "coarse", {"mask_cache": True, "inc_mask": True}),
    (32, 257, "geometry_searching", {"render": True, "tv": True}),
    (32, 300, "coarse", {"grad_mode": "raw"}),            # the reference's non-default gradient volumes (model/nerf.py:495-506)
    (32, 300, "coarse", {"grad_mode": "grad_conv"}),
])
def test_fused_coarse_vs_oracle_and_composed(dev, oracle, G, N, stage, extra):
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import render_losses
    from fgs_nerf_amd.nerf import MaskCache
    cfg = synth.COARSE_MODEL if stage == "coarse" else synth.GEOMETRY_MODEL
    rays_c = synth.random_rays(N, seed=33)
    rays = tuple(r.to(dev) for r in rays_c)
    target_c = torch.rand(N, 3, generator=torch.Generator().manual_seed(8))
    target = target_c.to(dev)
    lossw = synth.COARSE_LOSS
    over = {"grad_mode": extra["grad_mode"]} if extra.get("grad_mode") else {}
    a = synth.build_model(G, cfg, device=dev, fused=True, **over)
    b = synth.build_model(G, cfg, device=dev, fused=False, **over)
    if over:
        from fgs_nerf_amd import fused as fused_mod
        assert fused_mod.supports_coarse(a)              # the fused kernels cover these modes, not only the composed path
    for m in (a, b):
        if extra.get("mask_cache"):
            sdf_mask = ((m.sdf.grid.detach() < 0.25) * 1e-3).float()
            m.mask_cache = MaskCache(path=None, mask_cache_thres=1e-3 * 0.5, sdf_mask=sdf_mask.cpu(),
                                     xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1]).to(dev)
        if extra.get("inc_mask"):
            m.set_inc_mask([0.1, 0.0, 0.2], [0.9, 0.8, 1.0])
    rflags = dict(render_grad=True) if extra.get("render") else {}
    kw = dict(synth.RENDER_KWARGS, **rflags)
    tv_w = 0.1 if extra.get("tv") else 0.0

    def step(model):
        for p in model.parameters():
            p.grad = None
        res = model(*rays, global_step=700, **kw)
        loss = render_losses(res, target, lossw, model)
        if tv_w:   # model.gradient must stay an autograd node over sdf.grid (model/nerf.py:440-446)
            loss = loss + model.density_total_variation(sdf_tv=0, smooth_grad_tv=tv_w)
        loss.backward()
        return res, loss
    ra, la = step(a)
    rb, lb = step(b)

    P = synth.oracle_params(b)
    leaves = {'sdf': P['sdf'], 'k0': P['k0']}
    for i, (W, bias) in enumerate(P['refnet']):
        leaves[f'refnet.{i}.weight'], leaves[f'refnet.{i}.bias'] = W, bias
    for t in leaves.values():
        t.requires_grad_(True)
    ro = oracle.forward_coarse(P, *rays_c, global_step=700, near=2.0, stepsize=0.5, bg=1, stage=stage, **rflags)
    lo = render_losses(ro, target_c, lossw)
    if tv_w:
        gv = oracle.neus_sdf_gradient(P['sdf'], P['voxel_size']).permute(1, 0, 2, 3, 4)
        kern = b.tv_smooth_conv.weight.detach().cpu()
        sm = F.conv3d(F.pad(gv, (1,) * 6, mode='replicate'), kern)
        lo = lo + ((sm.detach() - gv) ** 2).mean() * tv_w
    lo.backward()

    assert ra["ray_id"].shape[0] > 100
    assert torch.equal(ra["ray_id"].cpu(), ro["ray_id"]) and torch.equal(ra["ray_id"], rb["ray_id"])
    assert torch.equal(ra["step_id"].cpu(), ro["step_id"])
    for key in ("rgb_marched", "sigmoid_rgb", "weights", "raw_rgb", "normal", "alphainv_cum", "raw_alpha", "gradient", "depth"):
        assert rel_l2(ra[key], ro[key]) < 1e-5, key
        assert rel_l2(ra[key], rb[key]) < 1e-5, key
    if extra.get("render"):
        assert rel_l2(ra["normal_marched"], ro["normal_marched"]) < 1e-5
    assert abs(float(la) - float(lo)) < 1e-6 and abs(float(la) - float(lb)) < 1e-6
    assert torch.equal(ra["mask"], rb["mask"]) and torch.equal(ra["mask_outbbox"], rb["mask_outbbox"])
    ga, gb = grads_of(a), grads_of(b)
    for k in ga:
        assert rel_l2(ga[k], leaves[k].grad) < 1e-3, k
        assert rel_l2(gb[k], leaves[k].grad) < 1e-2, k


def test_fused_coarse_full_size_vs_oracle_forward(dev, oracle):
    """BASELINE config 3 geometry at the bench size (160^3, 4096 rays): rendered pixels vs the CPU oracle."""
    from fgs_nerf_amd import synth
    model = synth.build_model(160, synth.COARSE_MODEL, device=dev)
    ro, rd, vd = synth.random_rays(4096)
    with torch.no_grad():
        res = model(ro.to(dev), rd.to(dev), vd.to(dev), global_step=1000, **synth.RENDER_KWARGS)
        ref = oracle.forward_coarse(synth.oracle_params(model), ro, rd, vd, global_step=1000, near=2.0, stepsize=0.5, bg=1)
    assert res["weights"].shape[0] > 20_000
    ia, ib, _ = match_survivors(res, ref, label="160^3 coarse")   # identical, or every difference explained and printed
    assert rel_l2(res["rgb_marched"], ref["rgb_marched"]) < 1e-5
    assert rel_l2(res["alphainv_cum"], ref["alphainv_cum"]) < 1e-5
    assert rel_l2(res["weights"].cpu()[ia], ref["weights"][ib]) < 1e-5
    assert rel_l2(res["raw_rgb"].cpu()[ia], ref["raw_rgb"][ib]) < 1e-5
    w_sum = torch.zeros(4096, device=dev).index_add_(0, res["ray_id"], res["weights"])
    assert float((w_sum + res["alphainv_cum"]).max()) <= 1.0 + 1e-5
    assert bool((res["ray_id"][1:] >= res["ray_id"][:-1]).all())


def test_interleaved_volume_lookups_are_bit_identical(dev):
    """The coarse march samples the smoothed SDF and the gradient volume either from four arrays (32 four-byte gathers per
    point) or from the voxel-interleaved [X,Y,Z,4] copy the gradient-volume pass leaves behind (8 sixteen-byte loads): same
    corner order, same fmaf chain -> every output of the render must be bit-identical, forward and backward."""
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.losses import fused_render_losses
    rays = tuple(t.to(dev) for t in synth.random_rays(700, seed=21))
    target = torch.rand(700, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    outs = {}
    old = fused.FLAGS['coarse_vol4']
    try:
        for flag in (True, False):
            fused.FLAGS['coarse_vol4'] = flag
            model = synth.build_model(40, synth.COARSE_MODEL, device=dev)
            res = model(*rays, global_step=300, **synth.RENDER_KWARGS)
            loss = fused_render_losses(res, target, synth.COARSE_LOSS, model)
            loss.backward()
            outs[flag] = ({k: res[k].detach().clone() for k in ('rgb_marched', 'weights', 'ray_id', 'step_id', 'raw_alpha',
                                                                 'gradient', 'alphainv_cum')},
                          model.sdf.grid.grad.clone(), float(loss))
    finally:
        fused.FLAGS['coarse_vol4'] = old
    for k in outs[True][0]:
        assert torch.equal(outs[True][0][k], outs[False][0][k]), k
    # (the loss scalar is a sum of per-workgroup partials added with float atomics: equal up to their order, like sdf.grad)
    assert abs(outs[True][2] - outs[False][2]) <= 1e-6 * abs(outs[False][2])
    # (sdf.grad goes through float atomics in both runs: equal up to their order)
    a, b = outs[True][1], outs[False][1]
    assert float((a - b).norm() / b.norm()) < 1e-5
