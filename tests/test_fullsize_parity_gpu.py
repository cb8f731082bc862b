"""Forward AND backward parity at BASELINE sizes against the CPU oracle, with a tolerance that separates summation-order
noise from bugs.

Yardstick: the oracle is run twice on the same rays -- in float32 (the reference arithmetic) and in float64 with every
discrete decision of the float32 run replayed (oracle.params_f64 / `decisions=`: same kept samples, same stopping points).
For every compared tensor

    e_hip = rel-L2(HIP, oracle-f64)        e_ref = rel-L2(oracle-f32, oracle-f64)

and the bar is  e_hip <= 2 * e_ref + FLOOR, plus a FLAT direct bar HIP-vs-oracle-f32: outputs 1e-5, gradients DIRECT_GRAD = 2e-5.
e_ref is what float32 itself costs on this computation (accumulation order, expf/sigmoid ulps); a kernel bug shows as
e_hip >> e_ref.  FLOOR = 2e-6 covers tensors whose e_ref happens to be ~0.  Both numbers are printed for every tensor.

Round 2 had to accept 0.5 x e_ref (~3.5e-3 on sdf.grad) as the direct gradient bar: HIP and the float32 oracle sat 1.4e-3 / 2.4e-3
apart on sdf.grad / k0.grad.  tests/test_stagewise_bwd_gpu.py located that difference: a few dozen of the 24 M hidden ReLU units have
a pre-activation within float32 rounding of zero and come out on different sides in the two forward passes; each flips a whole
term of its sample's gradient.  Both oracle runs here therefore replay the HIP forward's ReLU sign decisions (oracle.mlp_apply
relu_masks) the same way the float64 run replays the float32 run's threshold decisions -- all three evaluations then
differentiate the same piecewise-linear network -- and the direct bar is flat again.  The number of replayed units that
differ from the oracle's own decision is printed and bounded (< 1e-5 of all units).

Cases: configs[1] (160^3 fine, 1024 rays of bench batch 0), the coarse stage at 160^3 (configs[2]'s path at the bench
size), 256^3 fine (the reference's own grid after its rescale), a 320^3 fine shard (configs[4]'s per-GPU shape).  All through the fused HIP path and the C ABI.
"""
import pytest
import torch

from conftest import match_survivors, rel_l2

pytestmark = pytest.mark.gpu

FLOOR = 2e-6
DIRECT_OUT = 1e-5      # rendered pixels, weights, normals ... HIP vs the float32 reference arithmetic (north_star: <= 1e-5)
DIRECT_GRAD = 2e-5     # every parameter gradient, HIP vs the float32 reference arithmetic on the same ReLU decisions (flat)


def _leaves(P):
    L = {'sdf': P['sdf'], 'k0': P['k0']}
    for net in ('rgbnet', 'refnet'):
        if P.get(net) is None:
            continue
        for i, (W, b) in enumerate(P[net]):
            L[f'{net}.{i}.weight'], L[f'{net}.{i}.bias'] = W, b
    for t in L.values():
        t.requires_grad_(True)
    return L


def _hip_grads(model):
    from fgs_nerf_amd.nerf import mlp_layers
    out = {'sdf': model.sdf.grid.grad, 'k0': model.k0.grid.grad}
    for net in ('rgbnet', 'refnet'):
        if getattr(model, net, None) is None:
            continue
        for i, l in enumerate(mlp_layers(getattr(model, net))):
            out[f'{net}.{i}.weight'], out[f'{net}.{i}.bias'] = l.weight.grad, l.bias.grad
    return out


def _case(dev, oracle, G, stage, n_rays, ray_seed, label):
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import render_losses
    kw = synth.FINE_MODEL if stage == 'fine' else synth.COARSE_MODEL
    lossw = dict(synth.FINE_LOSS if stage == 'fine' else synth.COARSE_LOSS, weight_rgbper=0.05)
    model = synth.build_model(G, kw, device=dev)
    ro, rd, vd = (t[:n_rays].contiguous() for t in synth.random_rays(4096, seed=ray_seed))
    target = torch.rand(n_rays, 3, generator=torch.Generator().manual_seed(12))
    fwd = oracle.forward_fine if stage == 'fine' else oracle.forward_coarse
    okw = dict(global_step=1000, near=2.0, stepsize=0.5, bg=1)

    # HIP
    res = model(ro.to(dev), rd.to(dev), vd.to(dev), global_step=1000, **synth.RENDER_KWARGS)
    saved = res['rgb_marched'].grad_fn.run.saved          # the forward chain's post-ReLU activations -> its sign decisions
    if stage == 'fine':
        masks = dict(rgbnet=[(a > 0).cpu() for a in saved['acts_rgb'][1:]], refnet=[(a > 0).cpu() for a in saved['acts_ref'][1:]])
    else:
        masks = dict(refnet=[(a > 0).cpu() for a in saved['acts'][1:]])
    loss = render_losses(res, target.to(dev), lossw, model)
    loss.backward()
    hip = {k: v.detach().cpu() for k, v in _hip_grads(model).items()}
    okw['relu_masks'] = masks
    # oracle, float32 (the reference arithmetic)
    P = synth.oracle_params(model)
    L32 = _leaves(P)
    r32 = fwd(P, ro, rd, vd, **okw)
    l32 = render_losses(r32, target, lossw)
    l32.backward()
    # oracle, float64 on the float32 run's decisions (the yardstick)
    Q = oracle.params_f64(P)
    L64 = _leaves(Q)
    r64 = fwd(Q, ro, rd, vd, decisions=r32['decisions'], **okw)
    l64 = render_losses(r64, target.double(), lossw)
    l64.backward()

    ia, ib, n_flips = match_survivors(res, r32, label=label)
    rows, bad = [], []

    def check(name, a_hip, a32, a64):
        e_hip, e_ref, e_dir = rel_l2(a_hip, a64), rel_l2(a32, a64), rel_l2(a_hip, a32)
        # second bar, direct: the yardstick can be loose where float32 itself is (the NeuS alpha is a difference of two
        # sigmoids, so d alpha / d sdf cancels badly in float32 and BOTH float32 evaluations sit 1e-3..1e-2 from float64):
        # there a wrong kernel could hide under e_ref, so HIP must also sit within DIRECT of the float32 reference
        # arithmetic itself (same formulas, only summation order / ReLU-flip noise apart)
        ok = e_hip <= 2.0 * e_ref + FLOOR and e_dir <= (DIRECT_GRAD if name.startswith('grad') else DIRECT_OUT)
        rows.append((name, e_hip, e_ref, e_dir, ok))
        if not ok:
            bad.append(name)
    for key in ('rgb_marched', 'sigmoid_rgb', 'alphainv_cum'):
        check(key, res[key].detach(), r32[key].detach(), r64[key].detach())
    for key in ('weights', 'raw_rgb', 'normal', 'raw_alpha'):
        check(key, res[key].detach().cpu()[ia], r32[key].detach()[ib], r64[key].detach()[ib])
    check('loss', loss.detach(), l32.detach(), l64.detach())
    for k in hip:
        check('grad ' + k, hip[k], L32[k].grad, L64[k].grad)
    rs = r32['relu_stats']
    print(f"\n[{label}] rays {n_rays}, in-bbox samples {r32['n_inbbox']}, survivors {r32['weights'].shape[0]}, "
          f"kept-sample differences {n_flips}, ReLU decisions replayed against the oracle's own sign: {rs['relu_flips']} of "
          f"{rs['relu_units']}")
    assert rs['relu_flips'] < 1e-5 * rs['relu_units']
    print("    %-24s %-12s %-18s %-12s" % ("tensor", "e_hip", "e_ref(f32 vs f64)", "HIP vs f32"))
    for name, e_hip, e_ref, e_dir, ok in rows:
        print("    %-24s %-12.3e %-18.3e %-12.3e %s" % (name, e_hip, e_ref, e_dir, "" if ok else "  <-- out of tolerance"))
    assert res['weights'].shape[0] > 5_000
    assert not bad, bad
    # the north_star's own bars, flat: rendered pixels <= 1e-5 rel-L2 vs the float32 reference arithmetic
    assert rel_l2(res['rgb_marched'], r32['rgb_marched']) < 1e-5


def test_fine_160_fwd_bwd_vs_oracle(dev, oracle):
    from fgs_nerf_amd import synth
    _case(dev, oracle, 160, 'fine', 1024, synth.SEED, "configs[1] 160^3 fine, 1024 rays of bench batch 0")


def test_coarse_160_fwd_bwd_vs_oracle(dev, oracle):
    from fgs_nerf_amd import synth
    _case(dev, oracle, 160, 'coarse', 1024, synth.SEED, "160^3 coarse stage, 1024 rays of bench batch 0")


def test_fine_256_fwd_bwd_vs_oracle(dev, oracle):
    """The reference's own late fine-stage grid: 256^3 after the rescale at iteration 15 000 (config/shiny_blender.py:203-204,222)."""
    _case(dev, oracle, 256, 'fine', 1024, 9, "reference fine stage after its rescale: 256^3, 1024 rays")


def test_fine_320_shard_fwd_bwd_vs_oracle(dev, oracle):
    _case(dev, oracle, 320, 'fine', 1024, 4, "configs[4] per-GPU shape: 320^3 fine, 1024 rays")
