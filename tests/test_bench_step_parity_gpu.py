"""The step bench.py TIMES, tied to the oracle.

bench.py's headline number is one hipGraph replay per step of the configs[1] iteration (160^3, 4096 rays).  Until now the oracle chain
reached that step only through two hops (oracle <-> eager fused path at 1024 rays; eager <-> captured at 512 rays on a small grid).
Here the captured step is built by bench.py's own functions (`make_optimizer`, `make_batch`, `build_captured`, the in-backward k0 Adam
pass switched on as bench.py switches it on), replayed ONCE on batch 0, and compared with the CPU oracle on the full 4096-ray batch:

(a) the replay's loss scalar vs `oracle.forward_fine` + `render_losses`: <= 1e-6 relative;
(b) after that one replay, every trained tensor AND both of its Adam moments vs the oracle's backward pass + the oracle's C
    restatements of the TV add-grad and the Adam kernels (`orc_tv_add_grad`, `orc_adam_upd`; total_variation_kernel.cu:13-78,
    adam_upd_kernel.cu:8-72), reference: model/nerf_training.py:300-373.
      * exp_avg = (1 - beta1) g and exp_avg_sq = (1 - beta2) g^2 are the gradient itself, element by element: rel-L2 <= 2e-5 (the flat
        gradient bar of tests/test_fullsize_parity_gpu.py), and exp_avg != 0 on exactly the oracle's support for the masked group;
      * update arithmetic, element by element, NO floor: the oracle's Adam kernel applied to the HIP step's OWN gradient (recovered
        from its exp_avg = (1 - beta1) g, exact to an ulp) must reproduce every HIP parameter within 1e-5 of the update scale
        (|p_old| + lr); the exceptional set must be EMPTY.  Together with the moment bars this is the whole statement: same gradient
        (in norm, to float32 summation noise), same update rule (per element, to rounding).
      * parameters end to end, HIP vs oracle, element by element: every element within 2 lr (asserted), rel-L2 printed, and the
        counts VERDICT r3 asked for -- elements whose oracle |g| exceeds the floor max(1e-3 max|g|, 1000 e') must sit within 1e-5
        of the update scale, the rest are counted.  Why that floor is nearly vacuous HERE, and why the second bullet replaces it:
        Adam's first step moves an element by lr g / (|g| + e'), e' = eps / sqrt(1 - beta2) = 1e-7, and with the reference's loss
        scaling (means over 4096 x 3 pixels) all but ~500 of the 53 M gradient elements are BELOW 1000 e' -- the whole model trains
        in Adam's eps regime, where an ABSOLUTE gradient error dg moves the result by lr dg e' / (|g| + e')^2.  An sdf voxel's
        gradient is a float32 sum of hundreds of cancelling atomic contributions; its summation noise (1e-10 absolute) is worth up
        to 1e-2 lr there (first run: 2771 of 4.1 M sdf elements beyond the 1e-5 bar, worst 1.04e-2 lr; <= 14 per MLP tensor).
The ReLU sign decisions of the HIP forward chain are replayed in the oracle (`relu_masks`), as in the full-size parity test: a few
dozen of 10^8 hidden units sit within float32 rounding of zero (tests/test_stagewise_bwd_gpu.py).
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

FLOOR_FRAC = 1e-3      # "gradient above the floor" = |g| > max(FLOOR_FRAC * max|g| of that tensor, FLOOR_EPS * e')
FLOOR_EPS = 1000.0     # ... in units of e' = eps / sqrt(1 - beta2), the constant Adam adds to |g| in its first step
TIGHT = 1e-5           # parameter bar above the floor, relative to (|p_old| + lr)
MOMENT = 2e-5          # rel-L2 bar of both Adam moments (= tests/test_fullsize_parity_gpu.py DIRECT_GRAD)


def _named(model):
    from fgs_nerf_amd.nerf import mlp_layers
    out = [('sdf', model.sdf.grid, 'sdf'), ('k0', model.k0.grid, 'k0')]
    for net in ('rgbnet', 'refnet'):
        for i, l in enumerate(mlp_layers(getattr(model, net))):
            out += [(f'{net}.{i}.weight', l.weight, net), (f'{net}.{i}.bias', l.bias, net)]
    return out


def test_one_replay_of_the_timed_step_matches_the_oracle(dev, oracle):
    import bench
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.dist import GradAverager
    from fgs_nerf_amd.losses import render_losses
    assert bench.GRID == 160 and bench.RAYS_PER_GPU == 4096 and bench.GLOBAL_STEP == 1000
    N = bench.RAYS_PER_GPU
    # ---- exactly what bench.py main() builds for one GPU
    model = synth.build_model(bench.GRID, synth.FINE_MODEL, device=dev)
    opt = bench.make_optimizer(model)
    averager = GradAverager(model.parameters())
    averager.attach(model)
    averager.attach_optimizer(opt)
    fused.enable_early_update(model, opt, averager, inline=True)
    batch = bench.make_batch(0, 0, dev)
    P = synth.oracle_params(model)                          # CPU snapshot of the initial parameters
    old = {name: p.detach().cpu().clone() for name, p, _ in _named(model)}
    lr_of = {g['name']: g['lr'] for g in opt.param_groups}

    # one eager forward (no backward, no update): the survivor count that sizes the buffers, and the forward chain's ReLU decisions
    res = model(batch[0], batch[1], batch[2], global_step=bench.GLOBAL_STEP, **synth.RENDER_KWARGS)
    n_surv = int(res['weights'].shape[0])
    saved = res['rgb_marched'].grad_fn.run.saved
    masks = dict(rgbnet=[(a[:n_surv] > 0).cpu() for a in saved['acts_rgb'][1:]],
                 refnet=[(a[:n_surv] > 0).cpu() for a in saved['acts_ref'][1:]])
    del res, saved
    step = bench.build_captured(model, opt, bench.survivor_capacity(n_surv), n_iters=4, n_rays_global=N)
    step.capture(batch)
    for name, p, _ in _named(model):                        # capture() itself must not train
        assert torch.equal(p.detach().cpu(), old[name]), name
    loss_hip = float(step.replay(torch.stack(batch).contiguous()).clone())
    overflow, total = step.check()
    assert not overflow and total == n_surv, (overflow, total, n_surv)

    # ---- the oracle: forward_fine + losses + backward on the same 4096 rays, TV add-grad, Adam step 1
    ro, rd, vd, target = (t.cpu() for t in batch)
    leaves = {'sdf': P['sdf'], 'k0': P['k0']}
    for net in ('rgbnet', 'refnet'):
        for i, (W, b) in enumerate(P[net]):
            leaves[f'{net}.{i}.weight'], leaves[f'{net}.{i}.bias'] = W, b
    for t in leaves.values():
        t.requires_grad_(True)
    r32 = oracle.forward_fine(P, ro, rd, vd, global_step=bench.GLOBAL_STEP, near=2.0, stepsize=0.5, bg=1, relu_masks=masks)
    assert r32['weights'].shape[0] == n_surv
    l32 = render_losses(r32, target, synth.FINE_LOSS)
    l32.backward()
    l32 = l32.detach()
    rel_loss = abs(loss_hip - float(l32)) / abs(float(l32))
    rs = r32['relu_stats']
    print(f"\n[timed step vs oracle] 160^3, {N} rays, in-bbox samples {r32['n_inbbox']}, survivors {n_surv}; loss HIP {loss_hip:.9g} "
          f"oracle {float(l32):.9g} (rel {rel_loss:.2e}); ReLU decisions replayed against the oracle's own sign: "
          f"{rs['relu_flips']} of {rs['relu_units']}")
    assert rel_loss <= 1e-6
    g_sdf = np.ascontiguousarray(P['sdf'].grad.numpy())
    w_tv = float(np.float32(bench.tv_args(N)[0]) * np.float32(bench.GRID) / np.float32(128))        # model/nerf.py:462,466
    oracle.K.total_variation_add_grad(np.ascontiguousarray(P['sdf'].detach().numpy()), g_sdf, w_tv, w_tv, w_tv, True)
    grads = {k: (g_sdf if k == 'sdf' else np.ascontiguousarray(t.grad.numpy())) for k, t in leaves.items()}

    print("    %-20s %-10s %-11s %-11s %-11s %-10s %-18s %-14s" % ("tensor", "elements", "exp_avg", "exp_avg_sq", "param", "floor",
                                                                 "above floor", "sub-floor > tight   own-gradient update"))
    eps_eff = 1e-8 / np.sqrt(1.0 - 0.99)
    bad = []
    for name, p, grp in _named(model):
        lr = lr_of[grp]
        new = np.ascontiguousarray(leaves[name].detach().numpy()).reshape(-1).copy()
        m, v = np.zeros_like(new), np.zeros_like(new)
        g = grads[name].reshape(-1)
        oracle.K.adam_upd(new, g, m, v, 1, 0.9, 0.99, lr, 1e-8, mode=1 if grp == 'k0' else 0)
        st = opt.state[p]
        m_hip, v_hip = st['exp_avg'].detach().cpu().numpy().reshape(-1), st['exp_avg_sq'].detach().cpu().numpy().reshape(-1)
        p_hip = p.detach().cpu().numpy().reshape(-1)
        p_old = old[name].numpy().reshape(-1)
        # update arithmetic alone: the oracle's kernel on the HIP step's own gradient (exp_avg = (1 - beta1) g, inverted)
        g_own = (m_hip / (np.float32(1.0) - np.float32(0.9))).astype(np.float32)
        own = p_old.copy()
        oracle.K.adam_upd(own, g_own, np.zeros_like(own), np.zeros_like(own), 1, 0.9, 0.99, lr, 1e-8, mode=1 if grp == 'k0' else 0)
        own_bad = int((np.abs(p_hip.astype(np.float64) - own) > TIGHT * (np.abs(p_old).astype(np.float64) + lr)).sum())
        e_m, e_v, e_p = rel_l2(m_hip, m), rel_l2(v_hip, v), rel_l2(p_hip, new)
        floor = max(FLOOR_FRAC * float(np.abs(g).max()), FLOOR_EPS * eps_eff)
        above = np.abs(g) > floor
        diff = np.abs(p_hip.astype(np.float64) - new)
        tight = TIGHT * (np.abs(p_old).astype(np.float64) + lr)
        exceptional = int((above & (diff > tight)).sum())             # must be empty
        loose = int((~above & (diff > tight)).sum())                  # sub-floor elements that needed the 2 lr bar
        worst = float(diff.max())
        print("    %-20s %-10d %-11.3e %-11.3e %-11.3e %-10.2e %-18s %-14s" % (name, g.size, e_m, e_v, e_p, floor,
                                                                             f"{int(above.sum())} / bad {exceptional}",
                                                                             f"{loose} (max |d| {worst / lr:.2e} lr)   bad {own_bad}"))
        why = [w for w, c in (("exp_avg", e_m > MOMENT), ("exp_avg_sq", e_v > 2 * MOMENT), ("above-floor elements beyond the bar", exceptional),
                              ("update arithmetic on its own gradient", own_bad), ("beyond 2 lr", worst > 2 * lr),
                              (f"step counter {st['step']}", st['step'] != 1)) if c]
        if grp == 'k0':     # masked update: exactly the oracle's support moved (grad != 0 <=> a survivor's trilinear corner)
            if not np.array_equal(m_hip != 0, m != 0):
                why.append(f"support of exp_avg differs in {int(((m_hip != 0) != (m != 0)).sum())} elements")
            if np.any(p_hip[m_hip == 0] != p_old[m_hip == 0]):      # (an update below half an ulp moves nothing: only this direction is exact)
                why.append(f"{int((p_hip[m_hip == 0] != p_old[m_hip == 0]).sum())} elements without a gradient moved")
        if why:
            bad.append((name, why))
    assert not bad, bad
