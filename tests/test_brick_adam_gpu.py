"""MaskedAdam over the bricks a step touched, with a self-cleaning gradient buffer (csrc/bricks.hip fgs_adam_upd_bricks,
fused.py `_take_grid_grad`, adam.MaskedAdam._bricks).  Per element the update is masked_adam_upd
(model/cuda/adam_upd_kernel.cu:25-40; model/adam.py:205-221 picks it for skip_zero_grad groups): BIT-exact against the oracle
on the same gradient.  Checked: the kernel alone (flag and list selection, partial bricks, device step size, skip flag), the
occupancy really covers every element the feature-grid scatter writes (the buffer is all-zero again after the update), the
fused training step with and without the buffer gives the same parameters, and every way of breaking the "non-zero only inside
the recorded bricks" invariant falls back to the dense update."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cl(shape, dev, fill=None, rng=None):
    """channel-last [1,C,X,Y,Z] tensor (storage [X][Y][Z][C])"""
    _, C, X, Y, Z = shape
    t = torch.empty_strided(shape, (C * X * Y * Z, 1, Y * Z * C, Z * C, C), dtype=torch.float32, device=dev)
    if rng is not None:
        t.copy_(torch.from_numpy(rng.randn(*shape).astype(np.float32)))
    else:
        t.fill_(0.0 if fill is None else fill)
    return t


def _storage(t):
    """the tensor's elements in storage order, as numpy"""
    return t.detach().as_strided((t.numel(),), (1,)).cpu().numpy().copy()


def _flags_for(pts, lo, hi, dims, dev):
    from fgs_nerf_amd._lib import call, ptr, stream
    C, X, Y, Z = dims
    flags = torch.zeros(((X + 3) // 4) * ((Y + 3) // 4) * ((Z + 3) // 4), dtype=torch.int32, device=dev)
    call("fgs_brick_flags_pts", ptr(pts), pts.shape[0], (ctypes.c_float * 3)(*lo), (ctypes.c_float * 3)(*hi), X, Y, Z,
         ptr(flags), None, stream())
    return flags


def _voxel_mask(flags, dims):
    C, X, Y, Z = dims
    nb = ((X + 3) // 4, (Y + 3) // 4, (Z + 3) // 4)
    m = flags.reshape(nb).bool()
    m = m.repeat_interleave(4, 0).repeat_interleave(4, 1).repeat_interleave(4, 2)[:X, :Y, :Z]
    return m[None, None].expand(1, C, X, Y, Z)


@pytest.mark.parametrize("dims,select", [((12, 20, 24, 28), "flags"), ((12, 20, 24, 28), "list"), ((8, 18, 21, 23), "flags"),
                                         ((4, 5, 6, 7), "list"), ((16, 32, 32, 32), "flags")])
def test_brick_adam_is_bit_exact_and_consumes_the_gradient(dev, oracle, dims, select):
    from fgs_nerf_amd._lib import call, ptr, stream
    C, X, Y, Z = dims
    shape = (1, C, X, Y, Z)
    rng = np.random.RandomState(X * 7 + C)
    lo, hi = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
    pts = torch.from_numpy((rng.rand(150, 3) * 1.6 - 0.8).astype(np.float32)).to(dev)
    p, m, v = _cl(shape, dev, rng=rng), _cl(shape, dev), _cl(shape, dev)
    ref_p, ref_m, ref_v = _storage(p), _storage(m), _storage(v)
    for step in (1, 2, 3):
        flags = _flags_for(pts[(step - 1) * 50:step * 50], lo, hi, dims, dev)
        assert 0 < int(flags.sum()) <= flags.numel()
        g = _cl(shape, dev, rng=rng)
        g.mul_(_voxel_mask(flags, dims)).mul_(torch.from_numpy((rng.rand(*shape) > 0.4).astype(np.float32)).to(dev))
        g_ref = _storage(g)
        idx = count = None
        if select == "list":
            idx = torch.empty(flags.numel(), dtype=torch.int64, device=dev)
            count = torch.zeros(1, dtype=torch.int64, device=dev)
            call("fgs_brick_compact", ptr(flags), flags.numel(), ptr(idx), ptr(count), stream())
        call("fgs_adam_upd_bricks", ptr(p), ptr(g), ptr(m), ptr(v), C, X, Y, Z, ptr(idx), ptr(count), 0, ptr(flags), step, 0.9,
             0.99, 0.1, 1e-8, None, None, stream())
        oracle.K.adam_upd(ref_p, g_ref, ref_m, ref_v, step, 0.9, 0.99, 0.1, 1e-8, mode=1)
        assert np.array_equal(_storage(p), ref_p) and np.array_equal(_storage(m), ref_m) and np.array_equal(_storage(v), ref_v)
        assert not bool(g.any()) and not bool(flags.any())          # gradient consumed, occupancy cleared


def _masks_for(pts, lo, hi, dims, dev):
    from fgs_nerf_amd._lib import call, ptr, stream
    C, X, Y, Z = dims
    masks = torch.zeros(((X + 3) // 4) * ((Y + 3) // 4) * ((Z + 3) // 4) * 64, dtype=torch.uint8, device=dev)
    call("fgs_brick_masks_pts", ptr(pts), pts.shape[0], (ctypes.c_float * 3)(*lo), (ctypes.c_float * 3)(*hi), X, Y, Z,
         ptr(masks), None, stream())
    return masks


def _voxels_of(masks, dims):
    """bool [1,C,X,Y,Z]: the voxels whose byte (16 x' + 4 y' + z' of their brick's 64) is set"""
    C, X, Y, Z = dims
    nb = ((X + 3) // 4, (Y + 3) // 4, (Z + 3) // 4)
    m = masks.reshape(*nb, 4, 4, 4).cpu().numpy() != 0                    # [bx, by, bz, x', y', z']
    out = np.ascontiguousarray(m.transpose(0, 3, 1, 4, 2, 5)).reshape(nb[0] * 4, nb[1] * 4, nb[2] * 4)
    return torch.from_numpy(out[:X, :Y, :Z].copy())[None, None].expand(1, C, X, Y, Z)


@pytest.mark.parametrize("dims", [(12, 20, 24, 28), (8, 18, 21, 23), (4, 5, 6, 7), (16, 32, 32, 32), (64, 8, 8, 8)])
def test_voxel_adam_is_bit_exact_and_consumes_the_gradient(dev, oracle, dims):
    """fgs_brick_masks_pts + fgs_adam_upd_voxels: the recorded voxels are exactly the in-volume trilinear corners of the
    points (checked against a torch restatement of the index arithmetic), the update over them is bit-identical to the
    oracle's masked Adam, gradient and masks are zero afterwards.  With a device step size and the skip flag, too."""
    from fgs_nerf_amd._lib import call, lib, ptr, stream
    C, X, Y, Z = dims
    shape = (1, C, X, Y, Z)
    rng = np.random.RandomState(X + C)
    lo, hi = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
    pts = torch.from_numpy((rng.rand(90, 3) * 2.2 - 1.1).astype(np.float32)).to(dev)       # some outside the box
    p, m, v = _cl(shape, dev, rng=rng), _cl(shape, dev), _cl(shape, dev)
    ref_p, ref_m, ref_v = _storage(p), _storage(m), _storage(v)
    ss = torch.zeros(1, dtype=torch.float32, device=dev)
    skip = torch.zeros(1, dtype=torch.int32, device=dev)
    for step, (dev_sched, skipped) in enumerate([(False, False), (True, False), (True, True), (False, False)], start=1):
        masks = _masks_for(pts[(step - 1) * 20:step * 20 + 10], lo, hi, dims, dev)
        vox = _voxels_of(masks, dims).to(dev)
        if step == 1:       # the corners, restated: index = (p - lo) / (hi - lo) * (n - 1), floor and floor + 1 per axis
            q = pts[0:30].double().cpu()
            want = torch.zeros(X, Y, Z, dtype=torch.bool)
            f = (q + 1.0) / 2.0 * torch.tensor([X - 1, Y - 1, Z - 1], dtype=torch.float64)
            f0 = torch.floor(f.float()).long()      # the kernels compute the index in float32
            f32 = ((pts[0:30].cpu() - torch.tensor(lo)) / (torch.tensor(hi) - torch.tensor(lo))).float()
            for dx in (0, 1):
                for dy in (0, 1):
                    for dz in (0, 1):
                        c = f0 + torch.tensor([dx, dy, dz])
                        ok = ((c >= 0) & (c < torch.tensor([X, Y, Z]))).all(1)
                        want[c[ok, 0], c[ok, 1], c[ok, 2]] = True
            got = vox[0, 0].cpu()
            assert int((got ^ want).sum()) <= 2, int((got ^ want).sum())      # (a point within an ulp of a lattice plane)
        g = _cl(shape, dev, rng=rng)
        g.mul_(vox).mul_(torch.from_numpy((rng.rand(*shape) > 0.3).astype(np.float32)).to(dev))
        g_ref = _storage(g)
        skip.fill_(int(skipped))
        ss.fill_(lib().fgs_adam_step_size(step, 0.9, 0.99, 0.05))
        call("fgs_adam_upd_voxels", ptr(p), ptr(g), ptr(m), ptr(v), C, X, Y, Z, ptr(masks), step, 0.9, 0.99, 0.05, 1e-8,
             ptr(ss) if dev_sched else None, ptr(skip) if dev_sched else None, stream())
        if not skipped:
            oracle.K.adam_upd(ref_p, g_ref, ref_m, ref_v, step, 0.9, 0.99, 0.05, 1e-8, mode=1)
        assert np.array_equal(_storage(p), ref_p) and np.array_equal(_storage(m), ref_m) and np.array_equal(_storage(v), ref_v)
        assert not bool(g.any()) and not bool(masks.any())


def test_brick_adam_device_step_size_and_skip_flag(dev, oracle):
    from fgs_nerf_amd._lib import call, lib, ptr, stream
    dims = (12, 16, 16, 16)
    C, X, Y, Z = dims
    shape = (1, C, X, Y, Z)
    rng = np.random.RandomState(3)
    pts = torch.from_numpy((rng.rand(40, 3) - 0.5).astype(np.float32)).to(dev)
    lo, hi = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
    p, m, v = _cl(shape, dev, rng=rng), _cl(shape, dev), _cl(shape, dev)
    ref = [_storage(t) for t in (p, m, v)]
    ss = torch.tensor([lib().fgs_adam_step_size(7, 0.9, 0.99, 0.05)], dtype=torch.float32, device=dev)
    skip = torch.zeros(1, dtype=torch.int32, device=dev)
    for it, skipped in enumerate((False, True, False)):
        skip.fill_(int(skipped))
        flags = _flags_for(pts, lo, hi, dims, dev)
        g = _cl(shape, dev, rng=rng).mul_(_voxel_mask(flags, dims))
        g_ref = _storage(g)
        call("fgs_adam_upd_bricks", ptr(p), ptr(g), ptr(m), ptr(v), C, X, Y, Z, None, None, 0, ptr(flags), 0, 0.9, 0.99, 0.0,
             1e-8, ptr(ss), ptr(skip), stream())
        if not skipped:
            oracle.K.adam_upd(ref[0], g_ref, ref[1], ref[2], 7, 0.9, 0.99, 0.05, 1e-8, mode=1)
        for t, r in zip((p, m, v), ref):
            assert np.array_equal(_storage(t), r), (it, skipped)
        assert not bool(g.any()) and not bool(flags.any())          # consumed even when the update is skipped


def _train(dev, steps, brick, oracle, tv_dense_k0=False, G=48, N=1024):
    """A few eager fused training steps.  In EVERY step the feature grid's update is checked bit for bit: parameter, moments
    and gradient are snapshotted right before optimizer.step(), and the step's result must equal the oracle's masked Adam
    (model/cuda/adam_upd_kernel.cu:25-40) applied to that snapshot -- whichever kernel MaskedAdam chose."""
    import bench
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.losses import fused_render_losses
    old = fused.FLAGS['brick_adam']
    fused.FLAGS['brick_adam'] = brick
    try:
        model = synth.build_model(G, synth.FINE_MODEL, device=dev)
        opt = bench.make_optimizer(model)
        opt.ensure_state()
        k0 = model.k0.grid
        group = next(g for g in opt.param_groups if any(q is k0 for q in g['params']))
        used, bricked = [], []
        for it in range(steps):
            ro, rd, vd = (t.to(dev).contiguous() for t in synth.random_rays(N, seed=90 + it))
            target = torch.rand(N, 3, generator=torch.Generator().manual_seed(it)).to(dev)
            res = model(ro, rd, vd, global_step=2000 + it, **synth.RENDER_KWARGS)
            loss = fused_render_losses(res, target, synth.FINE_LOSS, model)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            gb = model._fused_cache.get('k0_grad')
            used.append(gb is not None and k0.grad.data_ptr() == gb['buf'].data_ptr())
            if tv_dense_k0 and it == 1:
                model.k0_total_variation_add_grad(1e-4, True)
            st = opt.state[k0]
            ref = [_storage(t) for t in (k0, st['exp_avg'], st['exp_avg_sq'])]
            g_ref = _storage(k0.grad)
            opt.step()
            torch.cuda.synchronize()
            b1, b2 = group['betas']
            oracle.K.adam_upd(ref[0], g_ref, ref[1], ref[2], st['step'], b1, b2, group['lr'], group['eps'], mode=1)
            for t, r, name in zip((k0, st['exp_avg'], st['exp_avg_sq']), ref, ("param", "exp_avg", "exp_avg_sq")):
                assert np.array_equal(_storage(t), r), (it, name)
            bricked.append(bool(gb is not None and used[-1] and gb['clean']))
            if bricked[-1]:       # the brick update ran: everything the scatter wrote has been consumed
                assert not bool(gb['buf'].any()) and not bool(gb['flags'].any()), it
        return model._fused_cache, used, bricked
    finally:
        fused.FLAGS['brick_adam'] = old


def test_fused_step_with_the_persistent_gradient_buffer_is_bit_exact(dev, oracle):
    """The fused backward pass scatters into the persistent buffer, MaskedAdam takes the brick update: in every step the
    result equals the oracle's masked Adam on that step's gradient bit for bit, and the buffer is all-zero afterwards (so
    the recorded bricks covered every element the scatter wrote).  The plain path (fresh zero-filled gradient, dense masked
    update) passes the same check."""
    cache, used, bricked = _train(dev, 4, True, oracle)
    assert all(used) and all(bricked) and cache['k0_grad']['clean'], (used, bricked)
    cache, used, bricked = _train(dev, 2, False, oracle)
    assert not any(used) and not any(bricked) and 'k0_grad' not in cache


def test_dense_tv_on_the_feature_grid_falls_back_and_recovers(dev, oracle):
    """A dense TV term on k0 (model/nerf.py:341-344 k0_total_variation_add_grad, dense_mode) writes outside the recorded
    bricks: that step takes the dense masked update (still bit-exact) and leaves the buffer dirty; the next step gets a
    zero-filled buffer again and is back on the brick path."""
    cache, used, bricked = _train(dev, 4, True, oracle, tv_dense_k0=True)
    assert all(used) and bricked == [True, False, True, True] and cache['k0_grad']['clean'], (used, bricked)


def test_gradient_accumulation_and_foreign_writes_are_detected(dev):
    import bench
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import fused_render_losses
    model = synth.build_model(48, synth.FINE_MODEL, device=dev)
    opt = bench.make_optimizer(model)
    ro, rd, vd = (t.to(dev).contiguous() for t in synth.random_rays(512, seed=5))
    target = torch.rand(512, 3).to(dev)

    def backward():
        res = model(ro, rd, vd, global_step=2000, **synth.RENDER_KWARGS)
        fused_render_losses(res, target, synth.FINE_LOSS, model).backward()

    # 1. two backward passes without zero_grad: the second must not reuse (and wipe) the buffer that IS p.grad
    opt.zero_grad(set_to_none=True)
    backward()
    g1 = model.k0.grid.grad.clone()
    gb = model._fused_cache['k0_grad']
    assert model.k0.grid.grad.data_ptr() == gb['buf'].data_ptr()
    backward()
    torch.cuda.synchronize()
    assert float((model.k0.grid.grad - 2 * g1).abs().max()) <= 1e-5 * float(g1.abs().max())
    before = model.k0.grid.detach().clone()
    opt.step()                       # autograd accumulated in place (version bump): the dense update must have run
    assert not gb['clean'] and not torch.equal(model.k0.grid.detach(), before)
    # 2. somebody adds a dense term with torch ops
    opt.zero_grad(set_to_none=True)
    backward()
    model.k0.grid.grad.add_(1e-3)
    opt.step()
    assert not gb['clean']
    # 3. a clean step afterwards is back on the brick path
    opt.zero_grad(set_to_none=True)
    backward()
    opt.step()
    torch.cuda.synchronize()
    assert gb['clean'] and not bool(gb['buf'].any())


def test_resumed_reference_layout_moments_take_the_same_step_as_the_oracle(dev, oracle):
    """moments loaded from a reference-layout (NCDHW-contiguous) optimizer state next to a channel-last k0 --
    after load_state_dict one MaskedAdam step equals the oracle's masked Adam on the same logical tensors, bit for bit."""
    from fgs_nerf_amd.adam import MaskedAdam
    from fgs_nerf_amd.grid import to_grid_layout
    g = torch.Generator().manual_seed(3)
    shape = (1, 12, 8, 12, 16)
    p0, m0, v0 = torch.randn(shape, generator=g), torch.randn(shape, generator=g) * 0.1, torch.rand(shape, generator=g) * 0.01
    grad = torch.randn(shape, generator=g) * (torch.rand(shape, generator=g) < 0.3)
    p = torch.nn.Parameter(to_grid_layout(p0.to(dev)))
    opt = MaskedAdam([{'params': [p], 'lr': 0.1, 'skip_zero_grad': True}])
    opt.ensure_state()
    sd = opt.state_dict()
    sd['state'][0].update(exp_avg=m0.clone(), exp_avg_sq=v0.clone(), step=4)       # NCDHW-contiguous, as the reference saves
    opt.load_state_dict(sd)
    p.grad = to_grid_layout(grad.to(dev))
    opt.step()
    pr, mr, vr = p0.numpy().copy(), m0.numpy().copy(), v0.numpy().copy()
    oracle.K.adam_upd(pr.reshape(-1), grad.numpy().reshape(-1), mr.reshape(-1), vr.reshape(-1), 5, 0.9, 0.99, 0.1, 1e-8, mode=1)
    assert torch.equal(p.detach().cpu().contiguous(), torch.from_numpy(pr))
    assert torch.equal(opt.state[p]['exp_avg'].cpu().contiguous(), torch.from_numpy(mr))


def test_second_backward_before_step_with_the_in_backward_update_raises(dev):
    """with k0's Adam pass issued from inside the backward pass, a second backward() before step() used to have
    its k0 gradient dropped without a word; it raises now."""
    import bench
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.losses import fused_render_losses
    model = synth.build_model(32, synth.FINE_MODEL, device=dev)
    opt = bench.make_optimizer(model)
    fused.enable_early_update(model, opt, None, inline=True)
    rays = tuple(r.to(dev) for r in synth.random_rays(256, seed=2))
    target = torch.rand(256, 3, device=dev)
    fused_render_losses(model(*rays, global_step=1000, **synth.RENDER_KWARGS), target, synth.FINE_LOSS, model).backward()
    with pytest.raises(RuntimeError, match="already updated"):
        fused_render_losses(model(*rays, global_step=1000, **synth.RENDER_KWARGS), target, synth.FINE_LOSS, model).backward()
