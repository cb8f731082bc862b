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
         ptr(flags), stream())
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


def _train(dev, steps, brick, tv_dense_k0=False, G=48, N=1024):
    """a few eager fused training steps; returns (parameters, fused cache)"""
    import bench
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.losses import fused_render_losses
    old = fused._BRICK_ADAM
    fused._BRICK_ADAM = brick
    try:
        model = synth.build_model(G, synth.FINE_MODEL, device=dev)
        opt = bench.make_optimizer(model)
        used = []
        for it in range(steps):
            ro, rd, vd = (t.to(dev).contiguous() for t in synth.random_rays(N, seed=90 + it))
            target = torch.rand(N, 3, generator=torch.Generator().manual_seed(it)).to(dev)
            res = model(ro, rd, vd, global_step=2000 + it, **synth.RENDER_KWARGS)
            loss = fused_render_losses(res, target, synth.FINE_LOSS, model)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            gb = model._fused_cache.get('k0_grad')
            used.append(gb is not None and model.k0.grid.grad.data_ptr() == gb['buf'].data_ptr())
            if tv_dense_k0 and it == 1:
                model.k0_total_variation_add_grad(1e-4, True)
            opt.step()
            if gb is not None and used[-1]:
                rec_used = gb['clean']
                torch.cuda.synchronize()
                if rec_used:      # the brick update ran: everything the scatter wrote has been consumed
                    assert not bool(gb['buf'].any()) and not bool(gb['flags'].any()), it
        return [p.detach().clone() for p in model.parameters()], model._fused_cache, used
    finally:
        fused._BRICK_ADAM = old


def _spread(a, b):
    return max(float((pa - pb).norm() / pb.norm().clamp_min(1e-30)) for pa, pb in zip(a, b))


def test_fused_step_with_the_persistent_gradient_buffer_matches_the_plain_path(dev):
    """Same batches, same kernels up to the feature-grid gradient's home.  Given the same gradient the two updates are
    bit-identical (tests above); whole steps differ by the order of the scatter kernels' float atomics, which Adam's first
    steps (update ~ lr * sign(g)) amplify: the buffer path must stay within the run-to-run spread of the plain path."""
    a, cache_a, used_a = _train(dev, 4, brick=True)
    b, cache_b, used_b = _train(dev, 4, brick=False)
    c, _, _ = _train(dev, 4, brick=False)
    assert all(used_a) and cache_a['k0_grad']['clean'], (used_a, cache_a['k0_grad']['clean'])
    assert not any(used_b) and 'k0_grad' not in cache_b
    assert _spread(a, b) < max(3.0 * _spread(b, c), 1e-5) and _spread(a, b) < 2e-3, (_spread(a, b), _spread(b, c))


def test_dense_tv_on_the_feature_grid_falls_back_and_recovers(dev):
    """A dense TV term on k0 (model/nerf.py:341-344 k0_total_variation_add_grad, dense_mode) writes outside the recorded
    bricks: that step takes the dense masked update and leaves the buffer dirty; the next step zero-fills it and is back on
    the brick path.  Result equal to the plain path's."""
    a, cache_a, used_a = _train(dev, 4, brick=True, tv_dense_k0=True)
    b, _, _ = _train(dev, 4, brick=False, tv_dense_k0=True)
    c, _, _ = _train(dev, 4, brick=False, tv_dense_k0=True)
    assert all(used_a) and cache_a['k0_grad']['clean'], used_a
    assert _spread(a, b) < max(3.0 * _spread(b, c), 1e-5) and _spread(a, b) < 2e-3, (_spread(a, b), _spread(b, c))


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
