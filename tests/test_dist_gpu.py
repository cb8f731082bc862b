"""Single-rank RCCL ("nccl" backend) smoke test of the gradient exchange on the GPU: process-group init as bench.py does it,
the brick-sparse path on a real 160^3 k0 gradient (world size 1: the exchange must leave the gradient unchanged)."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_sparse_exchange_on_real_gradient(dev):
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.dist import GradAverager
    from fgs_nerf_amd.losses import fused_render_losses
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        model = synth.build_model(160, synth.FINE_MODEL, device=dev)
        rays = tuple(r.to(dev) for r in synth.random_rays(4096))
        target = torch.rand(4096, 3, device=dev)
        res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
        fused_render_losses(res, target, synth.FINE_LOSS, model).backward()
        g = model.k0.grid.grad
        before = g.clone()
        avg = GradAverager(model.parameters(), force=True)   # run the exchange in a group of one: it must be the identity
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert avg._sparse(g, 1.0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.equal(g, before)
        assert 0 < avg.last_sparse_fill < 0.5, avg.last_sparse_fill
        print(f"sparse exchange: fill {avg.last_sparse_fill:.3f}, {dt * 1e3:.2f} ms single-rank")
        # the full driver (dense + bucket + sparse, ncclAvg inside the collectives) is the identity at one rank
        sdf_before = model.sdf.grid.grad.clone()
        w_before = [p.grad.clone() for p in model.refnet.parameters()]
        avg.average()
        assert torch.equal(model.sdf.grid.grad, sdf_before) and torch.equal(g, before)
        assert all(torch.equal(p.grad, w) for p, w in zip(model.refnet.parameters(), w_before))
    finally:
        dist.destroy_process_group()


def test_hinted_exchange_matches_gradient_occupancy(dev):
    """hint_touched (occupancy from the survivor points, on a side stream, count fetched asynchronously) must cover every
    non-zero brick of the real k0 gradient and leave the gradient unchanged at one rank."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.dist import GradAverager
    from fgs_nerf_amd.losses import fused_render_losses
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for stage, cfg, lossw in (("fine", synth.FINE_MODEL, synth.FINE_LOSS), ("coarse", synth.COARSE_MODEL, synth.COARSE_LOSS)):
            model = synth.build_model(64, cfg, device=dev)
            rays = tuple(r.to(dev) for r in synth.random_rays(2048, seed=5))
            target = torch.rand(2048, 3, device=dev)
            avg = GradAverager(model.parameters(), force=True, sparse_min_numel=1 << 16)
            res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
            avg.hint_touched(model.k0.grid, res['survivor_pts'], model.xyz_min, model.xyz_max)
            fused_render_losses(res, target, lossw, model).backward()
            g = model.k0.grid.grad
            before = g.clone()
            # occupancy read from the gradient itself (the un-hinted path's kernel)
            from fgs_nerf_amd._lib import call, ptr, stream
            C, X, Y, Z = g.shape[1:]
            flags = torch.empty((X // 4) * (Y // 4) * (Z // 4), dtype=torch.int32, device=dev)
            call("fgs_brick_flags", ptr(g), C, X, Y, Z, ptr(flags), stream())
            h = avg._hints[id(model.k0.grid)]
            assert h['armed']
            avg.average()
            assert not h['armed']
            hinted = torch.zeros_like(flags)
            hinted[h['idx'][:int(h['count_host'][0])]] = 1
            assert bool(((flags == 1) & (hinted == 0)).sum() == 0), stage        # superset of the non-zero bricks
            assert int(hinted.sum()) <= 2 * int(flags.sum()) + 8, stage           # and not much more
            assert torch.equal(g, before), stage                                  # world size 1: SUM and 1/P are identities
            assert 0 < avg.last_sparse_fill < 0.5
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, out_q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)                       # both ranks share the one GPU of the box
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fgs_nerf_amd import synth
        from fgs_nerf_amd.dist import GradAverager, shard_rays
        from fgs_nerf_amd.losses import fused_render_losses
        model = synth.build_model(64, synth.FINE_MODEL, device=dev)
        ro, rd, vd = synth.random_rays(2048, seed=8)
        sl = shard_rays(2048, rank, world)
        rays = tuple(t[sl].to(dev) for t in (ro, rd, vd))
        target = torch.rand(2048, 3, generator=torch.Generator().manual_seed(2))[sl].to(dev)
        # reference: the local gradients of this rank, from an identical model with no exchange attached
        twin = synth.build_model(64, synth.FINE_MODEL, device=dev)
        fused_render_losses(twin(*rays, global_step=1000, **synth.RENDER_KWARGS), target, synth.FINE_LOSS, twin).backward()
        avg = GradAverager(model.parameters(), sparse_min_numel=1 << 16)
        avg.attach(model)                                # k0 and the MLP gradients are exchanged from inside the backward pass
        res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
        avg.hint_touched(model.k0.grid, res['survivor_pts'], model.xyz_min, model.xyz_max)
        fused_render_losses(res, target, synth.FINE_LOSS, model).backward()
        params = [p for p in model.parameters() if p.grad is not None]
        local = [p.grad.detach().clone() for p in twin.parameters() if p.grad is not None]
        assert len(local) == len(params)
        avg.average()
        worst = 0.0
        for p, g in zip(params, local):                  # dense reference: plain all-reduce of the local copies
            ref = g.contiguous().cpu()
            dist.all_reduce(ref)
            ref = ref / world
            got = p.grad.detach().cpu()
            worst = max(worst, float((got - ref).norm() / ref.norm().clamp_min(1e-30)))
            assert torch.equal(got != 0, ref != 0) or p.grad.dim() != 5     # the union of touched voxels is preserved
        fill = avg.last_sparse_fill
        # second step, with the optimizer attached: average() no longer waits for the k0 exchange, MaskedAdam does when it
        # reaches k0 (before_param).  The twin takes the same step with plainly all-reduced gradients.
        import bench
        opt, opt_twin = bench.make_optimizer(model), bench.make_optimizer(twin)
        avg.attach_optimizer(opt)
        from fgs_nerf_amd import fused
        fused.enable_early_update(model, opt, avg)       # k0's Adam pass right behind its exchange, on the exchange stream
        assert avg.after_early is not None
        before, worst2 = [p.detach().clone() for p in model.parameters()], 0.0
        assert opt.before_param is not None and avg.defer_to_optimizer
        for m_ in (model, twin):
            for p in m_.parameters():
                p.grad = None
        res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
        avg.hint_touched(model.k0.grid, res['survivor_pts'], model.xyz_min, model.xyz_max)
        fused_render_losses(res, target, synth.FINE_LOSS, model).backward()
        avg.average()
        opt.step()
        assert not avg._deferred and not opt._early                 # nothing left pending: k0 was updated early, step() waited
        fused_render_losses(twin(*rays, global_step=1000, **synth.RENDER_KWARGS), target, synth.FINE_LOSS, twin).backward()
        for p in twin.parameters():
            if p.grad is not None:
                g = p.grad.detach().contiguous().cpu()
                dist.all_reduce(g)
                p.grad.copy_((g / world).to(dev).view_as(p.grad))
        opt_twin.step()
        # Adam turns a gradient into ~lr * sign(g) on its first step, so the few elements whose tiny gradient differs in
        # the last bits between two runs of the atomics-based kernels move differently: compare the updates in norm
        for p, q, p0 in zip(model.parameters(), twin.parameters(), before):
            dq = q.detach() - p0
            if float(dq.norm()) > 0:
                worst2 = max(worst2, float(((p.detach() - p0) - dq).norm() / dq.norm()))
        out_q.put((rank, worst, fill, worst2))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_a_dense_all_reduce(dev):
    """Two ranks (gloo, both on the box's single GPU) with different ray shards: the hinted brick-sparse exchange of
    k0.grad, the dense sdf exchange and the MLP bucket must equal a plain dense all-reduce of the local gradients."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    results = sorted(q.get(timeout=10) for _ in range(2))
    for rank, worst, fill, worst2 in results:
        assert worst2 < 1e-2, results                   # parameter updates of the step with the optimizer-side wait
        assert worst < 1e-6, results
        assert fill is not None and 0 < fill < 0.6, results


@pytest.mark.parametrize("total", [0, 1, 1023, 1024, 1025, 5000, 64000])
def test_brick_compact_matches_nonzero(dev, total):
    """fgs_brick_compact (one workgroup per 1024-flag tile, offsets counted from the flags in front of the tile): ascending
    indices of the set flags and their count, entirely on the device."""
    from fgs_nerf_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(total)
    flags = (torch.rand(max(total, 1), generator=g) < 0.18).to(torch.int32)[:total].to(dev)
    idx = torch.full((max(total, 1),), -1, dtype=torch.int64, device=dev)
    count = torch.full((1,), -7, dtype=torch.int64, device=dev)
    call("fgs_brick_compact", ptr(flags) if total else None, total, ptr(idx) if total else None, ptr(count), stream())
    ref = flags.nonzero(as_tuple=False).squeeze(1)
    assert int(count) == ref.numel()
    assert torch.equal(idx[:ref.numel()], ref)


def _edge_worker(rank, world, port, out_q, mode):
    """Two gloo ranks on the box's one GPU, rank-ASYMMETRIC situations that used to desynchronise the collective sequence:
      'empty_rank'  every ray of rank 1 misses the volume (M == 0 there): its backward must still issue the in-backward
                    exchanges the other rank issues (fused._backward_empty), and receive rank 0's gradients;
      'small_grid'  a grid below `sparse_min_numel` with hint_touched called every step and different survivor counts per
                    rank and step: the hint must be ignored on shape grounds alone (no collective issued by one rank only)."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fgs_nerf_amd import synth
        from fgs_nerf_amd.dist import GradAverager
        from fgs_nerf_amd.losses import fused_render_losses
        if mode == 'empty_rank':
            G, cfg, lossw, kw = 32, synth.FINE_MODEL, synth.FINE_LOSS, dict(sparse_min_numel=1 << 16)
        else:
            G, cfg, lossw, kw = 16, synth.COARSE_MODEL, synth.COARSE_LOSS, {}
        model = synth.build_model(G, cfg, device=dev)
        twin = synth.build_model(G, cfg, device=dev)
        avg = GradAverager(model.parameters(), **kw)
        avg.attach(model)
        worst, n_surv = 0.0, []
        for step in range(3):
            n = 256 if mode == 'empty_rank' else (200 + 90 * ((rank + step) % 3))
            ro, rd, vd = synth.random_rays(n, seed=40 + 7 * rank + step)
            if mode == 'empty_rank' and rank == 1:
                rd, vd = -rd, -vd                          # looking away from the box: no sample at all
            rays = tuple(t.to(dev) for t in (ro, rd, vd))
            target = torch.rand(n, 3, generator=torch.Generator().manual_seed(step)).to(dev)
            for m_ in (model, twin):
                for p in m_.parameters():
                    p.grad = None
            fused_render_losses(twin(*rays, global_step=1000, **synth.RENDER_KWARGS), target, lossw, twin).backward()
            res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
            n_surv.append(int(res['weights'].shape[0]))
            avg.hint_touched(model.k0.grid, res['survivor_pts'], model.xyz_min, model.xyz_max)
            fused_render_losses(res, target, lossw, model).backward()
            avg.average()
            for p, q in zip(model.parameters(), twin.parameters()):
                if q.grad is None:
                    continue
                ref = q.grad.detach().contiguous().cpu()
                dist.all_reduce(ref)
                ref = ref / world
                got = p.grad.detach().cpu()
                if float(ref.norm()) > 0:
                    worst = max(worst, float((got - ref).norm() / ref.norm()))
                else:
                    assert float(got.norm()) == 0
        out_q.put((rank, worst, n_surv, len(avg._hints)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["empty_rank", "small_grid"])
def test_rank_asymmetric_steps_keep_the_collectives_matched(dev, mode):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_edge_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, "a rank hung or failed (mismatched collectives?)"
    results = sorted(q.get(timeout=10) for _ in range(2))
    for rank, worst, n_surv, n_hints in results:
        assert worst < 1e-6, results
    if mode == 'empty_rank':
        assert all(n == 0 for n in results[1][2]) and all(n > 0 for n in results[0][2]), results
    else:
        assert all(r[3] == 0 for r in results), results            # the hint never armed anything on the small grid
        assert results[0][2] != results[1][2]                      # different survivor histories on the two ranks


def test_sync_free_eager_steps_with_the_exchange_match_plain_steps(dev):
    """What `bench.py` runs on several GPUs: eager launches with the survivor count left on the device (fused.set_sync_free)
    and the in-backward gradient exchange (a real RCCL group of one) -- against the same steps with host-sized tensors.
    Same survivor totals, parameters equal in norm; the hinted brick occupancy is computed under the device-side count."""
    import bench
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.dist import GradAverager
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        N, STEPS = 1024, 4
        batches = []
        for b in range(2):
            ro, rd, vd = synth.random_rays(N, seed=40 + b)
            batches.append(tuple(t.to(dev).contiguous() for t in (ro, rd, vd, torch.rand(N, 3, generator=torch.Generator().manual_seed(b)))))
        outs = {}
        for mode in ("plain", "sync_free"):
            model = synth.build_model(64, synth.FINE_MODEL, device=dev)
            opt = bench.make_optimizer(model)
            avg = GradAverager(model.parameters(), force=True, sparse_min_numel=1 << 16)
            avg.attach(model)
            avg.attach_optimizer(opt)
            bench.STEP_STATS.update(survivors=0, max_survivors=0)
            if mode == "sync_free":
                fused.set_sync_free(model, 32768)
            for i in range(STEPS):
                bench.train_step(model, opt, avg, batches[i % 2], N)
            torch.cuda.synchronize()
            if mode == "sync_free":
                overflow, total = fused.sync_free_state(model)
                assert not overflow
                fused.set_sync_free(model, None)
            else:
                total = bench.STEP_STATS["survivors"]
            assert avg.last_sparse_fill is not None and 0 < avg.last_sparse_fill < 0.6      # the sparse exchange ran
            outs[mode] = (total, [p.detach().clone() for p in model.parameters()])
        # (the two trainings see the same batches, but their gradients are summed with float atomics: after a few Adam steps a
        # sample sitting on the alpha threshold may fall on either side -- 29 644 vs 29 645 survivors has been seen)
        assert outs["sync_free"][0] > 0 and abs(outs["plain"][0] - outs["sync_free"][0]) <= 8
        for pa, pb in zip(outs["plain"][1], outs["sync_free"][1]):
            assert float((pa - pb).norm() / pa.norm().clamp_min(1e-30)) < 3e-3
    finally:
        dist.destroy_process_group()


def _train(model, opt, avg, batches, steps, n_rays, captured_capacity=None, exchange_capacity=None):
    """`steps` training iterations over `batches`: eager sync-free steps with the host-counted exchange (bench.train_step), or --
    with `captured_capacity` -- one hipGraph replay each with the exchange captured inside (device-counted k0 exchange)."""
    import bench
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.graph_step import CapturedFineStep
    if captured_capacity is None:
        fused.set_sync_free(model, 32768)
        opt.use_skip_flag(model._fused_cache['sync_free']['flags'][1:2].data_ptr())
        for i in range(steps):
            bench.train_step(model, opt, avg, batches[i % len(batches)], n_rays)
        torch.cuda.synchronize()
        state = fused.sync_free_state(model)
        fused.set_sync_free(model, None)
        opt.use_skip_flag(None)
        return state, None
    cap = CapturedFineStep(model, opt, synth.FINE_LOSS, synth.RENDER_KWARGS, n_rays, n_iters=steps + 4,
                           global_step_of=lambda it: bench.GLOBAL_STEP, lr_of=lambda it, g: g['lr'],
                           tv=(0.01 * 0.1 / n_rays, True), capacity=captured_capacity, averager=avg,
                           exchange_capacity=exchange_capacity)
    cap.capture(batches[0])
    for i in range(steps):
        cap.replay(batches[i % len(batches)])
    torch.cuda.synchronize()
    return cap.check(), cap


def test_captured_step_with_exchange_matches_eager_exchange(dev):
    """The multi-GPU iteration as ONE graph replay: RCCL collectives captured as graph nodes (a real RCCL group of one), the k0
    brick exchange in its device-counted form (fixed-capacity buffer, union count on the device), k0's Adam pass issued
    behind it on the exchange stream -- against the same iterations issued eagerly with the host-counted exchange.  Then the
    overflow contract: an exchange buffer that cannot hold the union raises the sticky flag, and from that replay on no
    parameter changes any more."""
    import bench
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.dist import GradAverager
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    caps = []
    try:
        N, STEPS = 1024, 5
        batches = []
        for b in range(2):
            ro, rd, vd = synth.random_rays(N, seed=60 + b)
            batches.append(tuple(t.to(dev).contiguous() for t in (ro, rd, vd, torch.rand(N, 3, generator=torch.Generator().manual_seed(b)))))
        outs = {}
        for mode in ("eager", "captured"):
            model = synth.build_model(64, synth.FINE_MODEL, device=dev)
            opt = bench.make_optimizer(model)
            # (captured: the 1-channel sdf gradient goes brick-sparse too, device-counted, its capacity sized by capture())
            avg = GradAverager(model.parameters(), force=True, sparse_min_numel=1 << 16,
                               sparse_1ch_min_numel=(1 << 16) if mode == "captured" else None)
            avg.attach(model)
            avg.attach_optimizer(opt)
            fused.enable_early_update(model, opt, avg)
            bench.STEP_STATS.update(survivors=0, max_survivors=0)
            (overflow, total), cap = _train(model, opt, avg, batches, STEPS, N, captured_capacity=32768 if mode == "captured" else None,
                                            exchange_capacity=2048 if mode == "captured" else None)
            caps.append(cap)
            assert not overflow and total > 0, mode
            if cap is not None:
                assert not cap.exchange_overflowed()
                assert cap.sdf_exchange_capacity is not None and 0 < cap.sdf_exchange_capacity <= 16 ** 3
                assert all(opt.state[p]['step'] == STEPS for g in opt.param_groups for p in g['params'])
            outs[mode] = (total, [p.detach().clone() for p in model.parameters()])
        assert abs(outs["eager"][0] - outs["captured"][0]) <= 8
        for pa, pb in zip(outs["eager"][1], outs["captured"][1]):
            assert float((pa - pb).norm() / pa.norm().clamp_min(1e-30)) < 3e-3
        # overflow: 16 bricks cannot hold the union of a 1024-ray batch
        model = synth.build_model(64, synth.FINE_MODEL, device=dev)
        opt = bench.make_optimizer(model)
        avg = GradAverager(model.parameters(), force=True, sparse_min_numel=1 << 16)
        avg.attach(model)
        avg.attach_optimizer(opt)
        fused.enable_early_update(model, opt, avg)
        trained = [p for g in opt.param_groups for p in g['params']]      # (model.s_val follows the schedule regardless)
        before = [p.detach().clone() for p in trained]
        (overflow, _), cap = _train(model, opt, avg, batches, 3, N, captured_capacity=32768, exchange_capacity=16)
        caps.append(cap)
        assert overflow and cap.exchange_overflowed()
        assert all(torch.equal(p.detach(), b) for p, b in zip(trained, before))     # every update was skipped
    finally:
        for c in caps:            # graphs holding RCCL nodes go before the communicator does (CapturedFineStep.release)
            if c is not None:
                c.release()
        dist.destroy_process_group()
