"""world_size-2 gloo tests of the data-parallel layer (fgs-nerf_amd/dist.py): ray sharding and gradient averaging,
including the brick-sparse exchange of the multi-channel grid gradient.

The GPU path uses the same code with backend "nccl" (RCCL over xGMI); the collective pattern is backend independent."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fgs_nerf_amd.dist import GradAverager, shard_rays
        torch.manual_seed(0)
        # identical replicas: a channel-last feature grid (brick-sparse path), an odd-sized 1-channel grid (dense path)
        # and two small "MLP" tensors (bucket path)
        k0 = torch.nn.Parameter(torch.randn(1, 4, 8, 12, 16).contiguous(memory_format=torch.channels_last_3d))
        sdf = torch.nn.Parameter(torch.randn(1, 1, 6, 7, 9))
        w = torch.nn.Parameter(torch.randn(5, 3))
        b = torch.nn.Parameter(torch.randn(5))
        n_total = 11
        x_all = torch.randn(n_total, 3)
        t_all = torch.randn(n_total, 5)
        vox = torch.randint(0, 8 * 12 * 16, (n_total,))          # each "ray" touches one k0 voxel (4 channels) ...
        vs = torch.randint(0, 6 * 7 * 9, (n_total,))              # ... and one sdf voxel

        def loss_of(k0_, sdf_, w_, b_, x, t, v, s):
            feat = k0_[0].permute(1, 2, 3, 0).reshape(-1, 4)[v]                       # [n,4]
            pred = x @ w_.T + b_ + feat.sum(-1, keepdim=True) + sdf_.reshape(-1)[s][:, None]
            return (pred - t).pow(2).mean()

        sl = shard_rays(n_total, rank, world)
        n_local = sl.stop - sl.start
        # local mean loss scaled by n_local * world / n_total so that the rank-average equals the global mean
        (loss_of(k0, sdf, w, b, x_all[sl], t_all[sl], vox[sl], vs[sl]) * (n_local * world / n_total)).backward()
        avg = GradAverager([k0, sdf, w, b], big_numel=256, sparse_min_numel=1024, sparse_max_fill=0.9)
        avg.average()
        used_sparse = avg.last_sparse_fill is not None and 0 < avg.last_sparse_fill < 0.9
        # reference: the whole batch on one process
        ref = [torch.nn.Parameter(p.detach().clone()) for p in (k0, sdf, w, b)]
        loss_of(*ref, x_all, t_all, vox, vs).backward()
        ok = all(torch.allclose(p.grad, r.grad, atol=1e-6) for p, r in zip((k0, sdf, w, b), ref))
        ok = ok and k0.grad.stride() == k0.stride() and used_sparse
        # a voxel touched by only one rank stays non-zero after the exchange (masked Adam keys on grad != 0)
        ok = ok and bool(((k0.grad != 0) == (ref[0].grad != 0)).all())
        # dense fallback when the occupancy is above the limit gives the same numbers
        k0.grad = None
        sdf.grad = None
        w.grad = None
        b.grad = None
        (loss_of(k0, sdf, w, b, x_all[sl], t_all[sl], vox[sl], vs[sl]) * (n_local * world / n_total)).backward()
        GradAverager([k0, sdf, w, b], big_numel=256, sparse_min_numel=1024, sparse_max_fill=0.0).average()
        ok = ok and torch.allclose(k0.grad, ref[0].grad, atol=1e-6)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_grad_averager_world2_matches_single_process(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world))


def test_shard_rays_partitions_the_batch():
    from fgs_nerf_amd.dist import shard_rays
    for n, p in ((4096, 8), (10, 4), (7, 8), (32768, 8)):
        parts = [shard_rays(n, r, p) for r in range(p)]
        assert parts[0].start == 0 and parts[-1].stop == n
        assert all(parts[i].stop == parts[i + 1].start for i in range(p - 1))
        sizes = [s.stop - s.start for s in parts]
        assert max(sizes) - min(sizes) <= 1


def _counted_worker(rank, world, port, out):
    """The device-counted form of the brick exchange (GradAverager.use_device_counts), on host tensors: fixed-capacity buffer,
    the union's count never used to size anything, overflow decided from the all-reduced count.  The two ranks touch
    DIFFERENT voxels in every step (different histories); the schedule makes step 2 overflow the capacity."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fgs_nerf_amd.dist import GradAverager
        k0 = torch.nn.Parameter(torch.zeros(1, 4, 8, 12, 16).contiguous(memory_format=torch.channels_last_3d))
        n_bricks = 2 * 3 * 4
        guard = torch.zeros(2, dtype=torch.int32)              # [0] sticky "something overflowed", [1] "skip this step"
        avg = GradAverager([k0], big_numel=256, sparse_min_numel=1024)
        avg.use_device_counts(k0, capacity=6, guard_flags=guard)
        assert avg._static[id(k0)]['buf'].shape == (6, 64 * 4)
        touched_per_step = [2, 3, 6, 1, 2]                     # voxels per rank; bricks in the union: <= 2x that
        log = []
        ok = True
        for step, n_vox in enumerate(touched_per_step):
            gen = torch.Generator().manual_seed(100 * step + rank)
            g = torch.zeros_like(k0)
            flat = g[0].permute(1, 2, 3, 0).reshape(-1, 4)
            # each voxel in its own brick (one voxel per brick index), rank-dependent choice
            bricks = torch.randperm(n_bricks, generator=gen)[:n_vox]
            for b in bricks.tolist():
                bx, by, bz = b // 12, (b // 4) % 3, b % 4
                flat[((bx * 4) * 12 + by * 4) * 16 + bz * 4] = torch.randn(4, generator=gen)
            k0.grad = g
            local = g.detach().clone()
            guard[1] = 1 if (step == 1 and rank == 1) else 0   # rank 1's own survivor list "overflowed" in step 1
            avg.average()
            ref = local.clone()
            dist.all_reduce(ref)
            ref /= world
            union = int(((ref != 0).reshape(1, 4, 2, 4, 3, 4, 4, 4).any(7).any(5).any(3).any(1)).sum())
            same = torch.allclose(k0.grad, ref, atol=1e-7)
            log.append((step, union, int(guard[0]), int(guard[1]), bool(same)))
        out[rank] = log
    finally:
        dist.destroy_process_group()


def test_device_counted_exchange_world2_overflow_protocol():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_counted_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert [r[:4] for r in a] == [r[:4] for r in b], (a, b)       # both ranks: same union counts, same flags, every step
    by_step = {r[0]: r for r in a}
    assert by_step[0][1] <= 6 and by_step[0][2:] == (0, 0, True)   # fits: equal to the dense all-reduce, nothing raised
    assert by_step[1][3] == 1 and by_step[1][4]                     # rank 1's skip flag reached rank 0 as well; values still exact
    assert by_step[2][1] > 6 and by_step[2][2:4] == (1, 1)         # the union did not fit: sticky + skip on both ranks
    assert by_step[3][1] <= 6 and by_step[3][2:4] == (1, 1)        # ... and it STAYS raised although this union fits
    assert all(r[4] for r in a if r[0] in (0, 1)) and all(r[4] for r in b if r[0] in (0, 1))


def _sdf_sparse_worker(rank, world, port, out):
    """The 1-channel sdf gradient through the brick-sparse exchange (host-counted and device-counted forms) == dense all-reduce."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fgs_nerf_amd.dist import GradAverager
        ok = True
        for counted in (False, True):
            sdf = torch.nn.Parameter(torch.zeros(1, 1, 8, 12, 16))
            gen = torch.Generator().manual_seed(7 + rank)
            g = torch.zeros_like(sdf)
            pick = torch.randperm(8 * 12 * 16, generator=gen)[:40]              # a sparse, rank-dependent set of voxels
            g.view(-1)[pick] = torch.randn(40, generator=gen)
            sdf.grad = g.clone()
            avg = GradAverager([sdf], big_numel=256, sparse_1ch_min_numel=1024, sparse_1ch_eager=True)
            if counted:
                avg.use_device_counts(sdf, capacity=24)
            avg.average()
            ref = g.clone()
            dist.all_reduce(ref)
            ref /= world
            ok = ok and torch.allclose(sdf.grad, ref, atol=1e-7) and bool(((sdf.grad != 0) == (ref != 0)).all())
            if counted:
                ok = ok and not avg.device_count_state(sdf)
            else:
                ok = ok and avg.last_sparse_fill_1ch is not None and 0 < avg.last_sparse_fill_1ch <= 1.0
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sdf_gradient_brick_sparse_exchange_world2(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sdf_sparse_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world))


def _spread_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        same = [torch.tensor(3.25, dtype=torch.float64), torch.tensor(1e6, dtype=torch.float64)]
        drift = [torch.tensor(3.25, dtype=torch.float64), torch.tensor(1e6 + rank, dtype=torch.float64)]
        out[rank] = (bench.replica_digest_spread(same), bench.replica_digest_spread(drift))
    finally:
        dist.destroy_process_group()


def test_bench_replica_check_sees_drift():
    """bench.py's end-of-run check that the ranks' parameters stayed identical (config.replicas_in_sync): 0.0 for equal digests,
    the relative spread otherwise, the same number on every rank."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_spread_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == out[1]
    assert out[0][0] == 0.0 and abs(out[0][1] - 1.0 / (1e6 + 1)) < 1e-12


def _tune_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fgs_nerf_amd.dist import GradAverager
        sdf = torch.nn.Parameter(torch.zeros(1, 1, 32, 32, 32))
        av = GradAverager([sdf], sparse_1ch_min_numel=1 << 24)
        # an absurd overhead figure forces "dense" on every rank, a negative one "sparse": the verdict is rank 0's either way
        r_dense = av.tune_sparse_1ch(sdf, kernel_overhead_us=1e9, reps=3)
        thr_dense = av.sparse_1ch_min_numel
        r_sparse = av.tune_sparse_1ch(sdf, kernel_overhead_us=-1e9, reps=3)
        thr_sparse = av.sparse_1ch_min_numel
        out[rank] = (r_dense['tuned'], r_dense['sparse'], thr_dense > sdf.numel(), r_sparse['sparse'], thr_sparse <= sdf.numel(),
                     av._sparse_1ch(sdf.detach()), r_sparse['buffer_bricks'])
    finally:
        dist.destroy_process_group()


def test_sdf_exchange_form_is_tuned_identically_on_every_rank():
    """GradAverager.tune_sparse_1ch: times the dense and the brick-sparse exchange of the sdf gradient with the group's own
    collectives and moves the shape threshold -- the same decision on every rank (rank 0's verdict is broadcast)."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_tune_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == out[1] == (True, False, True, True, True, True, out[0][6]) and 0 < out[0][6] <= 512
