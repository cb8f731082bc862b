"""world_size-2 gloo tests of the data-parallel layer (fgs-nerf_amd/dist.py): ray sharding and gradient averaging.

The GPU path uses the same code with backend "nccl" (RCCL over xGMI); the collective pattern -- one sum all-reduce per
large grid gradient, one bucket for the small MLP gradients -- is backend independent."""
import os
import socket

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
        # a "grid" (big, channel-last like DenseGrid storage) and two small "MLP" tensors, identical on all ranks
        grid = torch.nn.Parameter(torch.randn(1, 4, 6, 7, 8).contiguous(memory_format=torch.channels_last_3d))
        w = torch.nn.Parameter(torch.randn(5, 3))
        b = torch.nn.Parameter(torch.randn(5))
        n_total = 11
        x_all = torch.randn(n_total, 3)
        t_all = torch.randn(n_total, 5)
        idx_all = torch.randint(0, 4 * 6 * 7 * 8, (n_total,))

        def loss_of(x, t, idx):
            pred = x @ w.T + b + grid.reshape(-1)[idx][:, None]
            return (pred - t).pow(2).mean()

        sl = shard_rays(n_total, rank, world)
        n_local = sl.stop - sl.start
        # local mean loss scaled by n_local * world / n_total so that the rank-average equals the global mean
        loss = loss_of(x_all[sl], t_all[sl], idx_all[sl]) * (n_local * world / n_total)
        loss.backward()
        avg = GradAverager([grid, w, b], big_numel=512)      # the grid goes the "large message" way
        avg.average()
        # reference: the whole batch on one process
        g2 = torch.nn.Parameter(grid.detach().clone())
        w2, b2 = torch.nn.Parameter(w.detach().clone()), torch.nn.Parameter(b.detach().clone())
        pred = x_all @ w2.T + b2 + g2.reshape(-1)[idx_all][:, None]
        (pred - t_all).pow(2).mean().backward()
        ok = (torch.allclose(grid.grad, g2.grad, atol=1e-6) and torch.allclose(w.grad, w2.grad, atol=1e-6)
              and torch.allclose(b.grad, b2.grad, atol=1e-6) and grid.grad.stride() == grid.stride())
        # a voxel touched by only one rank stays non-zero after the sum (masked Adam keys on grad != 0)
        touched_union = (g2.grad != 0)
        ok = ok and bool(((grid.grad != 0) == touched_union).all())
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_grad_averager_world2_matches_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] and out[1]


def test_shard_rays_partitions_the_batch():
    from fgs_nerf_amd.dist import shard_rays
    for n, p in ((4096, 8), (10, 4), (7, 8), (32768, 8)):
        parts = [shard_rays(n, r, p) for r in range(p)]
        assert parts[0].start == 0 and parts[-1].stop == n
        assert all(parts[i].stop == parts[i + 1].start for i in range(p - 1))
        sizes = [s.stop - s.start for s in parts]
        assert max(sizes) - min(sizes) <= 1
