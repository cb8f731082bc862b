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
        avg = GradAverager(model.parameters())
        avg.world_size = 2                       # force the exchange code path; with one rank SUM is the identity
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert avg._sparse(g, 1.0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.equal(g, before)
        assert 0 < avg.last_sparse_fill < 0.5, avg.last_sparse_fill
        print(f"sparse exchange: fill {avg.last_sparse_fill:.3f}, {dt * 1e3:.2f} ms single-rank")
        # the full driver, dense + bucket + sparse, is the identity at one rank up to the 1/P factor
        sdf_before = model.sdf.grid.grad.clone()
        avg.average()
        assert torch.allclose(model.sdf.grid.grad, sdf_before * 0.5) and torch.allclose(g, before * 0.5)
    finally:
        dist.destroy_process_group()
