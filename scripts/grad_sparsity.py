"""Support of k0.grad / sdf.grad after one fine-stage step of the bench workload, at voxel, 2^3- and 4^3-brick granularity
(sizes the multi-GPU gradient exchange).  usage: python scripts/grad_sparsity.py"""
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from fgs_nerf_amd import synth  # noqa: E402
from fgs_nerf_amd.losses import fused_render_losses  # noqa: E402

dev = torch.device('cuda:0')
model = synth.build_model(bench.GRID, synth.FINE_MODEL, device=dev)
union = {}
for seed in range(8):                                    # 8 "ranks": different ray batches on the same scene
    rays_o, rays_d, viewdirs = (t.to(dev) for t in synth.random_rays(4096, seed=synth.SEED + seed))
    target = torch.rand(4096, 3, generator=torch.Generator().manual_seed(seed + 1)).to(dev)
    for p in model.parameters():
        p.grad = None
    res = model(rays_o, rays_d, viewdirs, global_step=1000, **synth.RENDER_KWARGS)
    fused_render_losses(res, target, synth.FINE_LOSS, model).backward()
    for name, g in (('k0', model.k0.grid.grad), ('sdf', model.sdf.grid.grad)):
        nz = (g[0] != 0).any(dim=0)                     # [X,Y,Z]
        X = nz.shape[0]
        b2 = nz.reshape(X // 2, 2, X // 2, 2, X // 2, 2).any(dim=5).any(dim=3).any(dim=1)
        b4 = nz.reshape(X // 4, 4, X // 4, 4, X // 4, 4).any(dim=5).any(dim=3).any(dim=1)
        u = union.setdefault(name, [torch.zeros_like(nz), torch.zeros_like(b2), torch.zeros_like(b4)])
        u[0] |= nz; u[1] |= b2; u[2] |= b4
        C = g.shape[1]
        if seed in (0, 7):
            print(f"{name} after {seed + 1} batch(es): this batch voxels {int(nz.sum())} ({int(nz.sum()) * C * 4 / 1e6:.1f} MB), "
                  f"2^3 bricks {int(b2.sum())} ({int(b2.sum()) * 8 * C * 4 / 1e6:.1f} MB), 4^3 bricks {int(b4.sum())} "
                  f"({int(b4.sum()) * 64 * C * 4 / 1e6:.1f} MB) | union voxels {int(u[0].sum())} "
                  f"({int(u[0].sum()) * C * 4 / 1e6:.1f} MB), 2^3 {int(u[1].sum())} ({int(u[1].sum()) * 8 * C * 4 / 1e6:.1f} MB), "
                  f"4^3 {int(u[2].sum())} ({int(u[2].sum()) * 64 * C * 4 / 1e6:.1f} MB); dense {g.numel() * 4 / 1e6:.1f} MB")
