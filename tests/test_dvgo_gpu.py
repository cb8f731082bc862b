"""GPU parity of the density-grid (DVGO) variant, model/dvgo.py:284-357, against the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def test_dvgo_forward_backward_vs_oracle(dev, oracle):
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.dvgo import dvgo
    G, N = 32, 400
    torch.manual_seed(3)
    model = dvgo(xyz_min=[-1., -1., -1.], xyz_max=[1., 1., 1.], num_voxels=G ** 3, num_voxels_base=G ** 3, alpha_init=1e-2,
                 fast_color_thres=1e-4)
    assert model.world_size.tolist() == [G, G, G] and list(model.state_dict().keys())[:2] == ['xyz_min', 'xyz_max']
    assert {'density.grid', 'k0.grid'} <= set(model.state_dict().keys())
    sc = synth.scene_tensors(G, k0_dim=3, radius=0.5, sdf_noise=0.0)
    with torch.no_grad():
        model.density.grid.copy_(-40.0 * sc['sdf'] + 2.0)        # dense inside the ball, empty outside
        model.k0.grid.copy_(sc['k0'] * 10)
    model = model.to(dev)
    rays_c = synth.random_rays(N, seed=9)
    target_c = torch.rand(N, 3, generator=torch.Generator().manual_seed(5))
    kw = dict(synth.RENDER_KWARGS)
    res = model(*(r.to(dev) for r in rays_c), **kw)
    loss = F.mse_loss(res['rgb_marched'], target_c.to(dev)) + 1e-3 * res['alphainv_cum'].mean()
    loss.backward()

    P = dict(xyz_min=model.xyz_min.cpu(), xyz_max=model.xyz_max.cpu(), voxel_size=model.voxel_size.cpu(),
             voxel_size_ratio=model.voxel_size_ratio.cpu(), density=model.density.grid.detach().cpu().contiguous().clone(),
             k0=model.k0.grid.detach().cpu().contiguous().clone(), act_shift=model.act_shift, fast_color_thres=1e-4)
    P['density'].requires_grad_(True)
    P['k0'].requires_grad_(True)
    ref = oracle.dvgo_forward(P, *rays_c, near=2.0, stepsize=0.5, bg=1)
    lref = F.mse_loss(ref['rgb_marched'], target_c) + 1e-3 * ref['alphainv_cum'].mean()
    lref.backward()

    assert res['weights'].shape[0] > 500
    assert torch.equal(res['ray_id'].cpu(), ref['ray_id'])
    for key in ('rgb_marched', 'weights', 'raw_alpha', 'raw_rgb', 'alphainv_cum', 'normal_marched'):
        assert rel_l2(res[key], ref[key]) < 1e-5, key
    assert abs(float(loss) - float(lref)) < 1e-6
    assert rel_l2(model.density.grid.grad, P['density'].grad) < 1e-3
    assert rel_l2(model.k0.grid.grad, P['k0'].grad) < 1e-3
