"""GPU tests of the SURVEY 8(f) "next" rows built so far: lazily materialised result-dict masks (f1), SDF field
extraction for mesh export (f3), stage hand-off checkpoint -> MaskCache (f4), per-voxel view counts."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def test_lazy_mask_entries_match_reference_semantics(dev):
    from fgs_nerf_amd import synth
    rays = tuple(r.to(dev) for r in synth.random_rays(400, seed=17))
    a = synth.build_model(40, synth.FINE_MODEL, device=dev, fused=True)
    b = synth.build_model(40, synth.FINE_MODEL, device=dev, fused=False)
    with torch.no_grad():
        ra = a(*rays, global_step=1000, **synth.RENDER_KWARGS)
        rb = b(*rays, global_step=1000, **synth.RENDER_KWARGS)
    assert 'mask' in ra and 'mask_outbbox' in ra                     # present as keys before being computed
    assert torch.equal(ra['mask'], rb['mask'])                        # weights > thres over the alpha-compacted list
    assert torch.equal(ra['mask_outbbox'], rb['mask_outbbox'])
    assert abs(float(ra['mask'].float().mean()) - float(rb['mask'].float().mean())) == 0.0   # the logged statistic


def test_extract_fields_matches_grid_sample(dev, oracle):
    from fgs_nerf_amd import synth
    m = synth.build_model(32, synth.FINE_MODEL, device=dev)
    lo, hi = torch.tensor([-1., -1., -1.]), torch.tensor([1., 1., 1.])
    u = m.extract_fields(lo, hi, resolution=70)                       # crosses the 64^3 block boundary
    assert u.shape == (70, 70, 70) and u.dtype == np.float32
    xs = torch.linspace(-1, 1, 70)
    pts = torch.stack(torch.meshgrid(xs, xs, xs, indexing='ij'), -1).reshape(-1, 3)
    ref = oracle.dense_grid_forward(-m.sdf.grid.detach().cpu(), pts, lo, hi).reshape(70, 70, 70)
    assert rel_l2(u, ref) < 1e-6
    verts, tris = m.extract_geometry(lo, hi, resolution=70)            # device marching cubes (tests/test_mcubes.py)
    assert verts.dtype == np.float64 and verts.shape[1] == 3 and tris.shape[1] == 3 and len(tris) > 0
    assert verts.min() >= -1.0 and verts.max() <= 1.0 and int(tris.max()) == len(verts) - 1


def test_stage_checkpoint_to_mask_cache_roundtrip(dev, tmp_path):
    """model/nerf_training.py:522-531 -> model/nerf.py:1192-1200: `set_sdf_mask`, save, next stage builds its MaskCache."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.nerf import nerf
    coarse = synth.build_model(24, synth.COARSE_MODEL, device=dev, fused=False)
    coarse.set_sdf_mask()
    path = os.path.join(tmp_path, "coarse_last.tar")
    torch.save({'global_step': 7, 'model_kwargs': coarse.get_kwargs(), 'MaskCache_kwargs': coarse.get_MaskCache_kwargs(),
                'model_state_dict': coarse.state_dict()}, path)
    assert 'sdf_mask.grid' in coarse.state_dict()
    fine = nerf(xyz_min=[-1., -1., -1.], xyz_max=[1., 1., 1.], num_voxels=32 ** 3, num_voxels_base=32 ** 3,
                mask_cache_path=path, mask_cache_thres=1e-3, **synth.FINE_MODEL).to(dev)
    assert fine.mask_cache is not None and fine.nonempty_mask.shape == fine.sdf.grid.shape
    # the cache keeps the neighbourhood of the coarse surface (sdf < 0.5) and drops the far corners
    inside = fine.mask_cache(torch.tensor([[0.0, 0.0, 0.55]], device=dev))
    corner = fine.mask_cache(torch.tensor([[0.98, 0.98, 0.98]], device=dev))
    assert bool(inside[0]) and not bool(corner[0])
    rays = tuple(r.to(dev) for r in synth.random_rays(128, seed=2))
    res = fine(*rays, global_step=10, **synth.RENDER_KWARGS)          # fused path with the mask cache active
    assert res['rgb_marched'].shape == (128, 3) and torch.isfinite(res['rgb_marched']).all()
    # the reference's compute_bbox_by_coarse_geo (nerf_training.py:40-58) reads the same file
    st = torch.load(path, weights_only=False)
    assert (st['model_state_dict']['sdf_mask.grid'] > 0).any()


@pytest.mark.parametrize("stage", ["fine", "coarse"])
def test_training_steps_do_not_leak_device_memory(dev, stage):
    """The fused autograd nodes must not keep their own outputs alive (output -> grad_fn -> ctx -> output is a cycle the
    Python collector cannot see): device memory after step 6 and after step 30 must agree."""
    import gc
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import fused_render_losses
    cfg, lossw = (synth.FINE_MODEL, synth.FINE_LOSS) if stage == "fine" else (synth.COARSE_MODEL, synth.COARSE_LOSS)
    model = synth.build_model(48, cfg, device=dev)
    rays = tuple(r.to(dev) for r in synth.random_rays(1024, seed=3))
    target = torch.rand(1024, 3, device=dev)
    gc.disable()
    try:
        marks = {}
        for i in range(31):
            for p in model.parameters():
                p.grad = None
            res = model(*rays, global_step=500, **synth.RENDER_KWARGS)
            fused_render_losses(res, target, lossw, model).backward()
            del res
            if i in (6, 30):
                torch.cuda.synchronize()
                marks[i] = torch.cuda.memory_allocated()
    finally:
        gc.enable()
    assert marks[30] - marks[6] < (8 << 20), marks


@pytest.mark.parametrize("masked", [False, True])
def test_density_total_variation_smooth_term(dev, oracle, masked):
    """density_total_variation(smooth_grad_tv) of the fine stage (model/nerf.py:430-447; the shipped fine config adds it
    every third iteration): lazily built gradient volume + HIP 3^3 smoothing + masked mean, value and sdf gradient
    against the reference expression written out in torch on the CPU."""
    import torch.nn.functional as F
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.nerf import MaskCache
    model = synth.build_model(40, synth.FINE_MODEL, device=dev)
    with torch.no_grad():
        model.sdf.grid += 0.05 * torch.randn(model.sdf.grid.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    if masked:
        sdf_mask = ((model.sdf.grid.detach() < 0.25) * 1e-3).float()
        model.mask_cache = MaskCache(path=None, mask_cache_thres=1e-3 * 0.5, sdf_mask=sdf_mask.cpu(),
                                     xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1]).to(dev)
        model._set_nonempty_mask()
        assert 0.05 < float(model.nonempty_mask.float().mean()) < 0.95
    rays = tuple(r.to(dev) for r in synth.random_rays(64, seed=2))
    model(*rays, global_step=10, **synth.RENDER_KWARGS)          # forward_fine marks model.gradient pending (:856)
    tv = model.density_total_variation(sdf_tv=0, smooth_grad_tv=0.05)
    model.sdf.grid.grad = None
    tv.backward()
    s = model.sdf.grid.detach().cpu().clone().requires_grad_(True)
    gv = oracle.neus_sdf_gradient(s, model.voxel_size.cpu()).permute(1, 0, 2, 3, 4)
    sm = F.conv3d(F.pad(gv, (1,) * 6, mode='replicate'), model.tv_smooth_conv.weight.detach().cpu())
    err = sm.detach() - gv
    if masked:
        err = err[model.nonempty_mask.cpu().repeat(3, 1, 1, 1, 1)]
    ref = (err ** 2).mean() * 0.05
    ref.backward()
    assert abs(float(tv) - float(ref)) < 1e-5 * abs(float(ref))
    assert rel_l2(model.sdf.grid.grad, s.grad) < 1e-5


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("stage", ["fine", "coarse"])
def test_degenerate_batches(dev, stage, fused):
    """Edge cases a training run meets: a batch whose rays all miss the volume (no sample survives: every per-sample
    list is empty, the pixels are background, backward still runs), a single ray, and an eval-mode call (global_step=None)."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import fused_render_losses
    cfg, lossw = (synth.FINE_MODEL, synth.FINE_LOSS) if stage == "fine" else (synth.COARSE_MODEL, synth.COARSE_LOSS)
    model = synth.build_model(32, cfg, device=dev, fused=None if fused else False)
    n = 37
    ro = torch.tensor([[0.0, 0.0, 4.0]], device=dev).repeat(n, 1)
    rd = torch.tensor([[0.0, 0.0, 1.0]], device=dev).repeat(n, 1)          # looking away from the [-1,1]^3 box
    vd = rd.clone()
    target = torch.rand(n, 3, device=dev)
    res = model(ro, rd, vd, global_step=100, **synth.RENDER_KWARGS)
    assert res['weights'].numel() == 0 and res['ray_id'].numel() == 0 and res['raw_rgb'].shape == (0, 3)
    assert torch.allclose(res['rgb_marched'], torch.ones(n, 3, device=dev))          # bg = 1
    assert torch.allclose(res['alphainv_cum'], torch.ones(n, device=dev))
    for p in model.parameters():
        p.grad = None
    fused_render_losses(res, target, lossw, model).backward()
    assert float(model.k0.grid.grad.abs().sum()) == 0.0 and float(model.sdf.grid.grad.abs().sum()) == 0.0
    assert res['mask'] is None or res['mask'].numel() == 0 or not bool(res['mask'].any())
    # one ray through the centre
    res1 = model(torch.tensor([[0.0, 0.0, 4.0]], device=dev), torch.tensor([[0.0, 0.0, -1.0]], device=dev),
                 torch.tensor([[0.0, 0.0, -1.0]], device=dev), global_step=100, **synth.RENDER_KWARGS)
    assert res1['rgb_marched'].shape == (1, 3) and res1['weights'].numel() > 0
    assert float(res1['weights'].sum() + res1['alphainv_cum'][0]) <= 1.0 + 1e-5
    # eval mode: no s_val update, no autograd
    with torch.no_grad():
        rays = tuple(r.to(dev) for r in synth.random_rays(200, seed=4))
        a = model(*rays, global_step=None, **synth.RENDER_KWARGS)
        b = model(*rays, global_step=None, **synth.RENDER_KWARGS)
    # the fused compositing is a fixed-order per-ray reduction; the operator form sums with atomics (index_add_)
    same = torch.equal if fused else (lambda x, y: torch.allclose(x, y, atol=1e-6))
    assert same(a['rgb_marched'], b['rgb_marched']) and bool(torch.isfinite(a['rgb_marched']).all())


def test_maskcache_ray_prefilter_on_the_device_matches_reference_fixture(dev, golden):
    """nerf_ray.get_training_rays_in_maskcache_sampling with the PRODUCT model (sample_ray_ori in torch on the device,
    MaskCache through the HIP trilinear kernel) against the fixture the reference's own model/nerf_ray.py:208-250 produced
    (oracle/make_golden.py make_maskcache_rays): same kept rays, same order, same values."""
    import numpy as np
    from fgs_nerf_amd import nerf_ray, synth
    from fgs_nerf_amd.nerf import MaskCache
    g = golden("maskcache_rays.npz")
    G = int(g["G"])
    model = synth.build_model(G, synth.FINE_MODEL, device=dev)
    model.mask_cache = MaskCache(path=None, mask_cache_thres=float(g["thres"]), sdf_mask=torch.from_numpy(g["sdf_mask"]),
                                 xyz_min=g["xyz_min"], xyz_max=g["xyz_max"]).to(dev)
    assert torch.equal(model.voxel_size.cpu(), torch.from_numpy(g["voxel_size"]))
    images = [torch.from_numpy(g[f"image{i}"]).to(dev) for i in range(3)]
    rgb_tr, ro, rd, vd, imsz = nerf_ray.get_training_rays_in_maskcache_sampling(
        images, torch.from_numpy(g["poses"]), g["HW"], g["Ks"], False, False, False, False, model,
        dict(near=float(g["near"]), far=float(g["far"]), stepsize=float(g["stepsize"])))
    assert [int(n) for n in imsz] == g["imsz"].tolist()
    assert ro.is_cuda and rgb_tr.is_cuda
    for got, key in ((rgb_tr, "rgb_tr"), (ro, "rays_o_tr"), (rd, "rays_d_tr"), (vd, "viewdirs_tr")):
        assert np.array_equal(got.cpu().numpy(), g[key]), key


def test_batch_gather_in_one_launch(dev):
    """fgs_gather_batch == the four advanced-indexing gathers of model/nerf_training.py:256-261 (bit-exact: copies), through
    CapturedFineStep.load_selected's argument order (rays_o, rays_d, viewdirs, target -> inputs[0..3])."""
    import torch
    from fgs_nerf_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(4)
    R, n = 50000, 8192
    srcs = [torch.randn(R, 3, generator=g).to(dev) for _ in range(4)]
    sel = torch.randperm(R, generator=g)[:n].to(dev)
    out = torch.full((4, n, 3), float('nan'), device=dev)
    call("fgs_gather_batch", ptr(sel), n, R, *(ptr(t) for t in srcs), ptr(out), stream())
    torch.cuda.synchronize()
    for a in range(4):
        assert torch.equal(out[a], srcs[a][sel])


def test_staged_batch_copy_kernel(dev):
    """fgs_copy_f32 (CapturedFineStep.load of a staged [4, n, 3] batch): a copy, bit-exact, tail elements included."""
    import torch
    from fgs_nerf_amd._lib import call, ptr, stream
    for n in (4 * 4096 * 3, 1027, 3):
        src = torch.randn(n + 8, device=dev)[:n]
        dst = torch.full((n + 4,), float('nan'), device=dev)
        call("fgs_copy_f32", ptr(src), ptr(dst), n, stream())
        torch.cuda.synchronize()
        assert torch.equal(dst[:n], src) and bool(torch.isnan(dst[n:]).all())
