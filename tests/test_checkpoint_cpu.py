"""Stage hand-off files (fgs_nerf_amd/checkpoint.py) against the behaviour of the reference's loaders
(model/utils.py:26-98, model/nerf_training.py:40-58,522-531): a weights-only round trip through save -> load_model /
load_checkpoint (with the fine stage's rescale) / load_weight_by_name / load_grid_data, and compute_bbox_by_coarse_geo against
the reference's dense-lattice expression."""
import os

import numpy as np
import pytest
import torch


def _model(G=12, seed=3, **kw):
    from fgs_nerf_amd import synth
    m = synth.build_model(G, synth.COARSE_MODEL, seed=seed, device='cpu', **kw)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    return m


def _save(tmp_path, model, name="coarse_last.tar", step=1234):
    import bench
    from fgs_nerf_amd import checkpoint
    opt = bench.make_optimizer(model)
    model.set_sdf_mask()
    path = os.path.join(tmp_path, name)
    checkpoint.save_checkpoint(path, model, opt, step)
    return path, opt


def test_save_then_load_model_restores_every_tensor(tmp_path):
    from fgs_nerf_amd import checkpoint, synth
    from fgs_nerf_amd.nerf import nerf
    model = _model()
    path, _ = _save(tmp_path, model)
    # get_kwargs (model/nerf.py:309-328) does not record the encoding sizes or the smoothing kernel: a caller whose config
    # differs from the constructor defaults passes them as new_kwargs, as with the reference
    extra = {k: synth.COARSE_MODEL[k] for k in ('posbase_pe', 'viewbase_pe', 'refbase_pe', 'smooth_ksize', 'smooth_sigma')}
    loaded, step = checkpoint.load_model(nerf, path, new_kwargs=extra, strict=False)
    assert step == 1234
    want = model.state_dict()
    got = loaded.state_dict()
    assert set(want) <= set(got) | {'sdf_mask.grid'}
    for k, v in want.items():
        if k in got:
            assert torch.equal(got[k], v), k
    # multi-channel grids keep this build's channel-last storage whatever layout the file holds
    assert loaded.k0.grid.stride() == model.k0.grid.stride()
    # new_kwargs override what the file says (model/utils.py:65-70)
    other, _ = checkpoint.load_model(nerf, path, new_kwargs=dict(extra, fast_color_thres=0.123), strict=False)
    assert other.fast_color_thres == 0.123


def test_load_weight_by_name_touches_only_the_named_parameters(tmp_path):
    from fgs_nerf_amd import checkpoint
    src, dst = _model(seed=3), _model(seed=9)
    path, _ = _save(tmp_path, src)
    before = {k: v.clone() for k, v in dst.named_parameters()}
    checkpoint.load_weight_by_name(dst, path, name='refnet')
    for k, v in dst.named_parameters():
        if 'refnet' in k:
            assert torch.equal(v, dict(src.named_parameters())[k]), k
        else:
            assert torch.equal(v, before[k]), k


def test_load_grid_data_raw_and_in_place(tmp_path):
    from fgs_nerf_amd import checkpoint
    src, dst = _model(seed=3), _model(seed=9)
    path, _ = _save(tmp_path, src)
    raw = checkpoint.load_grid_data(dst, path, name='sdf', return_raw=True)
    assert torch.equal(raw, src.sdf.grid.detach())
    checkpoint.load_grid_data(dst, path, name='k0')
    assert torch.equal(dst.k0.grid, src.k0.grid) and dst.k0.grid.stride() == src.k0.grid.stride()
    with pytest.raises(ValueError):
        checkpoint.load_grid_data(_model(G=16), path, name='sdf')       # another resolution


def test_load_checkpoint_resumes_and_rescales_for_the_fine_stage(tmp_path):
    """model/utils.py:42-60: stage='fine' loads the weights and rescales the grids to the new voxel budget; the optimizer
    state comes back as saved."""
    import bench
    from fgs_nerf_amd import checkpoint
    src = _model()
    path, opt = _save(tmp_path, src)
    # optimizer state in the file (MaskedAdam has no host form: the moments are filled by hand)
    opt.ensure_state()
    for st in opt.state.values():
        st['step'] = 5
        st['exp_avg'].normal_(0, 1e-3)
        st['exp_avg_sq'].uniform_(0, 1e-6)
    checkpoint.save_checkpoint(path, src, opt, 77)
    dst = _model(seed=11)
    opt2 = bench.make_optimizer(dst)
    budget = 20 ** 3
    dst, opt2, start = checkpoint.load_checkpoint(dst, opt2, path, no_reload_optimizer=False, stage='fine', num_voxels=budget,
                                                  strict=False)
    assert start == 77
    want = _model(seed=5)
    want.load_state_dict({k: v for k, v in src.state_dict().items()}, strict=False)
    want.scale_volume_grid(budget)
    assert tuple(dst.sdf.grid.shape) == tuple(want.sdf.grid.shape) != tuple(src.sdf.grid.shape)
    assert torch.equal(dst.sdf.grid, want.sdf.grid) and torch.equal(dst.k0.grid, want.k0.grid)
    for a, b in zip(src.refnet.parameters(), dst.refnet.parameters()):
        assert torch.equal(a, b)
    a, b = opt.state_dict()['state'], opt2.state_dict()['state']
    assert a.keys() == b.keys() and all(b[k]['step'] == 5 for k in b)
    # (moments of rescaled grids keep the old resolution, as in the reference: nerf_training.py builds a new optimizer
    # after a rescale; the MLP moments must be back exactly)
    small = [k for k in a if a[k]['exp_avg'].dim() <= 2]
    assert small and all(torch.equal(a[k]['exp_avg'], b[k]['exp_avg']) for k in small)


def test_compute_bbox_by_coarse_geo_equals_the_dense_lattice_expression(tmp_path):
    """model/nerf_training.py:40-58 written out: lattice = xyz_min * (1 - t) + xyz_max * t over a meshgrid of linspace(0,1,n),
    amin / amax over the voxels with sdf_mask > 0."""
    from fgs_nerf_amd import checkpoint
    model = _model(G=16)
    path, _ = _save(tmp_path, model)
    st = torch.load(path, weights_only=False)
    keep = torch.zeros_like(st['model_state_dict']['sdf_mask.grid'])
    keep[0, 0, 3:11, 2:9, 5:14] = 1                                   # a sub-box with a hole and a stray voxel
    keep[0, 0, 5:7, 4:6, 7:9] = 0
    keep[0, 0, 12, 1, 6] = 1
    st['model_state_dict']['sdf_mask.grid'] = st['model_state_dict']['sdf_mask.grid'] * keep
    torch.save(st, path)
    lo, hi = torch.tensor(st['model_kwargs']['xyz_min']), torch.tensor(st['model_kwargs']['xyz_max'])
    sm = st['model_state_dict']['sdf_mask.grid']
    t = torch.stack(torch.meshgrid(*[torch.linspace(0, 1, n) for n in sm.shape[2:]], indexing='ij'), -1)
    pts = (lo * (1 - t) + hi * t)[(sm > 0)[0, 0]]
    got_min, got_max = checkpoint.compute_bbox_by_coarse_geo(None, path, 0.0)
    assert torch.equal(got_min, pts.amin(0)) and torch.equal(got_max, pts.amax(0))
    assert bool((got_min > lo).all() and (got_max < hi).all())


def test_optimizer_state_written_in_the_reference_layout_is_relaid_on_load():
    """A reference-written checkpoint holds MaskedAdam moments NCDHW-contiguous (model/adam.py state); here k0 is channel-last
    and the update kernels walk parameter, gradient and both moments with ONE flat offset.  load_state_dict must re-lay the
    moments out like their parameter (same logical values), and a moment that does not share its parameter's layout must be
    refused by the kernels' host checks rather than silently paired with another voxel's elements."""
    import bench
    from fgs_nerf_amd.adam import MaskedAdam
    model = _model(G=8)
    opt = bench.make_optimizer(model)
    opt.ensure_state()
    k0 = model.k0.grid
    assert k0.stride() != k0.contiguous().stride()                      # channel-last storage
    sd = opt.state_dict()
    gen = torch.Generator().manual_seed(1)
    idx = [i for i, g in enumerate(opt.param_groups) if any(p is k0 for p in g['params'])][0]
    pid = sd['param_groups'][idx]['params'][0]
    m_ref = torch.randn(k0.shape, generator=gen).contiguous()            # what the reference would have saved: NCDHW-contiguous
    v_ref = torch.rand(k0.shape, generator=gen).contiguous()
    sd['state'][pid]['exp_avg'], sd['state'][pid]['exp_avg_sq'], sd['state'][pid]['step'] = m_ref, v_ref, 7
    opt2 = bench.make_optimizer(model)
    opt2.load_state_dict(sd)
    st = opt2.state[k0]
    assert st['step'] == 7
    assert st['exp_avg'].stride() == k0.stride() and st['exp_avg_sq'].stride() == k0.stride()
    assert torch.equal(st['exp_avg'], m_ref) and torch.equal(st['exp_avg_sq'], v_ref)       # same logical values
    # a moment forced into another layout is refused by the host check the kernels sit behind
    st['exp_avg'] = m_ref.clone()
    with pytest.raises(RuntimeError, match="memory layout"):
        MaskedAdam._check_layout(k0, st)
