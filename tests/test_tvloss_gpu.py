"""The autograd-form total-variation losses of the ori_tv configurations (model/nerf.py:430-459,1212-1221;
model/dvgo.py:206-215,420-428) as HIP value + gradient passes (csrc/tvloss.hip, dense.grid_tv_loss) against the oracle's
statement of the reference expression evaluated in float64.  The value is a sum of |differences| (float accumulation per
thread, double across threads): <= 1e-6 relative.  The gradient of every element is a sum of at most six signs times one
float factor: <= 2e-6 of the largest entry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grid(shape, layout, dev, seed):
    g = torch.Generator().manual_seed(seed)
    v = torch.rand(shape, generator=g) + 0.05
    v[0, :, 2:4, 3:5, 1:6] = 0.25                 # flat patches: sign(0) = 0 must match torch's abs backward
    v = v.to(dev)
    if layout == "last":
        v = v.contiguous(memory_format=torch.channels_last_3d)
    return v.requires_grad_(True)


@pytest.mark.parametrize("variant", ["nerf", "dvgo"])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("shape,layout", [((1, 1, 13, 17, 19), "first"), ((1, 12, 9, 11, 14), "last"), ((1, 4, 8, 8, 8), "first")])
def test_grid_tv_loss_value_and_gradient(dev, oracle, variant, masked, shape, layout):
    from fgs_nerf_amd import dense
    v = _grid(shape, layout, dev, seed=sum(shape))
    mask = None
    if masked:
        m = torch.rand((1, 1) + shape[2:], generator=torch.Generator().manual_seed(5)) > 0.3
        mask = m.to(dev).expand(1, shape[1], -1, -1, -1) if variant == "nerf" else m.to(dev)
        if variant == "dvgo" and shape[1] > 1:
            pytest.skip("model/dvgo.py indexes a C-channel difference with a 1-channel mask: the reference itself raises")
    tv = dense.grid_tv_loss(v, mask, per_axis_mean=(variant == "dvgo"))
    (tv * 0.7).backward()
    v64 = v.detach().double().cpu().contiguous().requires_grad_(True)
    m64 = None if mask is None else mask.cpu().contiguous()
    ref = oracle.total_variation(v64, m64, variant=variant)
    (ref * 0.7).backward()
    assert abs(float(tv) - float(ref)) <= 1e-6 * abs(float(ref)), (float(tv), float(ref))
    g, gr = v.grad.cpu().double(), v64.grad
    assert float((g - gr).abs().max()) <= 2e-6 * float(gr.abs().max()), float((g - gr).abs().max())
    assert v.grad.stride() == v.stride()


@pytest.mark.parametrize("masked", [False, True])
def test_model_tv_losses_match_the_reference_expression(dev, oracle, masked):
    """density_total_variation(sdf_tv > 0) and k0_total_variation through the model API (what nerf_training.py:329-351 adds
    to the loss under ori_tv), values and gradients w.r.t. both grids."""
    from fgs_nerf_amd import synth
    model = synth.build_model(32, synth.FINE_MODEL, device=dev)
    with torch.no_grad():
        model.sdf.grid.add_(torch.randn_like(model.sdf.grid) * 0.01)
        model.k0.grid.add_(torch.randn_like(model.k0.grid) * 0.1)
    if masked:
        m = torch.rand(model.sdf.grid.shape, generator=torch.Generator().manual_seed(1)) > 0.4
        model.nonempty_mask = m.to(dev)
    else:
        model.nonempty_mask = None
    loss = model.density_total_variation(sdf_tv=0.1, smooth_grad_tv=0) + 0.5 * model.k0_total_variation()
    for p in (model.sdf.grid, model.k0.grid):
        p.grad = None
    loss.backward()
    sdf64 = model.sdf.grid.detach().double().cpu().contiguous().requires_grad_(True)
    k064 = model.k0.grid.detach().double().cpu().contiguous().requires_grad_(True)
    m64 = None if not masked else model.nonempty_mask.cpu()
    ref = (oracle.total_variation(sdf64, m64) / 2 / float(model.voxel_size) * 0.1 +
           0.5 * oracle.total_variation(k064, None if m64 is None else m64.repeat(1, k064.shape[1], 1, 1, 1)))
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)), (float(loss), float(ref))
    for p, r in ((model.sdf.grid, sdf64), (model.k0.grid, k064)):
        d = float((p.grad.cpu().double() - r.grad).abs().max())
        assert d <= 3e-6 * float(r.grad.abs().max()), d


def test_grid_tv_loss_full_size_feature_grid(dev):
    """160^3 x 12 channels (the bench's feature grid), masked: finite, positive, gradient supported inside the mask only."""
    from fgs_nerf_amd import dense
    v = (torch.rand(1, 12, 160, 160, 160, device=dev) + 0.1).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    mask = (torch.rand(1, 1, 160, 160, 160, device=dev) > 0.5)
    tv = dense.grid_tv_loss(v, mask.expand(1, 12, -1, -1, -1))
    tv.backward()
    assert torch.isfinite(tv) and float(tv) > 0
    assert not bool((v.grad != 0)[~mask.expand_as(v.grad)].any())


@pytest.mark.parametrize("masked", [False, True])
def test_weighted_terms_added_to_a_loss_in_the_launches(dev, oracle, masked):
    """density_total_variation(..., weight=, add_to=) -- the form nerf_training uses: the launches scale the terms and add the loss
    so far themselves -- against weight * (reference expression) + loss in float64, values and gradients, with a registered unit
    seed (no grid-sized multiply in the backward pass) and with an arbitrary one; the scalars are fixed-order sums: two runs agree
    bit for bit."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import register_unit_seed
    model = synth.build_model(48, synth.COARSE_MODEL, device=dev)
    with torch.no_grad():
        model.sdf.grid.add_(torch.randn_like(model.sdf.grid) * 0.01)
    model.nonempty_mask = (torch.rand(model.sdf.grid.shape, generator=torch.Generator().manual_seed(2)) > 0.4).to(dev) if masked else None
    base_p = torch.tensor(0.37, device=dev, requires_grad=True)
    w = 0.01

    def run(seed):
        model.sdf.grid.grad = None
        base_p.grad = None
        loss = base_p * 1.0
        model.gradient = model.neus_sdf_gradient(sdf=model.sdf.grid)       # (the forward pass leaves it: an autograd node over sdf.grid)
        loss = model.density_total_variation(sdf_tv=0, smooth_grad_tv=0.05, weight=w, add_to=loss)
        loss = model.density_total_variation(sdf_tv=0.1, smooth_grad_tv=0, weight=w, add_to=loss)
        loss.backward(seed)
        return loss.detach().clone(), model.sdf.grid.grad.detach().clone(), base_p.grad.detach().clone()

    unit = register_unit_seed(torch.ones((), device=dev))
    la, ga, ba = run(unit)
    lb, gb, bb = run(unit)
    assert torch.equal(la, lb) and torch.equal(ga, gb)
    lc, gc, bc = run(torch.full((), 0.25, device=dev))
    assert float(ba) == 1.0 and float(bc) == 0.25
    assert float((gc * 4.0 - ga).abs().max()) <= 1e-6 * float(ga.abs().max())
    # the reference expressions (model/nerf.py:430-447), unfused, through the same model: torch ops around the two launches
    model.sdf.grid.grad = None
    model.gradient = model.neus_sdf_gradient(sdf=model.sdf.grid)
    ref = 0.37 + w * model.density_total_variation(sdf_tv=0, smooth_grad_tv=0.05) + w * model.density_total_variation(sdf_tv=0.1, smooth_grad_tv=0)
    ref.backward()
    gr = model.sdf.grid.grad
    assert abs(float(la) - float(ref)) <= 2e-6 * abs(float(ref)), (float(la), float(ref))
    assert float((ga - gr).abs().max()) <= 3e-6 * float(gr.abs().max())
    # ... and the sdf TV part against the oracle's float64 statement
    sdf64 = model.sdf.grid.detach().double().cpu().contiguous()
    m64 = None if not masked else model.nonempty_mask.cpu()
    tv64 = float(oracle.total_variation(sdf64, m64)) / 2 / float(model.voxel_size) * 0.1
    only = model.density_total_variation(sdf_tv=0.1, smooth_grad_tv=0, weight=w, add_to=None)
    assert abs(float(only) - w * tv64) <= 2e-6 * abs(w * tv64)
