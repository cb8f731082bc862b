"""The HIP kernels against golden vectors produced by the REFERENCE'S OWN pure-torch functions (tests/golden/ref_fns.npz: see
oracle/make_golden_ref_fns.py and tests/test_ref_pins_cpu.py).  These rows of SURVEY 8a are thereby pinned to reference-executed
arithmetic, not only to this repository's restatement: a4 mask cache, a5 trilinear lookup, a6 gradient volume and Gaussian
smoothing, a9 compositing weights (without early stop), a11 normal, f2 both total_variation variants and the training loop's TV
terms, the orientation loss."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref(golden):
    g = golden("ref_fns.npz")
    return {k: torch.from_numpy(g[k]) for k in g.files if g[k].dtype.kind in "fbiu"}


@pytest.mark.parametrize("C", [1, 3, 12])
@pytest.mark.parametrize("layout", ["reference", "channel_last"])
def test_trilerp_kernel_matches_reference_lookup(dev, ref, C, layout):
    """fgs_trilerp_fwd == DenseGrid.forward / grid_sampler of the reference (model/grid.py:49-68, model/nerf.py:639-672) on the
    same grid and points (inside, on the faces, outside: zero padding).  fp32 tolerance 1e-6 rel-L2 (the kernel and
    F.grid_sample order the eight products differently); exact zeros outside the box."""
    from fgs_nerf_amd.grid import DenseGrid, to_grid_layout
    grid = ref[f"tri_grid_{C}"].to(dev)
    g = DenseGrid(C, grid.shape[2:], ref["tri_lo"], ref["tri_hi"])
    g.grid.data = to_grid_layout(grid) if layout == "channel_last" else grid.contiguous()
    out = g(ref["tri_pts"].to(dev)).detach().cpu()
    want = ref[f"tri_dense_{C}"]
    assert rel_l2(out, want) < 1e-6
    assert torch.equal(out == 0, want == 0)


def test_gradient_volume_kernel_matches_reference(dev, ref):
    """fgs_sdf_gradvol_fwd == nerf.neus_sdf_gradient('interpolate') (model/nerf.py:485-494): same two float operations per
    element (difference, / 2, / voxel_size) -> bit-exact."""
    from fgs_nerf_amd import dense
    out = dense.sdf_gradient_volume(ref["gv_sdf"].to(dev), float(ref["gv_voxel_size"]))
    assert torch.equal(out.cpu(), ref["gv_interpolate"])


def test_raw_and_grad_conv_gradient_volumes_match_reference(dev, ref):
    """The reference's two non-default gradient volumes (model/nerf.py:495-506): 'raw' (forward difference: fgs_sdf_gradvol_fwd
    mode 1, bit-exact) and 'grad_conv' (its own Sobel-like weight, model/nerf.py:224-247: three fgs_smooth3d_fwd passes);
    their adjoints against torch autograd of the same expressions."""
    import torch.nn.functional as F
    from fgs_nerf_amd import dense
    sdf, vs = ref["gv_sdf"].to(dev), float(ref["gv_voxel_size"])
    assert torch.equal(dense.sdf_gradient_volume(sdf, vs, mode='raw').cpu(), ref["gv_raw"])
    w = ref["gradconv_w_05"].to(dev)             # the reference's grad_conv weight the fixture's volume was made with (sigma = 0.5)
    got = dense.sdf_gradient_volume(sdf, vs, mode='grad_conv', grad_conv_weight=w)
    assert rel_l2(got.cpu(), ref["gv_grad_conv"]) < 1e-6
    up = torch.randn(1, 3, *sdf.shape[2:], generator=torch.Generator().manual_seed(4)).to(dev)
    for mode in ('raw', 'grad_conv'):
        a = sdf.clone().requires_grad_(True)
        dense.sdf_gradient_volume(a, vs, mode=mode, grad_conv_weight=w).backward(up)
        b = sdf.clone().requires_grad_(True)
        if mode == 'raw':
            g = torch.zeros(1, 3, *sdf.shape[2:], device=dev)
            g[:, 0, :-1] = (b[:, 0, 1:] - b[:, 0, :-1]) / vs
            g[:, 1, :, :-1] = (b[:, 0, :, 1:] - b[:, 0, :, :-1]) / vs
            g[:, 2, :, :, :-1] = (b[:, 0, :, :, 1:] - b[:, 0, :, :, :-1]) / vs
        else:
            g = F.conv3d(F.pad(b, (1,) * 6, mode='replicate'), w)
        g.backward(up)
        assert rel_l2(a.grad, b.grad) < 2e-6, mode


@pytest.mark.parametrize("ks", [3, 5])
def test_smoothing_kernel_matches_reference_conv(dev, ref, ks):
    """fgs_smooth3d_fwd == the reference's replicate-padded Conv3d with its own Gaussian taps (model/nerf.py:260-272)."""
    from fgs_nerf_amd import dense
    out = dense.smooth3d(ref["gv_sdf"].to(dev), ref[f"smooth_w_{ks}"].to(dev))
    assert rel_l2(out.cpu(), ref[f"smooth_out_{ks}"]) < 1e-6


def test_mask_cache_matches_reference(dev, ref):
    """nerf.MaskCache (max-pool + fgs_trilerp_fwd + threshold) == MaskCache.forward of the reference (model/nerf.py:1193-1209)."""
    from fgs_nerf_amd.nerf import MaskCache
    mc = MaskCache(mask_cache_thres=1e-3, sdf_mask=ref["mc_raw"], xyz_min=ref["tri_lo"].numpy(), xyz_max=ref["tri_hi"].numpy()).to(dev)
    keep = mc(ref["mc_pts"].to(dev)).cpu()
    want = ref["mc_keep"].bool()
    # a lookup that lands within float rounding of the threshold may fall on either side: none does in this fixture
    assert torch.equal(keep, want)


def test_alpha2weight_kernel_matches_cumprod_compositing(dev, ref):
    """fgs_alpha2weight_fwd == get_ray_marching_ray of the reference (model/dvgo.py:409-417) on rays that never reach T < 1e-3;
    1e-6 relative (torch's CPU cumprod accumulates in double, see tests/test_ref_pins_cpu.py)."""
    from fgs_nerf_amd.render import Alphas2Weights
    alpha = ref["crm_alpha"]
    n_rays, n_s = alpha.shape
    ray_id = torch.arange(n_rays).repeat_interleave(n_s).to(dev)
    w, last = Alphas2Weights.apply(alpha.reshape(-1).contiguous().to(dev), ray_id, n_rays)
    np.testing.assert_allclose(w.reshape(n_rays, n_s).cpu().numpy(), ref["crm_weights"].numpy(), rtol=1e-6, atol=0)
    np.testing.assert_allclose(last.cpu().numpy(), ref["crm_alphainv_cum"][:, -1].numpy(), rtol=1e-6, atol=0)


def test_alpha2weight_kernel_early_stop_prefix_matches_cumprod_compositing(dev, golden):
    """fgs_alpha2weight_fwd on rays that DO reach T < 1e-3 against the reference's cumprod form executed on the same alphas
    (tests/test_ref_pins_cpu.py::test_alpha2weight_early_stop_prefix_matches_cumprod_compositing has the argument): prefix equal,
    zeros behind the stop, alphainv_last = T at the stop, i_end behind the stop sample."""
    from test_ref_pins_cpu import _early_stop_expectation
    from fgs_nerf_amd.ops import render_utils_cuda
    alpha, want_w, want_last, want_end, _ = _early_stop_expectation(golden)
    n_rays, n_s = alpha.shape
    ray_id = torch.arange(n_rays).repeat_interleave(n_s).to(dev)
    w, T, last, i_start, i_end = render_utils_cuda.alpha2weight(torch.from_numpy(alpha).reshape(-1).contiguous().to(dev), ray_id, n_rays)
    assert np.array_equal(i_end.cpu().numpy(), want_end) and np.array_equal(i_start.cpu().numpy(), np.arange(n_rays) * n_s)
    w = w.reshape(n_rays, n_s).cpu().numpy()
    np.testing.assert_allclose(w, want_w, rtol=1e-6, atol=0)
    assert np.all(w[want_w == 0] == 0)
    np.testing.assert_allclose(last.cpu().numpy(), want_last, rtol=1e-6, atol=0)


@pytest.mark.parametrize("variant", ["nerf", "dvgo"])
def test_tv_loss_kernels_match_reference_total_variation(dev, ref, variant):
    """csrc/tvloss.hip (dense.grid_tv_loss) == total_variation of model/nerf.py:1212-1221 and model/dvgo.py:420-428, with and
    without mask; 2e-6 relative (tree reduction vs torch's sum order)."""
    from fgs_nerf_amd import dense
    from fgs_nerf_amd.grid import to_grid_layout
    for v, m, tag in ((ref["tv_v1"], ref["tv_m1"], "v1"), (ref["tv_v12"], ref["tv_m12"], "v12")):
        vg = to_grid_layout(v.to(dev))
        got = dense.grid_tv_loss(vg, None, per_axis_mean=variant == "dvgo")
        assert abs(float(got) - float(ref[f"tv_{variant}_{tag}"])) <= 2e-6 * abs(float(ref[f"tv_{variant}_{tag}"]))
        got = dense.grid_tv_loss(vg, m.bool().to(dev), per_axis_mean=variant == "dvgo")
        assert abs(float(got) - float(ref[f"tv_{variant}_{tag}_m"])) <= 2e-6 * abs(float(ref[f"tv_{variant}_{tag}_m"]))


def test_training_tv_terms_match_reference(dev, ref):
    """nerf.density_total_variation / k0_total_variation (model/nerf.py:430-459) through the HIP passes: the sdf TV term and the
    smooth-gradient TV term, with and without nonempty_mask, on the reference's own inputs."""
    from fgs_nerf_amd import synth
    model = synth.build_model(16, synth.FINE_MODEL, device=dev)
    sdf, v12 = ref["gv_sdf"].to(dev), ref["tv_v12"].to(dev)
    from fgs_nerf_amd.grid import to_grid_layout
    model.sdf.grid = torch.nn.Parameter(sdf.contiguous())
    model.k0.grid = torch.nn.Parameter(to_grid_layout(v12))
    model.voxel_size = ref["gv_voxel_size"].to(dev)
    model.init_gradient_conv(sigma=0)
    model.gradient = ref["gv_interpolate"].to(dev).contiguous()
    for tag, mask in (("nomask", None), ("mask", ref["tv_m1"].bool().to(dev))):
        model.nonempty_mask = mask
        model.__dict__.pop('_nonempty_count', None)
        a = float(model.density_total_variation(sdf_tv=0.1, smooth_grad_tv=0).detach())
        b = float(model.density_total_variation(sdf_tv=0, smooth_grad_tv=0.05).detach())
        assert abs(a - float(ref[f"dtv_sdf_{tag}"])) <= 3e-6 * abs(float(ref[f"dtv_sdf_{tag}"])), tag
        assert abs(b - float(ref[f"dtv_smooth_{tag}"])) <= 3e-6 * abs(float(ref[f"dtv_smooth_{tag}"])), tag
    model.nonempty_mask = None
    assert abs(float(model.k0_total_variation()) - float(ref["ktv_nomask"])) <= 3e-6 * abs(float(ref["ktv_nomask"]))
    model.nonempty_mask = ref["tv_m12"][:, :1].bool().to(dev)
    assert abs(float(model.k0_total_variation()) - float(ref["ktv_mask"])) <= 3e-6 * abs(float(ref["ktv_mask"]))


def test_normal_and_orientation_loss_match_reference(dev, ref):
    """render.l2_normalize == nerf.l2_normalize (model/nerf.py:480-483, incl. the zero vector); the orientation term of the HIP
    loss kernels (fgs_fine_loss_fwd) == nerf.orientation_loss (model/nerf.py:469-478) on per-sample lists of one ray each."""
    from fgs_nerf_amd.render import l2_normalize
    got = l2_normalize(ref["l2n_x"].to(dev)).cpu()           # (torch on the device: sum / sqrt may round differently from the CPU's)
    assert rel_l2(got, ref["l2n_out"]) < 1e-6 and torch.equal(got[7], ref["l2n_out"][7])       # row 7: the zero vector
    from fgs_nerf_amd.losses import fused_render_losses
    M = ref["ori_weights"].shape[0]                  # one sample per ray: the per-sample view direction is the ray's
    z3 = torch.zeros(M, 3, device=dev)
    res = dict(rgb_marched=z3, sigmoid_rgb=z3, alphainv_cum=torch.zeros(M, device=dev), raw_rgb=z3, weights=ref["ori_weights"].to(dev),
               normal=ref["ori_normal"].to(dev), ray_viewdirs=ref["ori_viewdirs"].to(dev),
               ray_id=torch.arange(M, dtype=torch.int64, device=dev))
    loss = fused_render_losses(res, z3, dict(weight_main=0.0, weight_orientation=1.0))
    assert abs(float(loss) - float(ref["ori_loss"])) <= 2e-6 * abs(float(ref["ori_loss"]))


def test_sdf_tap_kernels_match_reference_sample_sdfs(dev, golden):
    """fgs_sdf_taps_fwd (render.sample_sdfs) == nerf.sample_sdfs of the reference (model/nerf.py:597-637, executed with
    Tensor.cuda as the identity: tests/golden/ref_fns_cuda_shim.npz): 6 K clamped taps and 3 K (normalised) differences for
    K = 4 and K = 1, points on the faces and outside the box included."""
    from fgs_nerf_amd.render import sample_sdfs
    g = golden("ref_fns_cuda_shim.npz")
    T = lambda k: torch.from_numpy(g[k]).to(dev)
    for tag, norm in (("k4", True), ("k4_raw", False), ("k1", False)):
        feat, grad = sample_sdfs(T("taps_pts"), T("taps_grid"), T("taps_lo"), T("taps_hi"), float(g["taps_voxel_size"]),
                                 g[f"taps_disp_{tag}"].tolist(), use_grad_norm=norm)
        assert rel_l2(feat.cpu(), torch.from_numpy(g[f"taps_feat_{tag}"])) < 1e-6, tag
        assert rel_l2(grad.cpu(), torch.from_numpy(g[f"taps_grad_{tag}"])) < 5e-6, tag
