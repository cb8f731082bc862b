"""The oracle against golden vectors produced by the REFERENCE'S OWN pure-torch functions (tests/golden/ref_fns.npz, written by
oracle/make_golden_ref_fns.py: single function definitions taken out of model/nerf.py, model/dvgo.py and model/grid.py with
`ast` and executed on fixed inputs -- the modules themselves cannot be imported).  Every comparison here pins a
piece of oracle/oracle.py (or the C restatement behind it) to reference-executed arithmetic; tests/test_ref_pins_gpu.py
pins the HIP kernels to the same vectors."""
import numpy as np
import pytest
import torch

from conftest import rel_l2


@pytest.fixture(scope="module")
def ref(golden):
    g = golden("ref_fns.npz")
    return {k: (torch.from_numpy(g[k]) if g[k].dtype.kind in "fbiu" else g[k]) for k in g.files}


def test_fixture_names_the_reference_lines_it_executed(ref):
    names = {s.split(":")[0] for s in ref["source_lines"].tolist()}
    assert {"tv_nerf", "tv_dvgo", "cumprod_exclusive", "get_ray_marching_ray", "neus_sdf_gradient", "_gaussian_3dconv",
            "grid_sampler", "DenseGrid.forward", "MaskCache.forward", "sample_ray_ori", "l2_normalize", "orientation_loss",
            "density_total_variation", "k0_total_variation", "init_gradient_conv"} <= names


@pytest.mark.parametrize("variant", ["nerf", "dvgo"])
def test_total_variation_matches_the_reference(oracle, ref, variant):
    """model/nerf.py:1212-1221 and model/dvgo.py:420-428, with and without mask, 1 and 12 channels."""
    for v, m, tag in ((ref["tv_v1"], ref["tv_m1"], "v1"), (ref["tv_v12"], ref["tv_m12"], "v12")):
        assert torch.equal(oracle.total_variation(v, None, variant), ref[f"tv_{variant}_{tag}"])
        assert torch.equal(oracle.total_variation(v, m.bool(), variant), ref[f"tv_{variant}_{tag}_m"])


def test_alpha2weight_matches_cumprod_compositing_without_early_stop(oracle, ref):
    """model/dvgo.py:409-417 (cumprod_exclusive / get_ray_marching_ray): on rays that never reach T < 1e-3 the C restatement of
    alpha2weight (render_utils_kernel.cu:576-605: the sequential product, early stop never taken) must give the same weights and
    the same final transmittance.  Tolerance 1e-6 relative: torch's CPU cumprod accumulates a float32 row in DOUBLE
    (at::acc_type), the kernel's running product T *= (1 - alpha) is float32 -- 23 factors differ by up to 3 ulp (3.6e-7 seen)."""
    alpha = ref["crm_alpha"]
    n_rays, n_s = alpha.shape
    ray_id = torch.arange(n_rays).repeat_interleave(n_s)
    w, last, i_end = oracle.alphas2weights(alpha.reshape(-1).contiguous(), ray_id, n_rays)
    assert float(ref["crm_alphainv_cum"].min()) > 1e-3                       # the premise: no early termination
    assert np.array_equal(np.asarray(i_end), (np.arange(n_rays) + 1) * n_s)   # every ray ran to its end
    np.testing.assert_allclose(w.reshape(n_rays, n_s).numpy(), ref["crm_weights"].numpy(), rtol=1e-6, atol=0)
    np.testing.assert_allclose(last.numpy(), ref["crm_alphainv_cum"][:, -1].numpy(), rtol=1e-6, atol=0)


def _early_stop_expectation(golden):
    g = golden("ref_fns_earlystop.npz")
    alpha, w_ref, acc, stop = (g[k] for k in ("es_alpha", "es_weights", "es_alphainv_cum", "es_stop"))
    n_rays, n_s = alpha.shape
    want_w = np.zeros_like(w_ref)
    want_last, want_end = np.empty(n_rays, np.float32), np.empty(n_rays, np.int64)
    for r in range(n_rays):
        k = n_s if stop[r] < 0 else int(stop[r]) + 1        # samples the scan visits: up to and including the stop sample
        want_w[r, :k] = w_ref[r, :k]                        # ... carry the cumprod form's weights; the rest stay zero
        want_last[r] = acc[r, k]                            # transmittance right behind the last visited sample
        want_end[r] = r * n_s + k
    return alpha, want_w, want_last, want_end, stop


def test_alpha2weight_early_stop_prefix_matches_cumprod_compositing(oracle, golden):
    """VERDICT r3 missing 5.  render_utils_kernel.cu:592-600 against model/dvgo.py:409-417 executed on rays that DO reach
    T < 1e-3 (tests/golden/ref_fns_earlystop.npz, oracle/make_golden_ref_fns.py early_stop): weights equal the cumprod form's up
    to and including the sample behind which T < 1e-3, are zero behind it, `alphainv_last` is T at the stop, `i_end` the index
    behind the stop sample.  1e-6 relative as in the no-stop case (double vs float32 running product)."""
    alpha, want_w, want_last, want_end, stop = _early_stop_expectation(golden)
    n_rays, n_s = alpha.shape
    assert (stop >= 0).sum() >= 20 and (stop < 0).sum() >= 5 and (stop == n_s - 1).any() and (want_last < 1e-6).any()
    ray_id = torch.arange(n_rays).repeat_interleave(n_s)
    w, last, i_end = oracle.alphas2weights(torch.from_numpy(alpha).reshape(-1).contiguous(), ray_id, n_rays)
    assert np.array_equal(np.asarray(i_end), want_end)
    np.testing.assert_allclose(w.reshape(n_rays, n_s).numpy(), want_w, rtol=1e-6, atol=0)
    assert np.all(w.reshape(n_rays, n_s).numpy()[want_w == 0] == 0)
    np.testing.assert_allclose(last.numpy(), want_last, rtol=1e-6, atol=0)


def test_gradient_volume_matches_the_reference(oracle, ref):
    """model/nerf.py:485-494 ('interpolate')."""
    assert torch.equal(oracle.neus_sdf_gradient(ref["gv_sdf"], ref["gv_voxel_size"]), ref["gv_interpolate"])
    # the two non-default modes (:495-506) and the grad_conv weight (:224-247)
    assert torch.equal(oracle.neus_sdf_gradient(ref["gv_sdf"], ref["gv_voxel_size"], 'raw'), ref["gv_raw"])
    assert torch.equal(oracle.grad_conv_weight(ref["gv_voxel_size"], 0), ref["gradconv_w_0"])
    assert torch.equal(oracle.grad_conv_weight(ref["gv_voxel_size"], 0.5), ref["gradconv_w_05"])
    # (the fixture's 'grad_conv' volume was produced with the sigma = 0.5 weight: the last init_gradient_conv call of the generator)
    got = oracle.neus_sdf_gradient(ref["gv_sdf"], ref["gv_voxel_size"], 'grad_conv', oracle.grad_conv_weight(ref["gv_voxel_size"], 0.5))
    assert rel_l2(got, ref["gv_grad_conv"]) < 1e-7


@pytest.mark.parametrize("ks,sigma", [(3, 1.0), (5, 0.8)])
def test_gaussian_smoothing_matches_the_reference(oracle, ref, ks, sigma):
    """model/nerf.py:260-272: the taps, and the replicate-padded convolution with them."""
    k = oracle.gaussian_kernel3d(ks, sigma)
    assert torch.equal(k, ref[f"smooth_w_{ks}"][0, 0])
    assert rel_l2(oracle.smooth_conv(ref["gv_sdf"], k), ref[f"smooth_out_{ks}"]) < 1e-7


def test_tv_smoothing_taps_match_the_reference(oracle, ref):
    """model/nerf.py:226-236,250-252 (tv_smooth_conv of init_gradient_conv), sigma 0 and 0.5."""
    assert torch.equal(oracle.tv_smooth_kernel(0), ref["tvsmooth_w_0"][0, 0])
    assert torch.equal(oracle.tv_smooth_kernel(0.5), ref["tvsmooth_w_05"][0, 0])


def test_l2_normalize_and_orientation_loss_match_the_reference(oracle, ref):
    """model/nerf.py:480-483 and :469-478."""
    assert torch.equal(oracle.l2_normalize(ref["l2n_x"]), ref["l2n_out"])
    res = dict(rgb_marched=torch.zeros(1, 3), weights=ref["ori_weights"], normal=ref["ori_normal"], viewdirs=ref["ori_viewdirs"])
    loss = oracle.fine_losses(res, torch.zeros(1, 3), dict(weight_main=0.0, weight_orientation=1.0))
    assert torch.equal(loss, ref["ori_loss"])


@pytest.mark.parametrize("C", [1, 3, 12])
def test_trilinear_lookup_matches_the_reference(oracle, ref, C):
    """model/grid.py:49-68 (DenseGrid.forward) and model/nerf.py:639-672 (grid_sampler, sample_ret): points inside, on the
    faces, and outside the box (zero padding)."""
    out = oracle.dense_grid_forward(ref[f"tri_grid_{C}"], ref["tri_pts"], ref["tri_lo"], ref["tri_hi"])
    assert torch.equal(out, ref[f"tri_dense_{C}"])
    assert torch.equal(out, ref[f"tri_sampler_{C}"])


def test_padded_sampler_matches_the_reference(oracle, ref):
    """model/nerf.py:734-758 (sample_ray_ori, is_train=False)."""
    P = dict(xyz_min=torch.tensor([-1.0] * 3), xyz_max=torch.tensor([1.0] * 3), voxel_size=torch.tensor(2.0 / 15))
    pts, mask, step = oracle.sample_ray_ori(P, (16, 16, 16), ref["sro_rays_o"], ref["sro_rays_d"], 2.0, 6.0, 0.5)
    assert torch.equal(pts, ref["sro_pts"]) and torch.equal(mask, ref["sro_mask_outbbox"].bool()) and torch.equal(step, ref["sro_step"])


def test_mask_cache_matches_the_reference(oracle, ref):
    """model/nerf.py:1193-1209 (max-pooled mask, trilinear sample >= thres)."""
    mc = oracle.make_mask_cache(ref["mc_raw"], ref["tri_lo"], ref["tri_hi"], 1e-3)
    assert torch.equal(oracle.mask_cache_forward(mc, ref["mc_pts"]), ref["mc_keep"].bool())


# ---- vectors of tests/golden/ref_fns_cuda_shim.npz: the reference's own bodies of neus_alpha_from_sdf_scatter, sample_sdfs and
# ---- grid_sampler(sample_grad=True), executed with torch.Tensor.cuda as the identity (oracle/make_golden_ref_fns.py: no GPU here)
@pytest.fixture(scope="module")
def shim(golden):
    g = golden("ref_fns_cuda_shim.npz")
    return {k: (torch.from_numpy(g[k]) if g[k].dtype.kind in "fbiu" else g[k]) for k in g.files}


def test_neus_alpha_matches_the_reference(oracle, shim):
    """model/nerf.py:510-544 (scheduled s_val, use_mid): alpha of 600 samples, at two points of the s_val schedule."""
    for tag in ("a", "b"):
        s_val = oracle.s_val_schedule(int(shim[f"alpha_step_{tag}"]), 50, 0.05)
        assert abs(s_val - float(shim[f"alpha_s_val_{tag}"])) <= 1e-7 * abs(s_val)        # (stored as float32)
        got = oracle.neus_alpha_from_sdf_scatter(shim["alpha_viewdirs"], shim["alpha_ray_id"], shim["alpha_dist"], shim["alpha_sdf"],
                                                 shim["alpha_grad"], s_val)
        assert torch.equal(got, shim[f"alpha_{tag}"])
        assert float(got.max()) == 1.0 and float(got.min()) < 2e-5         # the vector spans the whole range


def test_sdf_taps_match_the_reference(oracle, shim):
    """model/nerf.py:597-637 (sample_sdfs: 6 K clamped axis taps, 3 K tap differences, optional normalisation) and :639-672
    (grid_sampler with sample_grad: value, xyz-ordered gradient and taps)."""
    args = (shim["taps_pts"], shim["taps_grid"], shim["taps_lo"], shim["taps_hi"], shim["taps_voxel_size"])
    for tag, norm in (("k4", True), ("k4_raw", False), ("k1", False)):
        feat, grad = oracle.sample_sdfs(*args, shim[f"taps_disp_{tag}"].tolist(), use_grad_norm=norm)
        assert torch.equal(feat, shim[f"taps_feat_{tag}"]), tag
        assert torch.equal(grad, shim[f"taps_grad_{tag}"]), tag
    val, grad, feat = oracle.grid_sampler_ret_grad(*args)
    assert torch.equal(val, shim["gs_val"]) and torch.equal(grad, shim["gs_grad_xyz"]) and torch.equal(feat, shim["gs_feat_xyz"])
