"""Golden vectors from the REFERENCE'S OWN pure-torch functions (run once, in the build container; needs /root/reference).

    python oracle/make_golden_ref_fns.py      ->  tests/golden/ref_fns.npz

`model/nerf.py`, `model/dvgo.py` and `model/grid.py` cannot be imported (their import-time `load()` JIT-builds the CUDA
extensions: SURVEY.md 8c), but many of their functions are plain torch.  This script never imports those modules: it parses
the files with `ast`, takes single function definitions out of the tree, compiles each ON ITS OWN in a namespace holding only
torch / F / nn / np (and the sibling functions it names), and calls it on fixed inputs.  Methods are called with a
`types.SimpleNamespace` carrying exactly the attributes the body reads (`self.sdf.grid`, `self.voxel_size`, ...): data, not
behaviour.  Nothing of the reference's text is written anywhere: only inputs and outputs go into the .npz.

What is taken (file:line of the definitions executed) and what it pins (tests/test_oracle_cpu.py, tests/test_ref_pins_gpu.py):

  model/nerf.py:1212-1221  total_variation(v, mask)          -> oracle.total_variation('nerf'), fgs_tv_loss_* (f2)
  model/dvgo.py:420-428    total_variation(v, mask)          -> oracle.total_variation('dvgo'), fgs_tv_loss_* (f2)
  model/dvgo.py:409-417    cumprod_exclusive, get_ray_marching_ray
                                                             -> oracle alpha2weight / fgs_alpha2weight_fwd on rays that never
                                                                reach T < 1e-3 (a9)
  model/nerf.py:485-508    nerf.neus_sdf_gradient  ('interpolate', 'raw')
                                                             -> oracle.neus_sdf_gradient, fgs_sdf_gradvol_fwd (a6)
  model/nerf.py:260-272    nerf._gaussian_3dconv             -> oracle.gaussian_kernel3d / smooth_conv, fgs_smooth3d_fwd (a6)
  model/nerf.py:224-258    nerf.init_gradient_conv           -> oracle.tv_smooth_kernel (f2), the 'grad_conv' gradient mode
  model/nerf.py:430-447    nerf.density_total_variation      -> DenseGrid / nerf.density_total_variation here (f2)
  model/nerf.py:449-459    nerf.k0_total_variation           -> nerf.k0_total_variation here (f2)
  model/nerf.py:480-483    nerf.l2_normalize                 -> oracle.l2_normalize (a11)
  model/nerf.py:469-478    nerf.orientation_loss             -> oracle.fine_losses' orientation term, fgs_fine_loss_fwd
  model/nerf.py:639-672    nerf.grid_sampler (sample_ret)    -> oracle.dense_grid_forward, fgs_trilerp_fwd (a5)
  model/grid.py:49-68      DenseGrid.forward                 -> the same (C = 1, 3, 12)
  model/nerf.py:734-758    nerf.sample_ray_ori               -> oracle.sample_ray_ori (a3 cross-check: the padded sampler)
  model/nerf.py:1203-1209  MaskCache.forward                 -> oracle.mask_cache_forward, nerf.MaskCache here (a4)

Second file, tests/golden/ref_fns_cuda_shim.npz (`main_cuda_shim`).  Three more functions are pure torch except that their
bodies move small constants to the GPU with `.cuda()`: `nerf.neus_alpha_from_sdf_scatter` (model/nerf.py:510-544, `torch.ones(1)
.cuda()`), `nerf.sample_sdfs` (:597-637, three index tables), and through it `nerf.grid_sampler(sample_grad=True)` (:639-672).
There is no GPU in this container.  For THESE calls only, and kept apart from the vectors above, the generator runs the
unmodified function bodies with `torch.Tensor.cuda` temporarily replaced by the identity (the tensor stays on the CPU): a
change of placement, not of arithmetic -- every value is produced by the reference's own statements on torch's CPU ops, exactly
like the vectors of the first file.  They pin the NeuS alpha (a8) and the 6 K axis taps / tap differences (a7).

Third file, tests/golden/ref_fns_earlystop.npz (`main_early_stop`, `python oracle/make_golden_ref_fns.py early_stop`): the same
`get_ray_marching_ray` on rays that DO reach T < 1e-3 -- a prefix pin of the early-terminating scan (see the function).

Not runnable here under any honest arrangement, hence still unpinned: the CUDA extension kernels and their wrappers (a3 packed
sampler, a14, a15; a9's early stop only through the prefix pin above), torch_scatter.segment_coo, PyMCubes.
"""
from __future__ import annotations

import ast
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "ref_fns.npz")


def extract(path: str, name: str, cls: str = None, env: dict = None):
    """Compile ONE function definition of `path` (a method of `cls` if given) on its own and return the function object."""
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    fn = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.decorator_list = []                        # (torch.no_grad() on MaskCache.forward: applied by the caller below)
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "F": F, "nn": nn, "np": np, "__builtins__": __builtins__}
    ns.update(env or {})
    exec(compile(mod, path, "exec"), ns)          # noqa: S102 -- the definition only; nothing at module level runs
    return ns[name], (fn.lineno, fn.end_lineno)


def main() -> None:
    sys.dont_write_bytecode = True
    torch.manual_seed(777)
    g = torch.Generator().manual_seed(777)
    nerf_py, dvgo_py, grid_py = (os.path.join(REF, "model", f) for f in ("nerf.py", "dvgo.py", "grid.py"))
    out, lines = {}, {}

    # ---- total_variation, both variants, with and without mask --------------------------------------------------------
    tv_nerf, lines["tv_nerf"] = extract(nerf_py, "total_variation")
    tv_dvgo, lines["tv_dvgo"] = extract(dvgo_py, "total_variation")
    v1 = torch.randn(1, 1, 9, 10, 11, generator=g) + 0.5
    v12 = torch.randn(1, 12, 5, 6, 7, generator=g)
    m1 = torch.rand(1, 1, 9, 10, 11, generator=g) > 0.3
    m12 = (torch.rand(1, 1, 5, 6, 7, generator=g) > 0.3).repeat(1, 12, 1, 1, 1)
    out.update(tv_v1=v1, tv_v12=v12, tv_m1=m1, tv_m12=m12,
               tv_nerf_v1=tv_nerf(v1), tv_nerf_v1_m=tv_nerf(v1, m1), tv_nerf_v12=tv_nerf(v12), tv_nerf_v12_m=tv_nerf(v12, m12),
               tv_dvgo_v1=tv_dvgo(v1), tv_dvgo_v1_m=tv_dvgo(v1, m1), tv_dvgo_v12=tv_dvgo(v12), tv_dvgo_v12_m=tv_dvgo(v12, m12))

    # ---- cumprod compositing (no early stop) ----------------------------------------------------------------------------
    cpe, lines["cumprod_exclusive"] = extract(dvgo_py, "cumprod_exclusive")
    grm, lines["get_ray_marching_ray"] = extract(dvgo_py, "get_ray_marching_ray", env={"cumprod_exclusive": cpe})
    alpha = torch.rand(37, 23, generator=g) * 0.2            # 23 samples of alpha <= 0.2: T stays far above 1e-3
    alpha[3] = 0.0
    w, acc = grm(alpha)
    out.update(crm_alpha=alpha, crm_weights=w, crm_alphainv_cum=acc)

    # ---- gradient volume -------------------------------------------------------------------------------------------------
    nsg, lines["neus_sdf_gradient"] = extract(nerf_py, "neus_sdf_gradient", cls="nerf")
    sdf = torch.randn(1, 1, 9, 10, 11, generator=g)
    vs = torch.tensor(0.0371)
    me = types.SimpleNamespace(sdf=types.SimpleNamespace(grid=sdf), grad_mode='interpolate', voxel_size=vs)
    out.update(gv_sdf=sdf, gv_voxel_size=vs, gv_interpolate=nsg(me), gv_raw=nsg(me, mode='raw'))

    # ---- smoothing conv, gradient conv, TV smoothing conv ----------------------------------------------------------------
    g3, lines["_gaussian_3dconv"] = extract(nerf_py, "_gaussian_3dconv", cls="nerf")
    for ks, sigma in ((3, 1.0), (5, 0.8)):
        conv = g3(me, ksize=ks, sigma=sigma)
        with torch.no_grad():
            out[f"smooth_w_{ks}"] = conv.weight.detach().clone()
            out[f"smooth_out_{ks}"] = conv(sdf)
    igc, lines["init_gradient_conv"] = extract(nerf_py, "init_gradient_conv", cls="nerf")
    for sigma in (0, 0.5):
        me_gc = types.SimpleNamespace(voxel_size=vs)
        igc(me_gc, sigma=sigma)
        tag = "0" if sigma == 0 else "05"
        with torch.no_grad():
            out[f"gradconv_w_{tag}"] = me_gc.grad_conv.weight.detach().clone()
            out[f"tvsmooth_w_{tag}"] = me_gc.tv_smooth_conv.weight.detach().clone()
            out[f"gradconv_out_{tag}"] = me_gc.grad_conv(sdf)
    me.grad_conv = me_gc.grad_conv
    out["gv_grad_conv"] = nsg(me, mode='grad_conv')

    # ---- density / k0 total variation (the autograd TV terms of the training loop) ---------------------------------------
    dtv, lines["density_total_variation"] = extract(nerf_py, "density_total_variation", cls="nerf", env={"total_variation": tv_nerf})
    ktv, lines["k0_total_variation"] = extract(nerf_py, "k0_total_variation", cls="nerf", env={"total_variation": tv_nerf})
    me_gc0 = types.SimpleNamespace(voxel_size=vs)
    igc(me_gc0, sigma=0)
    gradient = nsg(me)
    for tag, mask in (("nomask", None), ("mask", m1)):
        me_tv = types.SimpleNamespace(sdf=types.SimpleNamespace(grid=sdf), k0=types.SimpleNamespace(grid=v12), voxel_size=vs,
                                      nonempty_mask=mask, gradient=gradient, tv_smooth_conv=me_gc0.tv_smooth_conv)
        out[f"dtv_sdf_{tag}"] = dtv(me_tv, sdf_tv=0.1, smooth_grad_tv=0)
        out[f"dtv_smooth_{tag}"] = dtv(me_tv, sdf_tv=0, smooth_grad_tv=0.05)
    me_k = types.SimpleNamespace(k0=types.SimpleNamespace(grid=v12), nonempty_mask=None)
    out["ktv_nomask"] = ktv(me_k)
    me_k.nonempty_mask = m12[:, :1]
    out["ktv_mask"] = ktv(me_k)

    # ---- l2_normalize, orientation loss -------------------------------------------------------------------------------------
    l2n, lines["l2_normalize"] = extract(nerf_py, "l2_normalize", cls="nerf")
    x = torch.randn(50, 3, generator=g)
    x[7] = 0.0
    x[8] = 1e-30
    out.update(l2n_x=x, l2n_out=l2n(None, x))
    ol, lines["orientation_loss"] = extract(nerf_py, "orientation_loss", cls="nerf")
    rr = dict(weights=torch.rand(200, generator=g), normal=l2n(None, torch.randn(200, 3, generator=g)),
              viewdirs=l2n(None, torch.randn(200, 3, generator=g)))
    out.update(ori_weights=rr['weights'], ori_normal=rr['normal'], ori_viewdirs=rr['viewdirs'], ori_loss=ol(None, rr))

    # ---- trilinear lookups: nerf.grid_sampler and DenseGrid.forward ---------------------------------------------------------
    gs, lines["grid_sampler"] = extract(nerf_py, "grid_sampler", cls="nerf")
    dgf, lines["DenseGrid.forward"] = extract(grid_py, "forward", cls="DenseGrid")
    lo, hi = torch.tensor([-1.0, -0.8, -1.2]), torch.tensor([1.1, 0.9, 1.0])
    pts = lo + (hi - lo) * (torch.rand(300, 3, generator=g) * 1.2 - 0.1)        # some outside the box
    pts[:8] = torch.stack([lo, hi, (lo + hi) / 2, lo + (hi - lo) * torch.tensor([1.0, 0.0, 0.5]),
                           lo - 1e-3, hi + 1e-3, lo + (hi - lo) * 0.999999, lo + (hi - lo) * 1e-7])
    me_s = types.SimpleNamespace(nearest=False, xyz_min=lo, xyz_max=hi)
    out.update(tri_lo=lo, tri_hi=hi, tri_pts=pts)
    for C in (1, 3, 12):
        grid = torch.randn(1, C, 9, 10, 11, generator=g)
        out[f"tri_grid_{C}"] = grid
        out[f"tri_sampler_{C}"] = gs(me_s, pts, grid)
        me_d = types.SimpleNamespace(channels=C, xyz_min=lo, xyz_max=hi, grid=grid)
        out[f"tri_dense_{C}"] = dgf(me_d, pts)

    # ---- padded ray sampler -----------------------------------------------------------------------------------------------
    sro, lines["sample_ray_ori"] = extract(nerf_py, "sample_ray_ori", cls="nerf")
    from fgs_nerf_amd import synth
    ro, rd, _ = synth.random_rays(48, n_views=4, H=64, W=64, seed=11)
    rd[5, 1] = 0.0                                                               # an axis-parallel component
    me_r = types.SimpleNamespace(sdf=types.SimpleNamespace(grid=torch.zeros(1, 1, 16, 16, 16)), xyz_min=torch.tensor([-1.0] * 3),
                                 xyz_max=torch.tensor([1.0] * 3), voxel_size=torch.tensor(2.0 / 15))
    rays_pts, mask_outbbox, step = sro(me_r, ro, rd, near=2.0, far=6.0, stepsize=0.5, is_train=False)
    out.update(sro_rays_o=ro, sro_rays_d=rd, sro_pts=rays_pts, sro_mask_outbbox=mask_outbbox, sro_step=step)

    # ---- MaskCache.forward ---------------------------------------------------------------------------------------------------
    mcf, lines["MaskCache.forward"] = extract(nerf_py, "forward", cls="MaskCache")
    raw_mask = (torch.rand(1, 1, 12, 13, 14, generator=g) > 0.8).float()
    pooled = F.max_pool3d(raw_mask, kernel_size=3, padding=1, stride=1)          # MaskCache.__init__ (model/nerf.py:1198-1199)
    me_m = types.SimpleNamespace(xyz_min=lo, xyz_max=hi, sdf_mask=pooled, mask_cache_thres=1e-3)
    with torch.no_grad():
        out.update(mc_raw=raw_mask, mc_pts=pts, mc_keep=mcf(me_m, pts))

    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    arrays["source_lines"] = np.array([f"{k}:{a}-{b}" for k, (a, b) in sorted(lines.items())])
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(arrays), "arrays")
    for s in arrays["source_lines"]:
        print("  ", s)


class _cuda_is_identity:
    """`with _cuda_is_identity():` -- torch.Tensor.cuda returns the tensor itself (see the module docstring)."""

    def __enter__(self):
        self.orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda = self.orig
        return False


def main_cuda_shim() -> None:
    sys.dont_write_bytecode = True
    g = torch.Generator().manual_seed(778)
    nerf_py = os.path.join(REF, "model", "nerf.py")
    out, lines = {}, {}
    alpha_fn, lines["neus_alpha_from_sdf_scatter"] = extract(nerf_py, "neus_alpha_from_sdf_scatter", cls="nerf")
    sdfs_fn, lines["sample_sdfs"] = extract(nerf_py, "sample_sdfs", cls="nerf")
    gs_fn, lines["grid_sampler"] = extract(nerf_py, "grid_sampler", cls="nerf")

    # ---- NeuS alpha on a spread of (sdf, gradient, view direction) incl. back-facing samples, tiny and large |sdf| --------------
    M, N = 600, 40
    ray_id = torch.sort(torch.randint(0, N, (M,), generator=g))[0]
    viewdirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    sdf = torch.randn(M, generator=g) * 0.05
    sdf[:20] *= 40.0
    sdf[20:30] = 0.0
    grad = torch.randn(M, 3, generator=g) * torch.rand(M, 1, generator=g) * 2.0
    dist = torch.tensor(0.5) * torch.tensor(0.0125)
    for tag, gstep in (("a", 1000), ("b", 15000)):
        me = types.SimpleNamespace(s_learn=False, s_ratio=50, s_start=0.05, step_start=0, s_val=nn.Parameter(torch.ones(1) * 0.05))
        with _cuda_is_identity(), torch.no_grad():
            s_val, alpha = alpha_fn(me, viewdirs, ray_id, dist, sdf, grad, gstep, True)
        out[f"alpha_{tag}"], out[f"alpha_s_val_{tag}"], out[f"alpha_step_{tag}"] = alpha, torch.tensor(s_val), torch.tensor(gstep)
    out.update(alpha_viewdirs=viewdirs, alpha_ray_id=ray_id, alpha_sdf=sdf, alpha_grad=grad, alpha_dist=dist)

    # ---- sample_sdfs: 6 K taps and 3 K tap differences, with and without normalisation; points on and outside the faces ----------
    lo, hi = torch.tensor([-1.0, -0.8, -1.2]), torch.tensor([1.1, 0.9, 1.0])
    pts = lo + (hi - lo) * (torch.rand(200, 3, generator=g) * 1.1 - 0.05)
    pts[:4] = torch.stack([lo, hi, lo + (hi - lo) * 0.999, lo + (hi - lo) * 0.001])
    grid = torch.randn(1, 1, 9, 10, 11, generator=g)
    vs = torch.tensor(0.21)
    me_s = types.SimpleNamespace(xyz_min=lo, xyz_max=hi, voxel_size=vs, nearest=False)
    me_s.sample_sdfs = lambda *a, **k: sdfs_fn(me_s, *a, **k)
    out.update(taps_lo=lo, taps_hi=hi, taps_pts=pts, taps_grid=grid, taps_voxel_size=vs)
    for tag, disp, norm in (("k4", [0.5, 1.0, 1.5, 2.0], True), ("k4_raw", [0.5, 1.0, 1.5, 2.0], False), ("k1", [1.0], False)):
        with _cuda_is_identity(), torch.no_grad():
            feat, gr = sdfs_fn(me_s, pts, grid, displace_list=disp, use_grad_norm=norm)
        out[f"taps_feat_{tag}"], out[f"taps_grad_{tag}"] = feat, gr
        out[f"taps_disp_{tag}"] = torch.tensor(disp)
    with _cuda_is_identity(), torch.no_grad():
        val, grad_xyz, feat_xyz = gs_fn(me_s, pts, grid, sample_ret=True, sample_grad=True)
    out.update(gs_val=val, gs_grad_xyz=grad_xyz, gs_feat_xyz=feat_xyz)

    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    arrays["source_lines"] = np.array([f"{k}:{a}-{b}" for k, (a, b) in sorted(lines.items())])
    path = os.path.join(ROOT, "tests", "golden", "ref_fns_cuda_shim.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes;", len(arrays), "arrays")
    for s_ in arrays["source_lines"]:
        print("  ", s_)


def main_early_stop() -> None:
    """Third file, tests/golden/ref_fns_earlystop.npz: `get_ray_marching_ray` (model/dvgo.py:409-417, with cumprod_exclusive) on
    rays that DO reach T < 1e-3.  The CUDA scan (render_utils_kernel.cu:592-600) multiplies the same factors in the same order and
    leaves the loop behind the first sample after which T < 1e-3: up to and including that sample its weights are the cumprod
    form's, behind it they stay zero, and `alphainv_last` is the cumprod form's transmittance right behind the stop sample.  A
    partial pin (the reference function has no stop) of the one piece of a9 that no reference-executed vector covered."""
    sys.dont_write_bytecode = True
    g = torch.Generator().manual_seed(779)
    dvgo_py = os.path.join(REF, "model", "dvgo.py")
    lines = {}
    cpe, lines["cumprod_exclusive"] = extract(dvgo_py, "cumprod_exclusive")
    grm, lines["get_ray_marching_ray"] = extract(dvgo_py, "get_ray_marching_ray", env={"cumprod_exclusive": cpe})
    n_rays, n_s = 48, 40
    alpha = torch.rand(n_rays, n_s, generator=g) * torch.linspace(0.08, 0.9, n_rays)[:, None]     # ray r: alphas up to 0.08 .. 0.9
    alpha[5] = 0.0                                        # an empty ray
    alpha[6, :] = 0.0
    alpha[6, 17] = 0.9995                                 # one sample takes T below 1e-3 at once
    alpha[7, 3] = 1.0 - 2.0 ** -23                        # an all but opaque sample (exactly 1 is where the two forms part: the
                                                          # cumprod form clamps 1 - alpha at 1e-10, the scan multiplies by 0)
    alpha[8, -1] = 0.99999                                # the stop falls on the LAST sample
    w, acc = grm(alpha)
    T_behind = acc[:, 1:]                                 # transmittance right behind sample j
    below = T_behind < 1e-3
    stop = torch.where(below.any(dim=1), below.float().argmax(dim=1), torch.full((n_rays,), -1, dtype=torch.long))
    # no decision of this fixture may hang on rounding: every T is at least 1e-4 (relative) away from the threshold
    assert float(((T_behind - 1e-3).abs() / 1e-3).min()) > 1e-4
    assert int((stop >= 0).sum()) >= 20 and int((stop < 0).sum()) >= 5
    arrays = dict(es_alpha=alpha.numpy(), es_weights=w.numpy(), es_alphainv_cum=acc.numpy(), es_stop=stop.numpy(),
                  source_lines=np.array([f"{k}:{a}-{b}" for k, (a, b) in sorted(lines.items())]))
    path = os.path.join(ROOT, "tests", "golden", "ref_fns_earlystop.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes; rays that stop:", int((stop >= 0).sum()), "of", n_rays)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    if len(sys.argv) > 1 and sys.argv[1] == "early_stop":
        main_early_stop()
        sys.exit(0)
    if len(sys.argv) < 2 or sys.argv[1] == "pure":
        main()
    if len(sys.argv) < 2 or sys.argv[1] == "cuda_shim":
        main_cuda_shim()
