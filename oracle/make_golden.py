"""Generates the committed golden fixtures under tests/golden/ (run once, in the build container).

    python oracle/make_golden.py            # everything
    python oracle/make_golden.py rays       # only the fixtures that need /root/reference
    python oracle/make_golden.py maskcache_rays   # the mask-cache ray pre-filter fixture alone (needs /root/reference)

Two kinds of fixture:

* ``rays_*.npz`` -- produced by IMPORTING the reference's own model/dvgo_ray.py (pure torch/numpy, loaded by
  file path, bytecode writing disabled).  These pin fgs-nerf_amd/rays.py to the reference bit for bit.
  /root/reference does not exist on the GPU box, so only the data travels.
* everything else -- produced by the oracle (oracle/oracle.py: C restatement + the torch CPU ops the
  reference calls).  They guard the oracle against drift (torch version, compiler) and give the GPU
  parity tests fixed inputs.  "parity unpinned" for the kernel restatement, see fgs_oracle.c.
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != os.path.join(ROOT, "oracle")]
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

from oracle import oracle as O  # noqa: E402


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


def load_reference_rays():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_dvgo_ray", os.path.join(REF, "model", "dvgo_ray.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_rays():
    """(H, W, K, c2w, flags) -> (rays_o, rays_d, viewdirs) from the reference's get_rays_of_a_view."""
    ref = load_reference_rays()
    from fgs_nerf_amd import synth
    H, W = 24, 32
    K = synth.intrinsics(H, W, fov_x=0.6911)
    cases = {
        "blender": dict(inverse_y=False, flip_x=False, flip_y=False, mode='center'),
        "dtu": dict(inverse_y=True, flip_x=False, flip_y=False, mode='center'),
        "flipped": dict(inverse_y=False, flip_x=True, flip_y=True, mode='lefttop'),
    }
    for name, fl in cases.items():
        c2w = torch.from_numpy(synth.look_at_origin(37.0, 25.0, 3.5))
        ro, rd, vd = ref.get_rays_of_a_view(H, W, K, c2w, False, fl['inverse_y'], fl['flip_x'], fl['flip_y'], mode=fl['mode'])
        save(f"rays_{name}.npz", H=H, W=W, K=K, c2w=c2w, inverse_y=fl['inverse_y'], flip_x=fl['flip_x'],
             flip_y=fl['flip_y'], mode=np.array(fl['mode']), rays_o=ro, rays_d=rd, viewdirs=vd)
    # random mode (RNG draw order) and NDC
    torch.manual_seed(1234)
    c2w = torch.from_numpy(synth.look_at_origin(200.0, 10.0, 4.0))
    ro, rd, vd = ref.get_rays_of_a_view(H, W, K, c2w, True, False, False, False, mode='random')
    save("rays_random_ndc.npz", H=H, W=W, K=K, c2w=c2w, seed=1234, rays_o=ro, rays_d=rd, viewdirs=vd)
    np.random.seed(5)
    gen = ref.batch_indices_generator(10, 4)
    save("batch_indices.npz", seed=5, N=10, BS=4, batches=np.stack([next(gen).numpy() for _ in range(5)]))


class DuckModel:
    """What get_training_rays_in_maskcache_sampling asks of `model` (model/nerf_ray.py:230-232), backed by the oracle:
    `sample_ray_ori` (model/nerf.py:734-758 restated) and a `mask_cache` callable (model/nerf.py:1202-1209 restated)."""

    def __init__(self, P, grid_shape, mc):
        self.P, self.grid_shape, self.mc = P, tuple(grid_shape), mc

    def sample_ray_ori(self, rays_o, rays_d, near, far, stepsize, is_train=False, **render_kwargs):
        return O.sample_ray_ori(self.P, self.grid_shape, rays_o, rays_d, near, far, stepsize)

    def mask_cache(self, pts):
        return O.mask_cache_forward(self.mc, pts)


def maskcache_scene():
    """Inputs of the mask-cache ray pre-filter fixture: 3 small views (two sizes), an occupied blob off-centre so that
    whole image regions are dropped.  Returned as plain tensors so the test can rebuild the same duck model."""
    from fgs_nerf_amd import synth
    G = 16
    lo, hi = torch.tensor([-1., -1., -1.]), torch.tensor([1., 1., 1.])
    voxel_size, world = O.grid_resolution(lo, hi, G ** 3)
    ax = torch.linspace(-1, 1, G)
    x, y, z = torch.meshgrid(ax, ax, ax, indexing='ij')
    occupied = ((x - 0.3) ** 2 + (y + 0.2) ** 2 + (z - 0.1) ** 2).sqrt() < 0.45
    sdf_mask = (occupied.float() * 1e-3)[None, None]
    HW = np.array([[40, 48], [40, 48], [33, 25]])
    Ks = np.stack([synth.intrinsics(int(h), int(w), fov_x=0.6911) for h, w in HW])
    poses = torch.stack([torch.from_numpy(synth.look_at_origin(a, e, 4.0)) for a, e in ((20., 30.), (140., 10.), (260., 50.))])
    gen = torch.Generator().manual_seed(31)
    images = [torch.rand(int(h), int(w), 3, generator=gen) for h, w in HW]
    return dict(G=G, xyz_min=lo, xyz_max=hi, voxel_size=voxel_size, sdf_mask=sdf_mask, thres=0.5e-3, HW=HW, Ks=Ks,
                poses=poses, images=images, render_kwargs=dict(near=2.0, far=6.0, stepsize=0.5))


def make_maskcache_rays():
    """model/nerf_ray.py:208-250 run by IMPORTING the reference's own file, on the duck model above."""
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_nerf_ray", os.path.join(REF, "model", "nerf_ray.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    sc = maskcache_scene()
    P = dict(xyz_min=sc['xyz_min'], xyz_max=sc['xyz_max'], voxel_size=sc['voxel_size'])
    mc = O.make_mask_cache(sc['sdf_mask'], sc['xyz_min'], sc['xyz_max'], sc['thres'])
    duck = DuckModel(P, (sc['G'],) * 3, mc)
    rgb_tr, ro_tr, rd_tr, vd_tr, imsz = ref.get_training_rays_in_maskcache_sampling(
        rgb_tr_ori=sc['images'], train_poses=sc['poses'], HW=sc['HW'], Ks=sc['Ks'], ndc=False, inverse_y=False, flip_x=False,
        flip_y=False, model=duck, render_kwargs=sc['render_kwargs'])
    n_all = int(sum(int(h) * int(w) for h, w in sc['HW']))
    assert 0 < len(rgb_tr) < n_all, (len(rgb_tr), n_all)      # the filter really dropped rays, and kept some
    save("maskcache_rays.npz", G=sc['G'], xyz_min=sc['xyz_min'], xyz_max=sc['xyz_max'], voxel_size=sc['voxel_size'],
         sdf_mask=sc['sdf_mask'], thres=sc['thres'], HW=sc['HW'], Ks=sc['Ks'], poses=sc['poses'],
         image0=sc['images'][0], image1=sc['images'][1], image2=sc['images'][2], near=2.0, far=6.0, stepsize=0.5,
         rgb_tr=rgb_tr, rays_o_tr=ro_tr, rays_d_tr=rd_tr, viewdirs_tr=vd_tr, imsz=np.array([int(n) for n in imsz]))


def special_rays():
    """~64 rays: hits, grazing, misses, axis-parallel (zero direction components), origins inside the bbox."""
    from fgs_nerf_amd import synth
    ro, rd, _ = synth.random_rays(40, n_views=5, H=64, W=64, seed=3)
    extra_o = torch.tensor([[0., 0., -4.], [0., 0., -4.], [0.2, 0.1, 0.3], [-0.5, 0.5, 0.0], [3., 3., 3.], [0., -4., 0.],
                            [1.0, 0., -4.], [0.999999, 0.3, -4.], [4., 0., 0.], [0., 0., 4.], [5., 5., 0.], [0.3, 0.3, -4.]])
    extra_d = torch.tensor([[0., 0., 1.], [0.01, 0.02, 1.], [0.3, -0.2, 0.5], [0., 1., 0.], [1., 1., 1.], [0., 2., 0.],
                            [0., 0., 1.], [0., 0., 1.], [-1., 0., 0.], [0., 0., -0.5], [-1., -1., 0.], [0., 0., -1.]])
    return torch.cat([ro, extra_o]).contiguous(), torch.cat([rd, extra_d]).contiguous()


def make_kernels():
    ro, rd = special_rays()
    lo, hi = np.array([-1., -1., -1.], np.float32), np.array([1., 1., 1.], np.float32)
    stepdist = np.float32(0.5 * 0.0625)
    pts, mask, ray_id, step_id, n_steps, t_min, t_max = O.K.sample_pts_on_rays(ro.numpy(), rd.numpy(), lo, hi, 0.2, 1e9, stepdist)
    save("sample_pts.npz", rays_o=ro, rays_d=rd, xyz_min=lo, xyz_max=hi, near=0.2, far=1e9, stepdist=stepdist,
         rays_pts=pts, mask_outbbox=mask, ray_id=ray_id, step_id=step_id, N_steps=n_steps, t_min=t_min, t_max=t_max)

    # alpha2weight incl. an early-terminating ray, an empty ray and a single-sample ray
    rng = np.random.RandomState(0)
    counts = [0, 1, 70, 5, 130, 0, 64, 65]
    ray_ids = np.concatenate([np.full(c, r, np.int64) for r, c in enumerate(counts)])
    alpha = rng.uniform(0, 0.08, size=ray_ids.size).astype(np.float32)
    alpha[ray_ids == 2] = rng.uniform(0.05, 0.5, size=70).astype(np.float32)     # terminates early
    alpha[ray_ids == 7] = 0.0
    w, T, last, i_s, i_e = O.K.alpha2weight(alpha, ray_ids, len(counts))
    gw = rng.randn(alpha.size).astype(np.float32)
    gl = rng.randn(len(counts)).astype(np.float32)
    g = O.K.alpha2weight_backward(alpha, w, T, last, i_s, i_e, len(counts), gw, gl)
    save("alpha2weight.npz", alpha=alpha, ray_id=ray_ids, n_rays=len(counts), weight=w, T=T, alphainv_last=last,
         i_start=i_s, i_end=i_e, grad_weights=gw, grad_last=gl, grad=g)

    # raw2alpha
    dens = rng.randn(257).astype(np.float32) * 3
    iv = rng.uniform(0.1, 1.0, 257).astype(np.float32)
    e, a = O.K.raw2alpha(dens, -4.0, 0.5)
    en, an = O.K.raw2alpha(dens, -4.0, iv)
    gb = rng.randn(257).astype(np.float32)
    save("raw2alpha.npz", density=dens, shift=-4.0, interval=0.5, interval_nonuni=iv, exp_d=e, alpha=a, exp_d_nonuni=en,
         alpha_nonuni=an, grad_back=gb, grad=O.K.raw2alpha_backward(e, gb, 0.5), grad_nonuni=O.K.raw2alpha_backward(en, gb, iv))

    # maskcache
    world = rng.rand(5, 6, 7) > 0.5
    xyz = (rng.rand(300, 3) * 2.6 - 1.3).astype(np.float32)
    scale = ((np.array(world.shape) - 1) / 2.0).astype(np.float32)
    shift = (1.0 * scale).astype(np.float32)
    save("maskcache.npz", world=world, xyz=xyz, scale=scale, shift=shift, out=O.K.maskcache_lookup(world, xyz, scale, shift))

    # TV + Adam single steps on a 5x6x7xC grid with a sparse grad
    C = 3
    param = rng.randn(1, C, 5, 6, 7).astype(np.float32)
    grad0 = (rng.randn(1, C, 5, 6, 7) * (rng.rand(1, C, 5, 6, 7) > 0.7)).astype(np.float32)
    maskf = (rng.rand(1, C, 5, 6, 7) > 0.3).astype(np.float32)
    out = {}
    for dense in (0, 1):
        g1 = grad0.copy(); O.K.total_variation_add_grad(param, g1, 0.3, 0.2, 0.1, dense)
        g2 = grad0.copy(); O.K.total_variation_add_grad(param, g2, 0.3, 0.2, 0.1, dense, mask=maskf)
        out[f"tv_dense{dense}"] = g1
        out[f"tv_masked_dense{dense}"] = g2
    save("tv.npz", param=param, grad=grad0, mask=maskf, wx=0.3, wy=0.2, wz=0.1, **out)
    perlr = rng.rand(1, C, 5, 6, 7).astype(np.float32)
    res = {}
    for mode in (0, 1, 2):
        p, m, v = param.copy(), np.zeros_like(param), np.zeros_like(param)
        for step in (1, 2, 3):
            O.K.adam_upd(p, grad0, m, v, step, 0.9, 0.99, 0.1, 1e-8, mode=mode, perlr=perlr)
        res.update({f"param_mode{mode}": p, f"exp_avg_mode{mode}": m, f"exp_avg_sq_mode{mode}": v})
    save("adam.npz", param=param, grad=grad0, perlr=perlr, beta1=0.9, beta2=0.99, lr=0.1, eps=1e-8, steps=3, **res)


def make_trilerp():
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(7)
    lo, hi = torch.tensor([-1., -0.5, 0.]), torch.tensor([1., 1.5, 3.])
    pts = torch.rand(200, 3, generator=gen) * (hi - lo) * 1.2 + lo - 0.1 * (hi - lo)   # some outside the volume
    pts[0] = lo; pts[1] = hi; pts[2] = (lo + hi) / 2
    out = dict(xyz_min=lo, xyz_max=hi, pts=pts)
    for C in (1, 3, 12):
        grid = torch.randn(1, C, 9, 10, 11, generator=gen, requires_grad=True)
        val = O.dense_grid_forward(grid, pts, lo, hi)
        go = torch.randn(val.shape, generator=gen)
        val.backward(go)
        out.update({f"grid_c{C}": grid.detach(), f"out_c{C}": val.detach(), f"grad_out_c{C}": go, f"grad_grid_c{C}": grid.grad})
    # axis taps on a 1-channel grid
    sdf = torch.randn(1, 1, 9, 10, 11, generator=gen, requires_grad=True)
    vs = torch.tensor(0.2)
    for name, disp, norm in (("k1", [1.0], False), ("k4", [0.5, 1.0, 1.5, 2.0], True)):
        feat, grad = O.sample_sdfs(pts[:, :], sdf, lo, hi, vs, disp, use_grad_norm=norm)
        out.update({f"taps_feat_{name}": feat.detach(), f"taps_grad_{name}": grad.detach()})
    out.update(sdf=sdf.detach(), voxel_size=vs)
    save("trilerp.npz", **out)


def make_e2e():
    """Tiny end-to-end forward_fine / forward_coarse (16^3, 32 rays, seed 777): outputs and all parameter grads."""
    from fgs_nerf_amd import synth
    for stage, kw, lossw in (("fine", synth.FINE_MODEL, synth.FINE_LOSS), ("coarse", synth.COARSE_MODEL, synth.COARSE_LOSS)):
        m = synth.build_model(16, kw, fused=False)
        P = synth.oracle_params(m)
        ro, rd, vd = synth.random_rays(32, n_views=4, H=64, W=64, seed=11)
        leaves = {'sdf': P['sdf'], 'k0': P['k0']}
        for net in ('rgbnet', 'refnet'):
            if P[net] is not None:
                for i, (W, b) in enumerate(P[net]):
                    leaves[f'{net}.{i}.weight'], leaves[f'{net}.{i}.bias'] = W, b
        for t in leaves.values():
            t.requires_grad_(True)
        fwd = O.forward_fine if stage == 'fine' else O.forward_coarse
        res = fwd(P, ro, rd, vd, global_step=1000, near=2.0, stepsize=0.5, bg=1)
        target = torch.rand(32, 3, generator=torch.Generator().manual_seed(1))
        loss = O.fine_losses(res, target, lossw)
        loss.backward()
        out = dict(rays_o=ro, rays_d=rd, viewdirs=vd, target=target, loss=loss.detach(), global_step=1000,
                   rgb_marched=res['rgb_marched'], sigmoid_rgb=res['sigmoid_rgb'], alphainv_cum=res['alphainv_cum'],
                   weights=res['weights'], ray_id=res['ray_id'], step_id=res['step_id'], raw_rgb=res['raw_rgb'],
                   raw_alpha=res['raw_alpha'], normal=res['normal'], n_total=res['n_total'], n_inbbox=res['n_inbbox'])
        for k, t in leaves.items():
            out['grad_' + k] = t.grad
        save(f"e2e_{stage}.npz", **out)


def make_extras():
    """Fixtures of the two rows outside the render path: the integrated directional encoding (restated generate_ide_fn,
    float32, sh_max_level 4; its known answer against scipy's Y_l^m is checked in tests/test_ide.py) and marching cubes
    (table-free restatement of the device convention, oracle/mcubes_ref.py, on a 12^3 off-centre sphere and a 7x8x9 noise
    field).  Neither has a counterpart fixture in the reference (parity unpinned)."""
    from oracle import mcubes_ref as R
    SEED = 777
    g = torch.Generator().manual_seed(SEED)
    d = torch.randn(64, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    kinv = torch.rand(64, 1, generator=g) * 0.5
    save("ide_deg4.npz", dirs=d, kappa_inv=kinv, ide=O.generate_ide_fn(4)(d, kinv))
    ax = np.linspace(-1, 1, 12, dtype=np.float32)
    x, y, z = np.meshgrid(ax, ax, ax, indexing='ij')
    sphere = (np.sqrt((x - 0.1) ** 2 + (y + 0.05) ** 2 + z ** 2) - 0.55).astype(np.float32)
    noise = np.random.default_rng(SEED).standard_normal((7, 8, 9)).astype(np.float32)
    out = {}
    for name, f, iso in (("sphere", sphere, 0.0), ("noise", noise, 0.2)):
        v, t = R.marching_cubes(f, iso)
        out.update({name + "_field": f, name + "_iso": np.float32(iso), name + "_vertices": v, name + "_triangles": t})
    save("mcubes.npz", **out)


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"rays", "kernels", "trilerp", "e2e", "extras"}
    if "extras" in what:
        make_extras()
    if "rays" in what:
        make_rays()
    if "rays" in what or "maskcache_rays" in what:
        make_maskcache_rays()
    if "kernels" in what:
        make_kernels()
    if "trilerp" in what:
        make_trilerp()
    if "e2e" in what:
        make_e2e()
