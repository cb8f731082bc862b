"""GPU parity tests of the fused fine-stage path (fgs-nerf_amd/fused.py) against the committed oracle outputs (16^3)
and against the operator-at-a-time HIP path / the CPU oracle at larger sizes.

Tolerances: rendered pixels <= 1e-5 rel-L2 (north_star); gradients <= 2e-4 rel-L2 (fp32 atomics and split-K sums are
order dependent); survivor indices (ray_id, step_id) bit-exact."""
import numpy as np
import pytest
import torch

from conftest import match_survivors, rel_l2

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dev)


def run_step(model, rays, target, lossw):
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import render_losses
    for p in model.parameters():
        p.grad = None
    res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
    loss = render_losses(res, target, lossw, model)
    loss.backward()
    return res, loss


def grads_of(model):
    from fgs_nerf_amd.nerf import mlp_layers
    out = {'sdf': model.sdf.grid.grad, 'k0': model.k0.grid.grad}
    for net in ('rgbnet', 'refnet'):
        for i, l in enumerate(mlp_layers(getattr(model, net))):
            out[f'{net}.{i}.weight'], out[f'{net}.{i}.bias'] = l.weight.grad, l.bias.grad
    return out


def test_fused_is_selected_and_matches_golden(dev, golden):
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.losses import render_losses
    g = golden("e2e_fine.npz")
    model = synth.build_model(16, synth.FINE_MODEL, device=dev)
    assert fused.supports(model)
    rays = (T(g["rays_o"], dev), T(g["rays_d"], dev), T(g["viewdirs"], dev))
    res = model(*rays, global_step=int(g["global_step"]), **synth.RENDER_KWARGS)
    assert isinstance(res, fused.LazyResult)
    assert np.array_equal(res["ray_id"].cpu().numpy(), g["ray_id"]) and np.array_equal(res["step_id"].cpu().numpy(), g["step_id"])
    for key, tol in (("rgb_marched", 1e-5), ("sigmoid_rgb", 1e-5), ("weights", 1e-5), ("raw_rgb", 1e-5), ("normal", 1e-5),
                     ("alphainv_cum", 1e-6), ("raw_alpha", 1e-5)):
        assert rel_l2(res[key], g[key]) < tol, key
    loss = render_losses(res, T(g["target"], dev), synth.FINE_LOSS, model)
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    loss.backward()
    for k, v in grads_of(model).items():
        assert rel_l2(v, g["grad_" + k]) < 2e-4, k
    # lazily evaluated reference entries
    assert res["mask"].dtype == torch.bool and res["mask_outbbox"].shape[0] == int(g["n_total"])


@pytest.mark.parametrize("G,N,extra", [(48, 512, {}), (40, 300, {"mask_cache": True}), (32, 257, {"render": True}),
                                       (32, 300, {"smooth": 3})])
def test_fused_vs_oracle_and_composed(dev, oracle, G, N, extra):
    """Mid-size scenes: the fused path against the CPU oracle (the checker) and against the operator-at-a-time HIP path.
    Gradient tolerances: 1e-3 rel-L2 vs the oracle (measured 1e-4 .. 3e-4: fp32 accumulation order, ReLU-boundary
    flips); the composed path, whose MLP runs through rocBLAS, sits 4e-4 .. 3e-3 from the same oracle."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import render_losses
    from fgs_nerf_amd.nerf import MaskCache
    rays_c = synth.random_rays(N, seed=21)
    rays = tuple(r.to(dev) for r in rays_c)
    target_c = torch.rand(N, 3, generator=torch.Generator().manual_seed(4))
    target = target_c.to(dev)
    lossw = dict(synth.FINE_LOSS, weight_rgbper=0.05)          # also exercises the raw_rgb gradient path
    cfg = dict(synth.FINE_MODEL)
    if extra.get("smooth"):          # smoothing inside the fine stage (model/nerf.py:791): every lookup samples conv(sdf.grid)
        cfg.update(smooth_ksize=extra["smooth"], smooth_sigma=0.8)
    a = synth.build_model(G, cfg, device=dev, fused=True)
    b = synth.build_model(G, cfg, device=dev, fused=False)
    if extra.get("smooth"):
        from fgs_nerf_amd import fused as _f
        assert _f.supports(a) and a.smooth_sdf
    if extra.get("mask_cache"):
        for m in (a, b):
            sdf_mask = ((m.sdf.grid.detach() < 0.25) * 1e-3).float()
            m.mask_cache = MaskCache(path=None, mask_cache_thres=1e-3 * 0.5, sdf_mask=sdf_mask.cpu(),
                                     xyz_min=[-1, -1, -1], xyz_max=[1, 1, 1]).to(dev)
    kw = dict(synth.RENDER_KWARGS)
    rflags = dict(render_grad=True, render_depth=True) if extra.get("render") else {}
    kw.update(rflags)

    def step(model):
        for p in model.parameters():
            p.grad = None
        res = model(*rays, global_step=1000, **kw)
        loss = render_losses(res, target, lossw, model)
        loss.backward()
        return res, loss
    ra, la = step(a)
    rb, lb = step(b)
    # the checker
    P = synth.oracle_params(b)
    leaves = {'sdf': P['sdf'], 'k0': P['k0']}
    for net in ('rgbnet', 'refnet'):
        for i, (W, bias) in enumerate(P[net]):
            leaves[f'{net}.{i}.weight'], leaves[f'{net}.{i}.bias'] = W, bias
    for t in leaves.values():
        t.requires_grad_(True)
    ro = oracle.forward_fine(P, *rays_c, global_step=1000, near=2.0, stepsize=0.5, bg=1, **rflags)
    lo = render_losses(ro, target_c, lossw)
    lo.backward()

    assert torch.equal(ra["ray_id"].cpu(), ro["ray_id"]) and torch.equal(ra["ray_id"], rb["ray_id"])
    if extra.get("mask_cache"):
        assert ra["ray_id"].shape[0] > 0
    for key in ("rgb_marched", "sigmoid_rgb", "weights", "raw_rgb", "normal", "alphainv_cum", "raw_alpha", "gradient"):
        assert rel_l2(ra[key], ro[key]) < 1e-5, key
        assert rel_l2(ra[key], rb[key]) < 1e-5, key
    if extra.get("render"):
        assert rel_l2(ra["normal_marched"], ro["normal_marched"]) < 1e-5 and rel_l2(ra["depth"], ro["depth"]) < 1e-5
    assert abs(float(la) - float(lo)) < 1e-6 and abs(float(la) - float(lb)) < 1e-6
    ga, gb = grads_of(a), grads_of(b)
    for k in ga:
        assert rel_l2(ga[k], leaves[k].grad) < 1e-3, k
        assert rel_l2(gb[k], leaves[k].grad) < 1e-2, k


def test_fused_full_size_vs_oracle_forward(dev, oracle):
    """BASELINE config 2 shape (160^3, 4096 rays): rendered pixels vs the CPU oracle, survivor set identical."""
    from fgs_nerf_amd import synth
    model = synth.build_model(160, synth.FINE_MODEL, device=dev)
    ro, rd, vd = synth.random_rays(4096)
    with torch.no_grad():
        res = model(ro.to(dev), rd.to(dev), vd.to(dev), global_step=1000, **synth.RENDER_KWARGS)
        ref = oracle.forward_fine(synth.oracle_params(model), ro, rd, vd, global_step=1000, near=2.0, stepsize=0.5, bg=1)
    assert res["weights"].shape[0] > 40_000
    ia, ib, _ = match_survivors(res, ref, label="160^3 fine")     # identical, or every difference explained and printed
    assert rel_l2(res["rgb_marched"], ref["rgb_marched"]) < 1e-5
    assert rel_l2(res["alphainv_cum"], ref["alphainv_cum"]) < 1e-5
    assert rel_l2(res["weights"].cpu()[ia], ref["weights"][ib]) < 1e-5
    assert rel_l2(res["raw_rgb"].cpu()[ia], ref["raw_rgb"][ib]) < 1e-5
    # size-independent properties
    w_sum = torch.zeros(4096, device=dev).index_add_(0, res["ray_id"], res["weights"])
    assert float((w_sum + res["alphainv_cum"]).max()) <= 1.0 + 1e-5
    assert bool((res["rgb_marched"] >= 0).all() and (res["rgb_marched"] <= 1).all())
    assert bool((res["ray_id"][1:] >= res["ray_id"][:-1]).all())


def test_gemm_variants(dev):
    from fgs_nerf_amd import fused_ops as fo
    torch.manual_seed(0)
    M, K, N, ld = 777, 108, 256, 108
    X, W, b = torch.randn(M, ld, device=dev), torch.randn(N, ld, device=dev) * 0.1, torch.randn(N, device=dev)
    Y, cs = torch.empty(M, N, device=dev), torch.zeros(N, device=dev)
    fo.gemm(fo.GEMM_NT, X, W, Y, M, N, ld, bias=b, relu=True, colsum=cs)
    ref = torch.relu(X.double() @ W.double().T + b.double())
    assert rel_l2(Y, ref) < 1e-6 and rel_l2(cs, ref.sum(0)) < 1e-5
    dY, act, dX = torch.randn(M, N, device=dev), torch.randn(M, ld, device=dev), torch.empty(M, ld, device=dev)
    fo.gemm(fo.GEMM_NN, dY, W, dX, M, ld, N, mask=act)
    assert rel_l2(dX, (dY.double() @ W.double()) * (act > 0)) < 1e-6
    dW = torch.zeros(N, ld, device=dev)
    fo.gemm(fo.GEMM_TN, dY, X, dW, N, ld, M)
    assert rel_l2(dW, dY.double().T @ X.double()) < 1e-6
    with pytest.raises(RuntimeError, match="multiple"):
        fo.gemm(fo.GEMM_NT, X[:, :106], W[:, :106], Y, M, N, 106)


@pytest.mark.parametrize("M,K,N", [(777, 256, 52), (1, 256, 64), (4099, 256, 4), (129, 64, 108), (513, 32, 36), (640, 256, 256),
                                   (300, 108, 52)])
def test_gemm_store_products_every_instantiation(dev, M, K, N):
    """The store-epilogue products (NT forward, NN data gradient) through each instantiation the launcher picks: K % 32 == 0 ->
    unguarded clamped-operand main loop (rows beyond M and columns beyond N read the operand's last row / last float4), N <= 64
    -> the 4 x (32 rows x 64 columns) wave arrangement, otherwise the guarded loop; bias / ReLU / mask / column sums in the
    epilogue, a partial last row tile, outputs beyond N untouched."""
    from fgs_nerf_amd import fused_ops as fo
    g = torch.Generator().manual_seed(M + K + N)
    ldc = N + 8
    X, W, b = torch.randn(M, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.1).to(dev), torch.randn(N, generator=g).to(dev)
    Y, cs = torch.full((M, ldc), 7.0, device=dev), torch.zeros(N, device=dev)
    fo.gemm(fo.GEMM_NT, X, W, Y[:, :N], M, N, K, bias=b, relu=True, colsum=cs)
    ref = torch.relu(X.double() @ W.double().T + b.double())
    assert rel_l2(Y[:, :N], ref) < 1e-6 and rel_l2(cs, ref.sum(0)) < 1e-5 and bool((Y[:, N:] == 7.0).all())
    # NN: dX[M, N] = dY[M, K] . Wt[K, N], masked
    dY, Wt = torch.randn(M, K, generator=g).to(dev), (torch.randn(K, N, generator=g) * 0.1).to(dev)
    act, dX, cs2 = torch.randn(M, N, generator=g).to(dev), torch.full((M, ldc), 7.0, device=dev), torch.zeros(N, device=dev)
    fo.gemm(fo.GEMM_NN, dY, Wt, dX[:, :N], M, N, K, mask=act, colsum=cs2)
    ref2 = (dY.double() @ Wt.double()) * (act > 0)
    assert rel_l2(dX[:, :N], ref2) < 1e-6 and rel_l2(cs2, ref2.sum(0)) < 1e-5 and bool((dX[:, N:] == 7.0).all())


def test_fused_losses_match_torch_losses(dev):
    """csrc/losses.hip vs the reference statements in torch (model/nerf_training.py:308-327): value and every gradient."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import fused_render_losses, render_losses
    N = 300
    rays = tuple(r.to(dev) for r in synth.random_rays(N, seed=33))
    target = torch.rand(N, 3, generator=torch.Generator().manual_seed(8)).to(dev)
    cfg = dict(weight_main=1.0, weight_rgbper=0.2, weight_entropy_last=0.001, weight_orientation=1e-2, sigmoid_rgb_loss=0.1)
    out = {}
    for name, fn in (("torch", render_losses), ("fused", fused_render_losses)):
        model = synth.build_model(32, synth.FINE_MODEL, device=dev)
        res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
        loss = fn(res, target, cfg, model)
        (loss * 1.7).backward()                      # a non-unit upstream gradient
        out[name] = (float(loss), grads_of(model))
    assert abs(out["torch"][0] - out["fused"][0]) < 1e-6
    for k in out["torch"][1]:
        assert rel_l2(out["fused"][1][k], out["torch"][1][k]) < 1e-4, k


def test_masked_adam_multi_tensor_matches_oracle(dev, oracle):
    from fgs_nerf_amd.adam import MaskedAdam
    rng = np.random.RandomState(12)
    shapes = [(256, 106), (256,), (3, 256), (3,), (17, 5)]
    params = [torch.nn.Parameter(torch.from_numpy(rng.randn(*s).astype(np.float32)).to(dev)) for s in shapes]
    ref = [p.detach().cpu().numpy().copy() for p in params]
    ms = [np.zeros_like(r) for r in ref]
    vs = [np.zeros_like(r) for r in ref]
    opt = MaskedAdam([{'params': params[:3], 'lr': 1e-3, 'name': 'a', 'skip_zero_grad': False},
                      {'params': params[3:], 'lr': 5e-2, 'name': 'b', 'skip_zero_grad': True}])
    for step in (1, 2, 3):
        grads = [(rng.randn(*s) * (rng.rand(*s) > 0.3)).astype(np.float32) for s in shapes]
        for p, g in zip(params, grads):
            p.grad = torch.from_numpy(g).to(dev)
        opt.step()
        for i, g in enumerate(grads):
            oracle.K.adam_upd(ref[i], g, ms[i], vs[i], step, 0.9, 0.99, 1e-3 if i < 3 else 5e-2, 1e-8, mode=0 if i < 3 else 1)
    for p, r in zip(params, ref):
        assert np.array_equal(p.detach().cpu().numpy(), r)


@pytest.mark.parametrize("M,K,N", [(49920, 256, 256), (4100, 108, 256), (4097, 308, 192), (65536, 256, 256),
                                   (12345, 64, 256), (50001, 192, 192), (4096, 32, 256), (200000, 256, 128)])
def test_gemm_stream_k_matches_fp64(dev, M, K, N):
    """The opt-in stream-K form of the NT / NN products (one equal range of chunk-units per resident workgroup, split tiles
    finished by their owner) against fp64, with every epilogue (bias, ReLU, mask, column sums), repeated launches on the
    same workspace (the flag words must come back to zero), ragged M and K that is not a multiple of the 32-chunk."""
    from fgs_nerf_amd import fused_ops as fo
    torch.manual_seed(M % 1000)
    ld = (K + 3) // 4 * 4
    X = torch.randn(M, ld, device=dev)
    X[:, K:] = 0
    W, b = torch.randn(N, ld, device=dev) * 0.1, torch.randn(N, device=dev)
    W[:, K:] = 0
    ref = torch.relu(X.double() @ W.double().T + b.double())
    for rep in range(3):
        Y, cs = torch.full((M, N), float('nan'), device=dev), torch.zeros(N, device=dev)
        fo.gemm(fo.GEMM_NT, X, W, Y, M, N, ld, bias=b, relu=True, colsum=cs, stream_k=True)
        assert rel_l2(Y, ref) < 1e-6 and rel_l2(cs, ref.sum(0)) < 1e-5, rep
    dY, act = torch.randn(M, N, device=dev), torch.randn(M, ld, device=dev)
    refd = (dY.double() @ W.double()) * (act > 0)
    for rep in range(2):
        dX, cs = torch.full((M, ld), float('nan'), device=dev), torch.zeros(ld, device=dev)
        fo.gemm(fo.GEMM_NN, dY, W, dX, M, ld, N, mask=act, colsum=cs, stream_k=True)
        assert rel_l2(dX, refd) < 1e-6 and rel_l2(cs, refd.sum(0)) < 1e-5, rep
    ws = fo.gemm_workspace(X.device)
    n_flags = 4 * 2 * torch.cuda.get_device_properties(dev).multi_processor_count
    assert int(ws[-n_flags:].view(torch.int32).abs().sum()) == 0          # every flag consumed and reset


@pytest.mark.parametrize("M", [1, 63, 64, 65, 1000, 16384, 49920])
def test_mlp_forward_one_launch_is_bit_identical_to_per_layer(dev, M):
    """fgs_mlp_fwd_f32 (all seven 256-wide layers of rgbnet + refnet in one persistent launch, activations resident in
    LDS) against the same layers issued one by one through fgs_gemm_f32: every saved activation bit for bit."""
    from fgs_nerf_amd import fused_ops as fo
    torch.manual_seed(M)
    Ks, relu = [108, 256, 256, 256, 308, 256, 256], [1, 1, 1, 0, 1, 1, 1]
    X0 = torch.randn(M, 108, device=dev)
    Z = torch.randn(M, 308, device=dev)
    Ws = [torch.randn(256, k, device=dev) * 0.05 for k in Ks]
    bs = [torch.randn(256, device=dev) * 0.1 for _ in Ks]
    outs = [torch.empty(M, 256, device=dev) for _ in Ks]
    outs[3] = Z                                                # the last rgbnet layer writes Z[:, :256]
    a = X0
    for i in range(7):
        fo.gemm(fo.GEMM_NT, a, Ws[i], outs[i], M, 256, Ks[i], bias=bs[i], relu=bool(relu[i]))
        a = outs[i]
    ref = [o.clone() for o in outs]
    for o in outs:
        o[:, :256] = float('nan')
    Z[:, 256:] = ref[3][:, 256:]
    fo.mlp_fwd(M, X0, 108, Z[:, 256:], 52, [(Ws[i], Ks[i], bs[i], relu[i], outs[i]) for i in range(7)])
    for i in range(7):
        assert torch.equal(outs[i], ref[i]), i
    x = X0.double()
    for i in range(4):                                         # and against fp64 for the first network
        x = x @ Ws[i].double().T + bs[i].double()
        x = torch.relu(x) if relu[i] else x
    assert rel_l2(outs[3][:, :256], x) < 1e-5


def test_config0_full_frame_render_128(dev, oracle):
    """BASELINE configs[0] shape: a 200x200 frame (40 000 rays, 8192-ray chunks as the reference's render loop feeds them)
    through the 128^3 fine model under no_grad; pixels against the CPU oracle at <= 1e-5 rel-L2."""
    from fgs_nerf_amd import dvgo_ray, synth
    H = W = 200
    model = synth.build_model(128, synth.FINE_MODEL, device=dev)
    K = synth.intrinsics(H, W)
    c2w = torch.from_numpy(synth.look_at_origin(45.0))
    ro, rd, vd = dvgo_ray.get_rays_of_a_view(H, W, K, c2w, ndc=False, inverse_y=False, flip_x=False, flip_y=False)
    ro, rd, vd = (t.reshape(-1, 3).contiguous() for t in (ro, rd, vd))
    P = synth.oracle_params(model)
    out, ref = [], []
    with torch.no_grad():
        for i in range(0, H * W, 8192):
            sl = slice(i, i + 8192)
            out.append(model(ro[sl].to(dev), rd[sl].to(dev), vd[sl].to(dev), global_step=1000, **synth.RENDER_KWARGS)['rgb_marched'])
            ref.append(oracle.forward_fine(P, ro[sl], rd[sl], vd[sl], global_step=1000, near=2.0, stepsize=0.5, bg=1)['rgb_marched'])
    img, img_ref = torch.cat(out).cpu(), torch.cat(ref)
    assert img.shape == (H * W, 3) and rel_l2(img, img_ref) < 1e-5
    assert float((img_ref - 1.0).abs().max()) > 0.05            # the object is in the frame (not an all-background image)


def test_config4_shape_320_grid_forward(dev, oracle):
    """BASELINE configs[4] per-GPU shard: 320^3 grids, 4096 of the 32768 rays; forward pixels and kept-sample set vs the oracle."""
    from fgs_nerf_amd import synth
    model = synth.build_model(320, synth.FINE_MODEL, device=dev)
    ro, rd, vd = synth.random_rays(4096, seed=4)
    with torch.no_grad():
        res = model(ro.to(dev), rd.to(dev), vd.to(dev), global_step=1000, **synth.RENDER_KWARGS)
        ref = oracle.forward_fine(synth.oracle_params(model), ro, rd, vd, global_step=1000, near=2.0, stepsize=0.5, bg=1)
    assert int(res['n_inbbox_visited'].sum()) > 0 and res['weights'].shape[0] > 50_000
    assert rel_l2(res["rgb_marched"], ref["rgb_marched"]) < 1e-5
    assert rel_l2(res["alphainv_cum"], ref["alphainv_cum"]) < 1e-5
    ia, ib, _ = match_survivors(res, ref, label="320^3 fine")
    assert rel_l2(res["weights"].cpu()[ia], ref["weights"][ib]) < 1e-5
    w_sum = torch.zeros(4096, device=dev).index_add_(0, res["ray_id"], res["weights"])
    assert float((w_sum + res["alphainv_cum"]).max()) <= 1.0 + 1e-5


@pytest.mark.parametrize("M", [1, 65, 1000, 49920])
def test_mlp_chain_backward_data_gradients_match_per_layer_gemms(dev, M):
    """fgs_mlp_chain_f32 as the backward data-gradient chain (ReLU masks, column sums, narrow last layer, transposed weights
    from fgs_transpose_multi) against the same products issued one by one through fgs_gemm_f32 (GEMM_NN): every dX bit for
    bit, column sums (atomics, order dependent) to 1e-5."""
    from fgs_nerf_amd import fused_ops as fo
    torch.manual_seed(M + 7)
    dY = torch.randn(M, 256, device=dev)
    Ws = [torch.randn(256, 256, device=dev) * 0.05 for _ in range(3)] + [torch.randn(256, 108, device=dev) * 0.05]
    big = torch.randn(256, 308, device=dev) * 0.05
    Ws[1] = big[:, :256]                                       # a strided weight (ld 308), like refnet layer 0
    masks = [torch.relu(torch.randn(M, 256, device=dev)) for _ in range(2)] + [None, None]
    WT = fo.transpose_multi(Ws)
    for w, wt in zip(Ws, WT):
        assert torch.equal(wt, w.t())
    # reference: one GEMM per layer
    ref_out, ref_cs, a = [], [], dY
    for i, W in enumerate(Ws):
        k_in = W.shape[1]
        out = torch.empty(M, k_in, device=dev)
        cs = torch.zeros(256, device=dev) if i < 3 else None
        fo.gemm(fo.GEMM_NN, a, W, out, M, k_in, 256, mask=masks[i], colsum=cs)
        ref_out.append(out); ref_cs.append(cs); a = out
    outs = [torch.full((M, 256), float('nan'), device=dev) for _ in range(3)] + [torch.full((M, 108), float('nan'), device=dev)]
    css = [torch.zeros(256, device=dev) for _ in range(3)] + [None]
    layers = [dict(W=WT[i], K=256, mask=masks[i], colsum=css[i], out=outs[i]) for i in range(3)]
    layers.append(dict(W=WT[3], K=256, n_rows=108, n_store=108, out=outs[3]))
    fo.mlp_chain(M, dY, 256, layers)
    for i in range(4):
        assert torch.equal(outs[i], ref_out[i]), i
        if css[i] is not None:
            assert rel_l2(css[i], ref_cs[i]) < 1e-5, i


def test_backward_chain_mode_matches_default_mode(dev, monkeypatch):
    """FGS_LINEAR_BWD=chain (one persistent launch for all data gradients) against the default per-layer k_linear_bwd
    launches on the same step: identical forward, gradients within the split-K / atomics order tolerance."""
    from fgs_nerf_amd import fused, synth
    rays = tuple(r.to(dev) for r in synth.random_rays(700, seed=5))
    target = torch.rand(700, 3, generator=torch.Generator().manual_seed(9)).to(dev)
    lossw = dict(synth.FINE_LOSS, weight_rgbper=0.05)
    got = {}
    monkeypatch.setattr(fused, "_MLP_IMPL", "lds")           # these are modes of the LDS-resident / per-layer GEMM path
    for mode in ("one", "chain"):
        monkeypatch.setattr(fused, "_LINEAR_BWD_MODE", mode)
        model = synth.build_model(48, synth.FINE_MODEL, device=dev)
        res, loss = run_step(model, rays, target, lossw)
        got[mode] = (res["rgb_marched"].clone(), float(loss), {k: v.clone() for k, v in grads_of(model).items()})
    assert torch.equal(got["one"][0], got["chain"][0]) and got["one"][1] == got["chain"][1]
    for k, v in got["one"][2].items():
        assert rel_l2(got["chain"][2][k], v) < 2e-5, k


def test_fused_full_size_backward_properties(dev):
    """BASELINE config 2 shape (160^3, 4096 rays), backward pass, through properties that need no oracle run of that size:
    gradients are linear in the upstream gradient (2 x loss -> 2 x every gradient, to fp32 summation-order noise); the
    integer outputs and the loss repeat bit for bit; k0.grad is supported only on the 8 trilinear corners of the survivors
    (the occupancy the multi-GPU exchange relies on); the channels of a touched k0 voxel are all written."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import render_losses
    model = synth.build_model(160, synth.FINE_MODEL, device=dev)
    rays = tuple(t.to(dev) for t in synth.random_rays(4096))
    target = torch.rand(4096, 3, generator=torch.Generator().manual_seed(12)).to(dev)
    lossw = dict(synth.FINE_LOSS, weight_rgbper=0.05)

    def step(scale):
        for p in model.parameters():
            p.grad = None
        res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
        loss = render_losses(res, target, lossw, model)
        (loss * scale).backward()
        return res, float(loss), {k: v.clone() for k, v in grads_of(model).items()}
    r1, l1, g1 = step(1.0)
    r2, l2, g2 = step(2.0)
    assert torch.equal(r1["ray_id"], r2["ray_id"]) and torch.equal(r1["step_id"], r2["step_id"]) and l1 == l2
    assert torch.equal(r1["rgb_marched"], r2["rgb_marched"])
    for k in g1:
        assert rel_l2(g2[k], 2.0 * g1[k]) < 2e-5, k
        assert bool(torch.isfinite(g1[k]).all()), k
    # support of k0.grad: corners of the survivors' cells
    pts = r1["survivor_pts"]
    idx = (pts - model.xyz_min) / (model.xyz_max - model.xyz_min) * 159.0
    base = idx.floor().long().clamp(0, 158)
    touched = torch.zeros(160, 160, 160, dtype=torch.bool, device=dev)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                touched[base[:, 0] + dx, base[:, 1] + dy, base[:, 2] + dz] = True
    nz = (g1["k0"][0] != 0)
    assert not bool((nz.any(dim=0) & ~touched).any())
    assert int(nz.any(dim=0).sum()) > 100_000 and int(touched.sum()) < 2 * int(nz.any(dim=0).sum())


@pytest.mark.parametrize("stage", ["fine", "coarse"])
def test_two_forwards_before_their_backwards_keep_their_own_records(dev, stage):
    """A second forward with the same ray count before the first backward (loss over two batches, gradient accumulation,
    a no_grad validation render while the graph is alive) must not overwrite the march records the first backward
    re-reads: each forward owns its record set until its backward ran (fused._workspace).  Checked against the same two
    batches run one after the other."""
    from fgs_nerf_amd import synth
    from fgs_nerf_amd.losses import render_losses
    from fgs_nerf_amd.nerf import mlp_layers
    cfg, lossw = (synth.FINE_MODEL, synth.FINE_LOSS) if stage == "fine" else (synth.COARSE_MODEL, synth.COARSE_LOSS)
    N = 300
    batches = []
    for seed in (5, 6):
        rays = tuple(r.to(dev) for r in synth.random_rays(N, seed=seed))
        batches.append((rays, torch.rand(N, 3, generator=torch.Generator().manual_seed(seed)).to(dev)))

    def grads(model):
        out = [model.sdf.grid.grad.clone(), model.k0.grid.grad.clone()]
        return out + [p.grad.clone() for p in model.refnet.parameters()]

    # reference: one after the other, gradients accumulated by autograd
    a = synth.build_model(32, cfg, device=dev)
    for rays, target in batches:
        render_losses(a(*rays, global_step=1000, **synth.RENDER_KWARGS), target, lossw, a).backward()
    ref = grads(a)
    # interleaved: forward 1, forward 2 (+ a no_grad render of the same size in between), then the sum's backward
    b = synth.build_model(32, cfg, device=dev)
    r1 = b(*batches[0][0], global_step=1000, **synth.RENDER_KWARGS)
    with torch.no_grad():
        b(*batches[1][0], global_step=1000, **synth.RENDER_KWARGS)
    r2 = b(*batches[1][0], global_step=1000, **synth.RENDER_KWARGS)
    (render_losses(r1, batches[0][1], lossw, b) + render_losses(r2, batches[1][1], lossw, b)).backward()
    for x, y in zip(grads(b), ref):
        assert rel_l2(x, y) < 2e-5
    # and a second backward through the same forward is refused loudly instead of reading released records
    r3 = b(*batches[0][0], global_step=1000, **synth.RENDER_KWARGS)
    l3 = render_losses(r3, batches[0][1], lossw, b)
    l3.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="backward called twice"):
        l3.backward()


@pytest.mark.parametrize("G,N", [(48, 700), (32, 33)])
def test_register_resident_mlp_path_matches_the_gemm_path(dev, monkeypatch, G, N):
    """FGS_MLP=rc (default: register-resident forward / backward chains + one-launch weight gradients) against FGS_MLP=lds
    (LDS-resident forward chain, one k_linear_bwd per layer) on the same step.  The two sum every dot product in a different
    order, so outputs agree to fp32 rounding (1e-6), gradients to the order / ReLU-flip tolerance of the other mode tests."""
    from fgs_nerf_amd import fused, synth
    rays = tuple(r.to(dev) for r in synth.random_rays(N, seed=5))
    target = torch.rand(N, 3, generator=torch.Generator().manual_seed(9)).to(dev)
    lossw = dict(synth.FINE_LOSS, weight_rgbper=0.05)
    got = {}
    for impl in ("rc", "lds"):
        monkeypatch.setattr(fused, "_MLP_IMPL", impl)
        model = synth.build_model(G, synth.FINE_MODEL, device=dev)
        res, loss = run_step(model, rays, target, lossw)
        got[impl] = (res["rgb_marched"].clone(), res["raw_rgb"].clone(), float(loss), {k: v.clone() for k, v in grads_of(model).items()})
    assert rel_l2(got["rc"][0], got["lds"][0]) < 1e-6 and rel_l2(got["rc"][1], got["lds"][1]) < 2e-6
    assert abs(got["rc"][2] - got["lds"][2]) < 1e-6
    for k, v in got["lds"][3].items():
        assert rel_l2(got["rc"][3][k], v) < 1e-4, (k, rel_l2(got["rc"][3][k], v))


@pytest.mark.parametrize("stage", ["fine", "coarse"])
def test_forked_weight_gradient_launch_matches_one_stream(dev, monkeypatch, stage):
    """fused._wgrad: the weight-gradient launch on a side stream beside the scatter kernels of the backward pass (default
    on one GPU) against the same step on ONE stream: identical forward, every gradient within the float-atomics order
    tolerance -- and two forwards followed by their two backwards (two weight-gradient launches pending on the side stream
    at once, the tensors they read kept alive until the join) give the same gradients as separate steps."""
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.losses import fused_render_losses
    cfg, lossw = (synth.FINE_MODEL, synth.FINE_LOSS) if stage == "fine" else (synth.COARSE_MODEL, synth.COARSE_LOSS)
    rays = [tuple(r.to(dev) for r in synth.random_rays(600, seed=s)) for s in (5, 6)]
    target = torch.rand(600, 3, generator=torch.Generator().manual_seed(9)).to(dev)
    got = {}
    for fork in (True, False):
        monkeypatch.setattr(fused, "_WGRAD_FORK", fork)
        model = synth.build_model(40, cfg, device=dev)
        res = model(*rays[0], global_step=1000, **synth.RENDER_KWARGS)
        loss = fused_render_losses(res, target, lossw, model)
        loss.backward()
        torch.cuda.synchronize()
        assert not fused._SIDE_PENDING                      # joined before the backward pass returned
        got[fork] = (res["rgb_marched"].detach().clone(), {k: v.clone() for k, v in grads_of_any(model).items()})
    assert torch.equal(got[True][0], got[False][0])
    for k, v in got[False][1].items():
        assert rel_l2(got[True][1][k], v) < 2e-5, k
    # two forwards, then their backwards (sum of the two losses): gradients == sum of the separate steps' gradients
    monkeypatch.setattr(fused, "_WGRAD_FORK", True)
    model = synth.build_model(40, cfg, device=dev)
    sep = []
    for r in rays:
        for p in model.parameters():
            p.grad = None
        fused_render_losses(model(*r, global_step=1000, **synth.RENDER_KWARGS), target, lossw, model).backward()
        sep.append({k: v.clone() for k, v in grads_of_any(model).items()})
    for p in model.parameters():
        p.grad = None
    l0 = fused_render_losses(model(*rays[0], global_step=1000, **synth.RENDER_KWARGS), target, lossw, model)
    l1 = fused_render_losses(model(*rays[1], global_step=1000, **synth.RENDER_KWARGS), target, lossw, model)
    (l0 + l1).backward()
    torch.cuda.synchronize()
    both = grads_of_any(model)
    for k in both:
        assert rel_l2(both[k], sep[0][k] + sep[1][k]) < 2e-5, k


def grads_of_any(model):
    from fgs_nerf_amd.nerf import mlp_layers
    out = {'sdf': model.sdf.grid.grad, 'k0': model.k0.grid.grad}
    for net in ('rgbnet', 'refnet'):
        if getattr(model, net, None) is not None:
            for i, l in enumerate(mlp_layers(getattr(model, net))):
                out[f'{net}.{i}.weight'], out[f'{net}.{i}.bias'] = l.weight.grad, l.bias.grad
    return out


@pytest.mark.parametrize("stage", ["fine", "coarse"])
def test_learnable_s_val_through_the_fused_path(dev, oracle, stage):
    """s_learn (model/nerf.py:512-522): s_val is a trained parameter and inv_s = 1 / s_val carries a gradient.  The fused march
    backward accumulates d loss / d inv_s (fgs_march_*_bwd g_inv_s); values and every gradient -- s_val's included -- against
    the CPU oracle run with the same parameter."""
    from fgs_nerf_amd import fused, synth
    from fgs_nerf_amd.losses import render_losses
    from fgs_nerf_amd.nerf import mlp_layers
    cfg = synth.FINE_MODEL if stage == "fine" else synth.COARSE_MODEL
    lossw = synth.FINE_LOSS if stage == "fine" else synth.COARSE_LOSS
    model = synth.build_model(32, cfg, device=dev, s_learn=True)
    with torch.no_grad():
        model.s_val.fill_(0.043)
    assert model.s_val.requires_grad and (fused.supports(model) if stage == "fine" else fused.supports_coarse(model))
    N = 300
    ro, rd, vd = synth.random_rays(N, seed=21)
    target = torch.rand(N, 3, generator=torch.Generator().manual_seed(5))
    res = model(ro.to(dev), rd.to(dev), vd.to(dev), global_step=500, **synth.RENDER_KWARGS)
    render_losses(res, target.to(dev), lossw, model).backward()
    P = synth.oracle_params(model)
    P['s_param'] = model.s_val.detach().cpu().clone().requires_grad_(True)
    P['sdf'].requires_grad_(True)
    fwd = oracle.forward_fine if stage == "fine" else oracle.forward_coarse
    ref = fwd(P, ro, rd, vd, global_step=500, near=2.0, stepsize=0.5, bg=1)
    render_losses(ref, target, lossw).backward()
    assert torch.equal(res['ray_id'].cpu(), ref['ray_id'])
    assert rel_l2(res['rgb_marched'].detach(), ref['rgb_marched'].detach()) < 1e-5
    assert rel_l2(model.sdf.grid.grad, P['sdf'].grad) < 1e-3
    g_hip, g_ref = float(model.s_val.grad), float(P['s_param'].grad)
    assert g_ref != 0.0 and abs(g_hip - g_ref) <= 2e-4 * abs(g_ref), (g_hip, g_ref)


def test_fused_loss_scalar_is_deterministic_and_matches_the_torch_form(dev):
    """fgs_fine_loss_fwd with its scratch buffer: per-block sums added in a fixed order by the last block to arrive -- the same
    bits on every call (the atomic form differed in the last place from run to run), equal to losses.render_losses (the torch
    statement of model/nerf_training.py:308-327) to float32 rounding, the counter word left zero, and the atomic fallback (no
    scratch) within an ulp of it."""
    import ctypes
    from fgs_nerf_amd import losses, synth
    from fgs_nerf_amd._lib import call, ptr, stream
    model = synth.build_model(48, synth.FINE_MODEL, device=dev)
    ro, rd, vd = (t.to(dev) for t in synth.random_rays(2048, seed=9))
    target = torch.rand(2048, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    with torch.no_grad():
        res = model(ro, rd, vd, global_step=1000, **synth.RENDER_KWARGS)
    vals = [losses.fused_render_losses(res, target, synth.FINE_LOSS, model) for _ in range(6)]
    torch.cuda.synchronize()
    assert all(torch.equal(v, vals[0]) for v in vals)
    ref = losses.render_losses(res, target, synth.FINE_LOSS, model)
    assert abs(float(vals[0]) - float(ref)) <= 2e-6 * abs(float(ref))
    assert int(losses._SCRATCH[dev.index].view(torch.int32)[0]) == 0
    # the form without a scratch buffer (memset + atomics)
    N, M = res['rgb_marched'].shape[0], res['weights'].shape[0]
    args = (res['rgb_marched'], res['sigmoid_rgb'], target, res['alphainv_cum'], res['weights'], res['normal'], res['raw_rgb'],
            res['ray_id'], res['ray_viewdirs'])
    out = torch.full((), 7.0, device=dev)
    call("fgs_fine_loss_fwd", N, M, *(ptr(a.contiguous()) for a in args), losses._w5(synth.FINE_LOSS), ptr(out), None, 0, None, stream())
    assert abs(float(out) - float(vals[0])) <= 2e-6 * abs(float(vals[0]))
