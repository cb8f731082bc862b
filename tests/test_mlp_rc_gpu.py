"""GPU tests of the register-resident MLP chains (csrc/mlp_rc.hip, fgs_mlp_rc_chain) against float64 torch: forward
(bias, ReLU, appended columns, saved activations, ReLU sign bits), backward data gradients (transposed images, masks from
the forward's bits, narrow and 308-wide outputs), ragged M, padding columns holding NaN, coarse-stage widths, and the
device-side row count (fgs_dyn_t.row_count)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _fine_setup(M, dev, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    X0 = r(M, 108).to(dev)
    X0[:, 106:] = float('nan')                       # padding columns of the operand buffer: must be ignored
    Z = torch.full((M, 308), float('nan'), device=dev)
    Z[:, 256:307] = r(M, 51).to(dev)
    Ws = [r(256, 106) * 0.1, r(256, 256) * 0.06, r(256, 256) * 0.06, r(256, 256) * 0.06, r(256, 307) * 0.06,
          r(256, 256) * 0.06, r(256, 256) * 0.06]
    Ws = [w.to(dev) for w in Ws]
    bs = [(r(256) * 0.1).to(dev) for _ in Ws]
    relu = [1, 1, 1, 0, 1, 1, 1]
    return X0, Z, Ws, bs, relu


def _forward(M, X0, Z, Ws, bs, relu, dev, cap=None, rows_dev=None):
    from fgs_nerf_amd import fused_ops as fo
    cap = cap or M
    outs = [torch.full((cap, 256), float('nan'), device=dev) for _ in Ws]
    outs[3] = Z                                       # the last rgbnet layer writes Z[:, :256]
    bits = [fo.rc_mask_bits(cap, dev) if relu[i] else None for i in range(7)]
    layers = []
    for i in range(7):
        L = dict(W=Ws[i], bias=bs[i], relu=relu[i], mask_bits=bits[i], out=outs[i], n_store=256)
        if i == 4:
            L.update(ext=Z[:, 256:], ext_cols=52)
        layers.append(L)
    fo.rc_chain(False, cap, X0, 108, layers, rows_dev=rows_dev)
    return outs, bits


def _reference_forward(X0, Z, Ws, bs, relu):
    x = X0[:, :106].double()
    acts = []
    for i in range(7):
        if i == 4:
            x = torch.cat([x, Z[:, 256:307].double()], 1)
        x = x @ Ws[i].double().T + bs[i].double()
        if relu[i]:
            x = torch.relu(x)
        acts.append(x)
    return acts


def _unpack_bits(bits, M):
    """[groups][64 lanes] x 4 words -> bool [M][256] in (sample, feature) order (the kernel's register layout)."""
    w = bits.view(-1, 64, 4).cpu().to(torch.int64) & 0xffffffff
    G = w.shape[0]
    out = torch.zeros(G * 32, 256, dtype=torch.bool)
    for t in range(8):
        for r in range(16):
            word, sh = t >> 1, 31 - ((t & 1) * 16 + r)           # element e of a word sits at bit 31 - e (v_alignbit shift-in)
            b = ((w[:, :, word] >> sh) & 1).bool()            # [G][64]
            for h in range(2):
                f = 32 * t + 8 * (r >> 2) + 4 * h + (r & 3)
                out[:, f] = b[:, 32 * h:32 * h + 32].reshape(-1)
    return out[:M]


@pytest.mark.parametrize("M", [1, 31, 33, 128, 129, 1000, 50001])
def test_rc_forward_matches_fp64(dev, M):
    X0, Z, Ws, bs, relu = _fine_setup(M, dev, seed=M)
    outs, bits = _forward(M, X0, Z, Ws, bs, relu, dev)
    ref = _reference_forward(X0, Z, Ws, bs, relu)
    for i in range(7):
        got = outs[i][:, :256]
        assert bool(torch.isfinite(got).all()), i
        assert rel_l2(got, ref[i]) < 2e-6, (i, rel_l2(got, ref[i]))
    assert bool(torch.isfinite(Z[:, 256:307]).all()) and bool(torch.isnan(Z[:, 307]).all())   # appended columns untouched
    for i in (0, 1, 2, 4, 5, 6):
        assert torch.equal(_unpack_bits(bits[i], M), (outs[i][:, :256] > 0).cpu()), i


@pytest.mark.parametrize("M", [1, 100, 4097])
def test_rc_backward_matches_fp64(dev, M):
    from fgs_nerf_amd import fused_ops as fo
    X0, Z, Ws, bs, relu = _fine_setup(M, dev, seed=M + 3)
    outs, bits = _forward(M, X0, Z, Ws, bs, relu, dev)
    g = torch.Generator().manual_seed(M)
    dY = torch.randn(M, 256, generator=g).to(dev)
    # backward chain: layers 6 .. 1 (every product 256 columns wide); the output of step i is masked by the ReLU of layer
    # i - 1 (none below layer 4).  The two narrow products -- the 52 reflection-encoding columns of dZ and dX0 (108 columns)
    # -- are not part of the chain (one output width per launch): fgs_gemm_f32 on the dY tensors the chain wrote out.
    d5, d4, d3, d2, d1, d0 = (torch.full((M, 256), float('nan'), device=dev) for _ in range(6))
    layers = [dict(W=Ws[6], mask_bits=bits[5], out=d5, n_store=256), dict(W=Ws[5], mask_bits=bits[4], out=d4, n_store=256),
              dict(W=Ws[4][:, :256], out=d3, n_store=256), dict(W=Ws[3], mask_bits=bits[2], out=d2, n_store=256),
              dict(W=Ws[2], mask_bits=bits[1], out=d1, n_store=256), dict(W=Ws[1], mask_bits=bits[0], out=d0, n_store=256)]
    fo.rc_chain(True, M, dY, 256, layers)
    acts = [o[:, :256].double() for o in outs]
    gcur = dY.double()
    refs = []
    for i in range(6, 0, -1):
        gcur = gcur @ Ws[i].double()[:, :256]
        if i in (6, 5, 3, 2, 1):
            gcur = gcur * (acts[i - 1] > 0)
        refs.append(gcur)
    for k, (got, ref) in enumerate(zip((d5, d4, d3, d2, d1, d0), refs)):
        assert rel_l2(got, ref) < 2e-6, (k, rel_l2(got, ref))


@pytest.mark.parametrize("width,n_in,M", [(192, 90, 777), (128, 72, 513)])
def test_rc_coarse_widths(dev, width, n_in, M):
    """The coarse (90 -> 192 -> 192) and geometry_searching (72 -> 128 -> 128) refnet trunks (config/shiny_blender.py:90-92,
    163-167): widths that are not 256, forward and backward."""
    from fgs_nerf_amd import fused_ops as fo
    g = torch.Generator().manual_seed(width)
    ld = (n_in + 3) // 4 * 4
    X0 = torch.randn(M, ld, generator=g).to(dev)
    X0[:, n_in:] = float('nan')
    W0, W1 = (torch.randn(width, n_in, generator=g) * 0.1).to(dev), (torch.randn(width, width, generator=g) * 0.08).to(dev)
    b0, b1 = (torch.randn(width, generator=g) * 0.1).to(dev), (torch.randn(width, generator=g) * 0.1).to(dev)
    o0, o1 = (torch.full((M, width), float('nan'), device=dev) for _ in range(2))
    m0, m1 = fo.rc_mask_bits(M, dev), fo.rc_mask_bits(M, dev)
    fo.rc_chain(False, M, X0, ld, [dict(W=W0, bias=b0, relu=1, mask_bits=m0, out=o0, n_store=width),
                                   dict(W=W1, bias=b1, relu=1, mask_bits=m1, out=o1, n_store=width)])
    r0 = torch.relu(X0[:, :n_in].double() @ W0.double().T + b0.double())
    r1 = torch.relu(r0 @ W1.double().T + b1.double())
    assert rel_l2(o0, r0) < 2e-6 and rel_l2(o1, r1) < 2e-6
    dY = torch.randn(M, width, generator=g).to(dev)
    d0 = torch.full((M, width), float('nan'), device=dev)
    fo.rc_chain(True, M, dY, width, [dict(W=W1, mask_bits=m0, out=d0, n_store=width)])
    g0 = (dY.double() @ W1.double()) * (r0 > 0)
    assert rel_l2(d0, g0) < 2e-6


def test_rc_device_row_count(dev):
    """fgs_dyn_t.row_count: the host count is only the capacity; rows beyond the device count are never written, rows
    below it are bit-identical to a plain launch with that count."""
    cap, M = 3000, 1777
    X0, Z, Ws, bs, relu = _fine_setup(cap, dev, seed=5)
    outs_a, _ = _forward(M, X0[:M].contiguous(), Z[:M].clone(), Ws, bs, relu, dev)
    count = torch.tensor([M], dtype=torch.int64, device=dev)
    Zb = Z.clone()
    outs_b, _ = _forward(M, X0, Zb, Ws, bs, relu, dev, cap=cap, rows_dev=count.data_ptr())
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a[:M, :256], b[:M, :256])
        assert bool(torch.isnan(b[M:, :256]).all())


@pytest.mark.parametrize("M", [1, 2, 63, 1000, 50001])
def test_wgrad_all_layers_one_launch_matches_fp64(dev, M):
    """fgs_mlp_wgrad: the seven weight gradients and bias gradients of the fine-stage MLPs in one launch (dW += dY^T X with
    the samples split over the chip, fp32 atomics) against float64; ragged M, the 106- and 307-column inputs (column blocks of
    128, 256 + 64), strided dY (dZ's first 256 columns), padding columns untouched."""
    from fgs_nerf_amd import fused_ops as fo
    g = torch.Generator().manual_seed(M)
    n_in = [106, 256, 256, 256, 307, 256, 256]
    ld_x = [108, 256, 256, 256, 308, 256, 256]
    Xs = [torch.randn(M, ld, generator=g).to(dev) for ld in ld_x]
    dZ = torch.randn(M, 308, generator=g).to(dev)
    dYs = [torch.randn(M, 256, generator=g).to(dev) for _ in range(7)]
    dYs[3] = dZ[:, :256]                                   # a strided dY (ld 308)
    dWs = [torch.zeros(256, ld, device=dev) for ld in ld_x]
    dbs = [torch.zeros(256, device=dev) for _ in range(7)]
    fo.mlp_wgrad(M, [(dYs[i], Xs[i], dWs[i], dbs[i], 256, n_in[i]) for i in range(7)])
    for i in range(7):
        ref = dYs[i].double().T @ Xs[i][:, :n_in[i]].double()
        assert rel_l2(dWs[i][:, :n_in[i]], ref) < 2e-6, (i, rel_l2(dWs[i][:, :n_in[i]], ref))
        assert float(dWs[i][:, n_in[i]:].abs().max()) == 0.0 if n_in[i] < ld_x[i] else True
        assert rel_l2(dbs[i], dYs[i].double().sum(0)) < 2e-6, i
    # coarse-stage widths: 192 x 90 and 192 x 192, accumulating into a non-zero dW
    W = 192
    X0, X1 = torch.randn(M, 92, generator=g).to(dev), torch.randn(M, W, generator=g).to(dev)
    d0, d1 = torch.randn(M, W, generator=g).to(dev), torch.randn(M, W, generator=g).to(dev)
    g0, g1 = torch.ones(W, 92, device=dev), torch.zeros(W, W, device=dev)
    b1 = torch.zeros(W, device=dev)
    fo.mlp_wgrad(M, [(d0, X0, g0, None, W, 90), (d1, X1, g1, b1, W, W)])
    assert rel_l2(g0[:, :90], 1.0 + d0.double().T @ X0[:, :90].double()) < 2e-6
    assert rel_l2(g1, d1.double().T @ X1.double()) < 2e-6 and rel_l2(b1, d1.double().sum(0)) < 2e-6
    # every wave arrangement of the narrow layers: n_out 192 with 307 columns (blocks of 192 + 115), n_out 128 with 200
    # columns (128 + 72) and with 39 columns (geometry_searching's first layer), n_out 160 with 64 columns, n_out 100
    for n_out, n_in in [(192, 307), (128, 200), (128, 39), (160, 64), (100, 128)]:
        ld = (n_in + 3) // 4 * 4
        X = torch.randn(M, ld, generator=g).to(dev)
        dY = torch.randn(M, n_out, generator=g).to(dev)
        dW, db = torch.zeros(n_out, ld, device=dev), torch.zeros(n_out, device=dev)
        fo.mlp_wgrad(M, [(dY, X, dW, db, n_out, n_in)])
        assert rel_l2(dW[:, :n_in], dY.double().T @ X[:, :n_in].double()) < 2e-6, (n_out, n_in)
        assert float(dW[:, n_in:].abs().max()) == 0.0 if n_in < ld else True
        assert rel_l2(db, dY.double().sum(0)) < 2e-6, (n_out, n_in)


def test_pad_cols_multi_matches_torch_pad(dev):
    """fgs_pad_cols_multi: the K-padded first-layer weights of both MLPs in one launch == F.pad per matrix (views with a
    row pitch larger than their width included)."""
    from fgs_nerf_amd import fused_ops as fo
    torch.manual_seed(3)
    big = torch.randn(256, 400, device=dev)
    mats = [torch.randn(256, 106, device=dev), big[:, 7:314], torch.randn(3, 5, device=dev)]
    widths = [108, 308, 8]
    outs = fo.pad_cols_multi(mats, widths)
    for w, width, o in zip(mats, widths, outs):
        assert torch.equal(o, torch.nn.functional.pad(w, (0, width - w.shape[1])))
    # the gather form: column ranges of one source written side by side into caller-provided column slices of one tensor (the
    # first-layer weights without the encodings' columns), in the same launch as an ordinary padded copy; nothing outside the
    # slices is touched (canary rows / columns around the destination)
    W = torch.randn(256, 106, device=dev)
    canvas = torch.full((258, 60), 7.0, device=dev)
    dst = canvas[1:257, 4:56]                                  # [256, 52] inside the canvas, pitch 60
    o = fo.pad_cols_multi([W[:, :12], W[:, 66:], big[:, 7:314]], [12, 40, 308], outs=[dst[:, :12], dst[:, 12:], None])
    assert torch.equal(dst, torch.cat([W[:, :12], W[:, 66:]], 1))
    assert torch.equal(o[2], torch.nn.functional.pad(big[:, 7:314], (0, 1)))
    canvas[1:257, 4:56] = 7.0
    assert bool((canvas == 7.0).all())
    with pytest.raises(RuntimeError):                          # a destination whose shape is not [rows, width] is refused
        fo.pad_cols_multi([W[:, :12]], [12], outs=[dst[:, :13]])


def test_canary_catches_the_old_pad_cols_index_expression(dev):
    """Up to commit a5aee36 the kernel wrote dst[e] for e < rows * ld_dst -- right for a destination that IS a [rows, ld_dst]
    matrix, the only kind it was given.  The compact-dX0 change (b4fb5c1) began to pass column slices of a wider tensor
    (W0c[:, :12], W0c[:, 12:52]); on its first development snapshot the old expression was still in place, and the full GPU
    suite of that snapshot died with SIGABRT in the first fused forward pass (gpurun_out/abort.log).  This
    test runs the old expression (kept as fgs_debug_pad_cols_old_indexing) on the canary layout of
    test_pad_cols_multi_matches_torch_pad and shows that the canary check fails on it: the run of rows * ld_dst floats covers
    the canary columns of every row and ends 4 floats + one row behind the slice."""
    from fgs_nerf_amd._lib import call, ptr, stream
    torch.manual_seed(3)
    W = torch.randn(256, 106, device=dev)
    canvas = torch.full((258, 60), 7.0, device=dev)
    dst = canvas[1:257, 4:56]                                  # [256, 52] inside the canvas, pitch 60
    src = W[:, 66:]                                            # 40 columns -> dst[:, 12:]
    call("fgs_debug_pad_cols_old_indexing", ptr(src), 256, 40, src.stride(0), ptr(dst[:, 12:]), dst.stride(0), stream())
    inside = torch.zeros_like(canvas, dtype=torch.bool)
    inside[1:257, 16:56] = True
    overwritten = int(((canvas != 7.0) & ~inside).sum())
    assert overwritten > 256 * 10                               # canary columns of every row: what the canary test asserts against
    # and the current entry point leaves every canary cell alone on the same destination
    canvas.fill_(7.0)
    from fgs_nerf_amd import fused_ops as fo
    fo.pad_cols_multi([src], [40], outs=[dst[:, 12:]])
    assert int(((canvas != 7.0) & ~inside).sum()) == 0 and torch.equal(dst[:, 12:], src)


def test_in_kernel_stamps_time_the_launches_and_change_nothing(dev):
    """fgs_dyn_t.stamps: the chain, weight-gradient and tiled-product launches write wall-clock readings of their workgroups into
    the slot a device-side counter selects (bench.py times the launches of a captured step with them).  Same results as without
    them, one plausible duration per launch and slot, nothing written outside the launch's region."""
    from fgs_nerf_amd import fused_ops as fo
    M = 30000
    X0, Z, Ws, bs, relu = _fine_setup(M, dev)
    ref_outs, _ = _forward(M, X0, Z.clone(), Ws, bs, relu, dev)
    dY, Xa = torch.randn(M, 256, device=dev), torch.randn(M, 256, device=dev)
    dW_ref, db_ref = torch.zeros(256, 256, device=dev), torch.zeros(256, device=dev)
    fo.mlp_wgrad(M, [(dY, Xa, dW_ref, db_ref, 256, 256)])
    Wn = torch.randn(256, 52, device=dev) * 0.05
    C_ref = torch.empty(M, 52, device=dev)
    fo.gemm(fo.GEMM_NN, dY, Wn, C_ref, M, 52, 256)

    SLOTS = 3
    buf = torch.zeros(SLOTS, fo.STAMP_LAUNCHES, fo.STAMP_WORDS, dtype=torch.int64, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    fo.STAMPS.update(buf=buf, counter=counter)
    try:
        for step in range(SLOTS):
            counter.fill_(step + 7)                       # slot (step + 7) % 3: every slot once
            fo.stamps_begin_step()
            Z2 = Z.clone()
            outs, _ = _forward(M, X0, Z2, Ws, bs, relu, dev)
            dW, db = torch.zeros(256, 256, device=dev), torch.zeros(256, device=dev)
            fo.mlp_wgrad(M, [(dY, Xa, dW, db, 256, 256)], flop=2.0 * M * 256 * 256)
            C = torch.empty(M, 52, device=dev)
            fo.gemm(fo.GEMM_NN, dY, Wn, C, M, 52, 256, stamp=("k_gemm", 2.0 * M * 52 * 256))
        torch.cuda.synchronize()
        launches = list(fo.STAMPS["launches"])
        rec = fo.stamps_read()
    finally:
        fo.STAMPS.update(buf=None, counter=None)
    assert [k for _, _, k in launches] == ["rc", "wgrad", "gemm"]
    for (label, flop, durs), lo in zip(rec, (20e-6, 10e-6, 3e-6)):
        assert len(durs) == SLOTS, (label, durs)
        assert all(lo < d < 5e-3 for d in durs), (label, durs)
    assert int((buf[:, 3:] != 0).sum()) == 0                          # launch regions 3.. untouched
    assert int((buf[:, 2, 2:] != 0).sum()) == 0                       # the tiled product writes two words
    for a, b in zip(outs, ref_outs):
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    assert torch.equal(C, C_ref)
    assert rel_l2(dW, dW_ref) < 1e-6 and rel_l2(db, db_ref) < 1e-6    # (atomic flush: order dependent)


def test_wgrad_store_mode_is_bit_reproducible_and_matches_the_atomic_form(dev, monkeypatch):
    """fgs_mlp_wgrad_ws (default, FGS_WGRAD_STORE=1): partial blocks by plain stores, added in slice order by a second launch --
    the same weight gradients as the atomic form to float32 summation order, and ONE bit pattern over repeated launches (the
    atomic form's sums depend on arrival order); accumulates into a non-zero dW like the atomic form."""
    from fgs_nerf_amd import fused_ops as fo
    M = 20011
    g = torch.Generator().manual_seed(M)
    n_in, ld_x = [106, 256, 307], [108, 256, 308]
    Xs = [torch.randn(M, ld, generator=g).to(dev) for ld in ld_x]
    dYs = [torch.randn(M, 256, generator=g).to(dev) for _ in range(3)]

    def run(store):
        monkeypatch.setattr(fo, "_WGRAD_STORE", store)
        dWs = [torch.full((256, ld), 0.5, device=dev) for ld in ld_x]
        dbs = [torch.zeros(256, device=dev) for _ in range(3)]
        fo.mlp_wgrad(M, [(dYs[i], Xs[i], dWs[i], dbs[i], 256, n_in[i]) for i in range(3)])
        torch.cuda.synchronize()
        return dWs, dbs

    a, _ = run(True)
    b, _ = run(True)
    c, _ = run(False)
    for i in range(3):
        assert torch.equal(a[i], b[i]), i
        ref = 0.5 + dYs[i].double().T @ Xs[i][:, :n_in[i]].double()
        assert rel_l2(a[i][:, :n_in[i]], ref) < 2e-6 and rel_l2(c[i][:, :n_in[i]], ref) < 2e-6
        assert float((a[i][:, n_in[i]:] - 0.5).abs().max()) == 0.0 if n_in[i] < ld_x[i] else True
