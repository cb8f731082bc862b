"""GPU tests of the feature-split MLP chains (csrc/mlp_rc2.hip, fgs_mlp_rc2_chain) against float64 torch -- the same cases as
tests/test_mlp_rc_gpu.py for the register-resident form: forward (bias, ReLU, appended columns, saved activations, ReLU sign
bits in this form's own layout), backward data gradients (transposed images, masks from the forward's bits) INCLUDING the two
narrow products as side layers (the reflection-encoding columns of dZ, the compact dX0), ragged M (every slab size 1..4 and the
uneven deal of sample tiles to workgroups), padding columns holding NaN, the device-side row count."""
import pytest
import torch

from conftest import rel_l2
from test_mlp_rc_gpu import _fine_setup, _reference_forward

pytestmark = pytest.mark.gpu


def _forward2(M, X0, Z, Ws, bs, relu, dev, cap=None, rows_dev=None):
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
    fo.rc_chain(False, cap, X0, 108, layers, rows_dev=rows_dev, form=2)
    return outs, bits


def _unpack_bits2(bits, M):
    """[tiles][4 waves][64 lanes] words -> bool [M][256]: element e = 16 f + r of a word sits at bit 31 - e and is feature
    64 wave + 32 f + 8 (r >> 2) + 4 h + (r & 3) of sample 32 tile + j (lane = 32 h + j)."""
    T = (M + 31) // 32
    w = bits[:T * 256].view(T, 4, 64).cpu().to(torch.int64) & 0xffffffff
    out = torch.zeros(T * 32, 256, dtype=torch.bool)
    for wave in range(4):
        for f in range(2):
            for r in range(16):
                b = ((w[:, wave, :] >> (31 - (16 * f + r))) & 1).bool()        # [T][64]
                for h in range(2):
                    feat = 64 * wave + 32 * f + 8 * (r >> 2) + 4 * h + (r & 3)
                    out[:, feat] = b[:, 32 * h:32 * h + 32].reshape(-1)
    return out[:M]


@pytest.mark.parametrize("M", [1, 31, 33, 128, 129, 1000, 8192 + 17, 50001, 65536])
def test_rc2_forward_matches_fp64(dev, M):
    X0, Z, Ws, bs, relu = _fine_setup(M, dev, seed=M)
    outs, bits = _forward2(M, X0, Z, Ws, bs, relu, dev)
    ref = _reference_forward(X0, Z, Ws, bs, relu)
    for i in range(7):
        got = outs[i][:, :256]
        assert bool(torch.isfinite(got).all()), i
        assert rel_l2(got, ref[i]) < 2e-6, (i, rel_l2(got, ref[i]))
    assert bool(torch.isfinite(Z[:, 256:307]).all()) and bool(torch.isnan(Z[:, 307]).all())   # appended columns untouched
    for i in (0, 1, 2, 4, 5, 6):
        assert torch.equal(_unpack_bits2(bits[i], M), (outs[i][:, :256] > 0).cpu()), i


@pytest.mark.parametrize("M", [1, 100, 4097, 40000])
def test_rc2_backward_with_side_layers_matches_fp64(dev, M):
    from fgs_nerf_amd import fused_ops as fo
    X0, Z, Ws, bs, relu = _fine_setup(M, dev, seed=M + 3)
    outs, bits = _forward2(M, X0, Z, Ws, bs, relu, dev)
    g = torch.Generator().manual_seed(M)
    dY = torch.randn(M, 256, generator=g).to(dev)
    d5, d4, d2, d1, d0 = (torch.full((M, 256), float('nan'), device=dev) for _ in range(5))
    dZ = torch.full((M, 308), float('nan'), device=dev)
    # the compact first-layer weight: 52 of the 106 columns (as fused_fine builds W0c), and the K-padded refnet layer 0
    W0c = torch.cat([Ws[0][:, :12], Ws[0][:, 66:]], 1).contiguous()
    dX0 = torch.full((M, 52), float('nan'), device=dev)
    V0p = torch.nn.functional.pad(Ws[4], (0, 1)).contiguous()          # [256, 308]
    layers = [dict(W=Ws[6], mask_bits=bits[5], out=d5, n_store=256), dict(W=Ws[5], mask_bits=bits[4], out=d4, n_store=256),
              # side: the reflection-encoding columns of dZ, from the carried dY_ref0 (BEFORE the main layer replaces it)
              dict(W=V0p[:, 256:], out=dZ[:, 256:], n_store=52, side=True),
              dict(W=Ws[4][:, :256], out=dZ, n_store=256), dict(W=Ws[3], mask_bits=bits[2], out=d2, n_store=256),
              dict(W=Ws[2], mask_bits=bits[1], out=d1, n_store=256), dict(W=Ws[1], mask_bits=bits[0], out=d0, n_store=256),
              dict(W=W0c, out=dX0, n_store=52, side=True)]
    fo.rc_chain(True, M, dY, 256, layers, form=2)
    acts = [o[:, :256].double() for o in outs]
    gcur = dY.double()
    refs = {}
    for i in range(6, 0, -1):
        if i == 4:
            refs['enc'] = gcur @ Ws[4].double()[:, 256:]
        gcur = gcur @ Ws[i].double()[:, :256]
        if i in (6, 5, 3, 2, 1):
            gcur = gcur * (acts[i - 1] > 0)
        refs[i] = gcur
    refs['dx0'] = gcur @ W0c.double()
    for name, got, ref in (("d5", d5, refs[6]), ("d4", d4, refs[5]), ("dZ main", dZ[:, :256], refs[4]), ("d2", d2, refs[3]),
                           ("d1", d1, refs[2]), ("d0", d0, refs[1]), ("dZ enc", dZ[:, 256:307], refs['enc']),
                           ("dX0 compact", dX0, refs['dx0'])):
        assert bool(torch.isfinite(got).all()), name
        assert rel_l2(got, ref) < 2e-6, (name, rel_l2(got, ref))
    assert float(dZ[:, 307].abs().max()) == 0.0          # the padding column of dZ: a zero weight column, written as 0


def test_rc2_device_row_count(dev):
    cap, M = 3000, 1777
    X0, Z, Ws, bs, relu = _fine_setup(cap, dev, seed=5)
    outs_a, _ = _forward2(M, X0[:M].contiguous(), Z[:M].clone(), Ws, bs, relu, dev)
    count = torch.tensor([M], dtype=torch.int64, device=dev)
    Zb = Z.clone()
    outs_b, _ = _forward2(M, X0, Zb, Ws, bs, relu, dev, cap=cap, rows_dev=count.data_ptr())
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a[:M, :256], b[:M, :256])
        assert bool(torch.isnan(b[M:, :256]).all())


def test_rc2_is_deterministic_and_close_to_the_first_form(dev):
    from test_mlp_rc_gpu import _forward
    M = 20000
    X0, Z, Ws, bs, relu = _fine_setup(M, dev, seed=11)
    a, _ = _forward2(M, X0, Z.clone(), Ws, bs, relu, dev)
    b, _ = _forward2(M, X0, Z.clone(), Ws, bs, relu, dev)
    c, _ = _forward(M, X0, Z.clone(), Ws, bs, relu, dev)
    for x, y, z in zip(a, b, c):
        assert torch.equal(x[:, :256], y[:, :256])
        assert rel_l2(x[:, :256], z[:, :256]) < 1e-6


@pytest.mark.parametrize("width,n_in,M,compact", [(192, 90, 777, 48), (192, 90, 40000, 48), (128, 72, 513, 36), (128, 72, 30000, 36)])
def test_rc2_coarse_widths(dev, width, n_in, M, compact):
    """The coarse (90 -> 192 -> 192) and geometry_searching (72 -> 128 -> 128) refnet trunks: 6 feature tiles (one per wave plus
    tiles 4, 5 dealt by (tile, sample tile)) and 4 (one per wave); forward, and backward with the compact dX0 as a side layer."""
    from fgs_nerf_amd import fused_ops as fo
    g = torch.Generator().manual_seed(width + M)
    ld = (n_in + 3) // 4 * 4
    X0 = torch.randn(M, ld, generator=g).to(dev)
    X0[:, n_in:] = float('nan')
    W0, W1 = (torch.randn(width, n_in, generator=g) * 0.1).to(dev), (torch.randn(width, width, generator=g) * 0.08).to(dev)
    b0, b1 = (torch.randn(width, generator=g) * 0.1).to(dev), (torch.randn(width, generator=g) * 0.1).to(dev)
    o0, o1 = (torch.full((M, width), float('nan'), device=dev) for _ in range(2))
    m0, m1 = fo.rc_mask_bits(M, dev), fo.rc_mask_bits(M, dev)
    fo.rc_chain(False, M, X0, ld, [dict(W=W0, bias=b0, relu=1, mask_bits=m0, out=o0, n_store=width),
                                   dict(W=W1, bias=b1, relu=1, mask_bits=m1, out=o1, n_store=width)], form=2)
    r0 = torch.relu(X0[:, :n_in].double() @ W0.double().T + b0.double())
    r1 = torch.relu(r0 @ W1.double().T + b1.double())
    assert bool(torch.isfinite(o0).all()) and bool(torch.isfinite(o1).all())
    assert rel_l2(o0, r0) < 2e-6 and rel_l2(o1, r1) < 2e-6
    dY = torch.randn(M, width, generator=g).to(dev)
    d0 = torch.full((M, width), float('nan'), device=dev)
    W0c = W0[:, :compact].contiguous()                        # (any column subset: the product only sees a [width, compact] matrix)
    dX0 = torch.full((M, compact), float('nan'), device=dev)
    fo.rc_chain(True, M, dY, width, [dict(W=W1, mask_bits=m0, out=d0, n_store=width),
                                     dict(W=W0c, out=dX0, n_store=compact, side=True)], form=2)
    g0 = (dY.double() @ W1.double()) * (r0 > 0)
    assert bool(torch.isfinite(d0).all()) and bool(torch.isfinite(dX0).all())
    assert rel_l2(d0, g0) < 2e-6
    assert rel_l2(dX0, g0 @ W0c.double()) < 2e-6
