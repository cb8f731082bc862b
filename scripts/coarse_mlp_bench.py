"""Micro-benchmark: the coarse stages' MLP launches alone (refnet 90 -> 192 -> 192 -> 3 of the shipped coarse config at the bench's
survivor count; geometry_searching: 128 wide): forward chain, backward chain, the dX0 product, the weight-gradient launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgs_nerf_amd import fused_ops as fo

dev = torch.device('cuda:0')


def timed(fn, reps=5, rounds=5):
    for _ in range(3):
        fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(out)[len(out) // 2]


for W, K0, M in ((192, 90, 98304), (192, 90, 354000), (128, 66, 98304)):
    torch.manual_seed(0)
    ld0 = (K0 + 3) // 4 * 4
    X0 = torch.randn(M, ld0, device=dev)
    W0, W1 = torch.randn(W, K0, device=dev) * 0.1, torch.randn(W, W, device=dev) * 0.07
    b0, b1 = torch.zeros(W, device=dev), torch.zeros(W, device=dev)
    a0, a1 = torch.empty(M, W, device=dev), torch.empty(M, W, device=dev)
    m0, m1 = fo.rc_mask_bits(M, dev), fo.rc_mask_bits(M, dev)
    fwd = [dict(W=W0, bias=b0, relu=1, mask_bits=m0, out=a0, n_store=W), dict(W=W1, bias=b1, relu=1, mask_bits=m1, out=a1, n_store=W)]
    dY1, dY0 = torch.randn(M, W, device=dev), torch.empty(M, W, device=dev)
    bwd = [dict(W=W1, mask_bits=m0, out=dY0, n_store=W)]
    V0c = torch.randn(W, 48, device=dev)
    dX0 = torch.empty(M, 48, device=dev)
    g0, g1 = torch.zeros(W, ld0, device=dev), torch.zeros(W, W, device=dev)
    gb0 = torch.zeros(W, device=dev)
    items = [(dY0, X0, g0, gb0, W, K0), (dY1, a0, g1, None, W, W)]
    t_f = timed(lambda: fo.rc_chain(False, M, X0, ld0, fwd))
    t_b = timed(lambda: fo.rc_chain(True, M, dY1, W, bwd))
    t_g = timed(lambda: fo.gemm(fo.GEMM_NN, dY0, V0c, dX0, M, 48, W))
    t_w = timed(lambda: fo.mlp_wgrad(M, items))
    fl_f = 2.0 * M * W * (K0 + W)
    print(f"W={W} K0={K0} M={M}: fwd chain {t_f:6.1f} us ({fl_f / t_f / 1e6:5.1f} TF)  bwd chain {t_b:6.1f} us ({2.0 * M * W * W / t_b / 1e6:5.1f} TF)  "
          f"dX0 {t_g:5.1f} us  wgrad {t_w:6.1f} us ({fl_f / t_w / 1e6:5.1f} TF)", flush=True)
