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
    bwd2 = [dict(W=W1, mask_bits=m0, out=dY0, n_store=W), dict(W=V0c, out=dX0, n_store=48, side=True)]
    t_f2 = timed(lambda: fo.rc_chain(False, M, X0, ld0, fwd, form=2))
    t_b2 = timed(lambda: fo.rc_chain(True, M, dY1, W, bwd2, form=2))
    print(f"W={W} K0={K0} M={M}: form 2: fwd chain {t_f2:6.1f} us ({2.0 * M * W * (K0 + W) / t_f2 / 1e6:5.1f} TF)  bwd chain + dX0 side layer "
          f"{t_b2:6.1f} us ({2.0 * M * W * (W + 48) / t_b2 / 1e6:5.1f} TF)", flush=True)
    if os.environ.get("RC2_PHASES"):
        import numpy as np
        for what, backward, layers, in0, cols in (("fwd", False, fwd, X0, ld0), ("bwd", True, bwd2, dY1, W)):
            buf = torch.zeros(1, fo.STAMP_LAUNCHES, fo.STAMP_WORDS, dtype=torch.int64, device=dev)
            for _ in range(3):
                fo.STAMPS.update(buf=buf, counter=None)
                fo.stamps_begin_step()
                fo.rc_chain(backward, M, in0, cols, layers, form=2)
                torch.cuda.synchronize()
            fo.STAMPS.update(buf=None, counter=None)
            wds = buf[0, 0].cpu().numpy().astype(np.uint64).reshape(-1, 8)
            wds = wds[wds[:, 1] > 0]
            cyc = (wds[:, 2] - wds[:, 0]).astype(np.float64)
            wall = (wds[:, 3] - wds[:, 1]).astype(np.float64) / 100.0
            print(f"    {what} form 2: {len(wds)} workgroups, in-kernel {np.median(wall):.1f} us (max {wall.max():.1f}), clock "
                  f"{np.median(cyc / wall) / 1e3:.2f} GHz; wave 0 cycles: total {np.median(cyc):.0f} = slab input {np.median(wds[:, 7]):.0f} + "
                  f"init {np.median(wds[:, 4]):.0f} + reductions {np.median(wds[:, 5]):.0f} + epilogues/barriers {np.median(wds[:, 6]):.0f}", flush=True)
    t_f = timed(lambda: fo.rc_chain(False, M, X0, ld0, fwd))
    t_b = timed(lambda: fo.rc_chain(True, M, dY1, W, bwd))
    t_g = timed(lambda: fo.gemm(fo.GEMM_NN, dY0, V0c, dX0, M, 48, W))
    t_w = timed(lambda: fo.mlp_wgrad(M, items))
    fl_f = 2.0 * M * W * (K0 + W)
    print(f"W={W} K0={K0} M={M}: fwd chain {t_f:6.1f} us ({fl_f / t_f / 1e6:5.1f} TF)  bwd chain {t_b:6.1f} us ({2.0 * M * W * W / t_b / 1e6:5.1f} TF)  "
          f"dX0 {t_g:5.1f} us  wgrad {t_w:6.1f} us ({fl_f / t_w / 1e6:5.1f} TF)", flush=True)
