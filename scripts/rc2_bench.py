"""Micro-benchmark: the two forms of the one-launch MLP chains (fgs_mlp_rc_chain / fgs_mlp_rc2_chain) alone on the chip, fine-stage
shapes, forward and backward (form 1: chain + the two narrow k_gemm products; form 2: the narrow products as side layers).

    python scripts/rc2_bench.py [M ...]        (default: 50000 56700 57600 64075 65536)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fgs_nerf_amd import fused_ops as fo       # noqa: E402


def timed(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    from test_mlp_rc_gpu import _fine_setup
    dev = torch.device("cuda:0")
    Ms = [int(a) for a in sys.argv[1:] if not a.startswith('-')] or [50000, 56700, 57600, 64075, 65536]
    spin = torch.randn(8192, 8192, device=dev)
    for _ in range(30):           # clocks up
        spin @ spin
    for M in Ms:
        X0, Z, Ws, bs, relu = _fine_setup(M, dev, seed=1)
        X0[:, 106:] = 0
        Z[torch.isnan(Z)] = 0
        outs = [torch.empty(M, 256, device=dev) for _ in Ws]
        outs[3] = Z
        bits = [fo.rc_mask_bits(M, dev) if relu[i] else None for i in range(7)]
        fwd = []
        for i in range(7):
            L = dict(W=Ws[i], bias=bs[i], relu=relu[i], mask_bits=bits[i], out=outs[i], n_store=256)
            if i == 4:
                L.update(ext=Z[:, 256:], ext_cols=52)
            fwd.append(L)
        flop_f = 2.0 * M * 256 * (106 + 256 * 3 + 307 + 256 * 2)
        dY = torch.randn(M, 256, device=dev)
        d = [torch.empty(M, 256, device=dev) for _ in range(5)]
        dZ = torch.empty(M, 308, device=dev)
        dX0 = torch.empty(M, 52, device=dev)
        W0c = torch.cat([Ws[0][:, :12], Ws[0][:, 66:]], 1).contiguous()
        V0p = torch.nn.functional.pad(Ws[4], (0, 1)).contiguous()
        main_l = [dict(W=Ws[6], mask_bits=bits[5], out=d[0], n_store=256), dict(W=Ws[5], mask_bits=bits[4], out=d[1], n_store=256),
                  dict(W=Ws[4][:, :256], out=dZ, n_store=256), dict(W=Ws[3], mask_bits=bits[2], out=d[2], n_store=256),
                  dict(W=Ws[2], mask_bits=bits[1], out=d[3], n_store=256), dict(W=Ws[1], mask_bits=bits[0], out=d[4], n_store=256)]
        side_enc = dict(W=V0p[:, 256:], out=dZ[:, 256:], n_store=52, side=True)
        side_x0 = dict(W=W0c, out=dX0, n_store=52, side=True)
        bwd2 = main_l[:2] + [side_enc] + main_l[2:] + [side_x0]
        flop_b = 2.0 * M * 256 * 256 * 6
        flop_n = 2.0 * M * 256 * 52 * 2

        def bwd1():
            fo.rc_chain(True, M, dY, 256, main_l, form=1)
            fo.gemm(fo.GEMM_NN, d[1], V0p[:, 256:], dZ[:, 256:], M, 52, 256)
            fo.gemm(fo.GEMM_NN, d[4], W0c, dX0, M, 52, 256)

        res = {}
        for form in (1, 2):
            res[("fwd", form)] = timed(lambda: fo.rc_chain(False, M, X0, 108, fwd, form=form))
        res[("bwd", 1)] = timed(bwd1)
        res[("bwd", 2)] = timed(lambda: fo.rc_chain(True, M, dY, 256, bwd2, form=2))
        res[("bwd-main-only", 2)] = timed(lambda: fo.rc_chain(True, M, dY, 256, main_l, form=2))
        print(f"M = {M}  ({M / 32 / 256:.2f} tiles per CU; form 1: {M / 128 / 256:.2f} rounds)")
        for (what, form), (med, best) in res.items():
            fl = flop_f if what == "fwd" else flop_b + (flop_n if what == "bwd" else 0)
            print(f"    {what:<14} form {form}: median {med:7.1f} us  best {best:7.1f} us   {fl / med / 1e6:6.1f} TFLOP/s (incl. pack launch)")
    if "--phases" in sys.argv or os.environ.get("RC2_PHASES"):
        # (make -C fgs_nerf_amd/csrc rc2-stamps): wave 0 of every workgroup: shader cycles per phase, shader / wall clock
        import numpy as np
        M = Ms[-1]
        for what, backward, layers, in0, cols in (("fwd", False, fwd, X0, 108), ("bwd", True, bwd2, dY, 256)):
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
            wall = (wds[:, 3] - wds[:, 1]).astype(np.float64) / 100.0      # us
            print(f"    {what} form 2, M = {M}: {len(wds)} workgroups, in-kernel {np.median(wall):.1f} us (max {wall.max():.1f}), "
                  f"clock {np.median(cyc / wall) / 1e3:.2f} GHz; cycles of wave 0 (median): total {np.median(cyc):.0f} = slab input "
                  f"{np.median(wds[:, 7]):.0f} + accumulator init {np.median(wds[:, 4]):.0f} + reductions {np.median(wds[:, 5]):.0f} + "
                  f"epilogues and barriers {np.median(wds[:, 6]):.0f} + rest")
    print("done")


if __name__ == "__main__":
    main()
