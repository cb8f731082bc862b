"""Forward MLP chain: one persistent launch (fgs_mlp_fwd_f32) vs one k_gemm launch per layer, over M."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgs_nerf_amd import fused_ops as fo
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


Ks = [108, 256, 256, 256, 308, 256, 256]
relu = [1, 1, 1, 0, 1, 1, 1]
for M in (16384, 32768, 49920, 65536, 105000):
    X0 = torch.randn(M, 108, device=dev)
    Z = torch.randn(M, 308, device=dev)
    Ws = [torch.randn(256, k, device=dev) * 0.05 for k in Ks]
    bs = [torch.randn(256, device=dev) * 0.1 for _ in Ks]
    outs = [torch.empty(M, 256, device=dev) for _ in Ks]
    outs[3] = Z                                             # the 4th layer writes Z[:, :256]

    def fused():
        fo.mlp_fwd(M, X0, 108, Z[:, 256:], 52, [(Ws[i], Ks[i], bs[i], relu[i], outs[i]) for i in range(7)])

    def layers():
        a = X0
        for i in range(7):
            fo.gemm(fo.GEMM_NT, a, Ws[i], outs[i], M, 256, Ks[i], bias=bs[i], relu=bool(relu[i]))
            a = outs[i]
    layers()
    ref = [o.clone() for o in outs]
    for o in outs:
        if o is not Z:
            o.zero_()
    Z[:, :256] = 0
    fused()
    same = all(torch.equal(o, r) for o, r in zip(outs, ref))
    tf, tl = timeit(fused), timeit(layers)
    fl = 2.0 * M * 256 * sum(Ks)
    print(f"M={M:6d}: one launch {tf*1e6:7.1f} us {fl/tf/1e12:6.1f} TF/s | per layer {tl*1e6:7.1f} us {fl/tl/1e12:6.1f} TF/s | "
          f"bit-identical {same}", flush=True)
