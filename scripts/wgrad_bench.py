"""Micro-benchmark: all weight gradients of the fine-stage MLPs, one fgs_mlp_wgrad launch vs seven split-K fgs_gemm_f32 (TN)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgs_nerf_amd import fused_ops as fo
dev = torch.device('cuda:0')
for M in (49920, 58430, 65536):
    torch.manual_seed(0)
    n_in = [106, 256, 256, 256, 307, 256, 256]
    ld_x = [108, 256, 256, 256, 308, 256, 256]
    Xs = [torch.randn(M, ld, device=dev) for ld in ld_x]
    dYs = [torch.randn(M, 256, device=dev) for _ in range(7)]
    dWs = [torch.zeros(256, ld, device=dev) for ld in ld_x]
    dbs = [torch.zeros(256, device=dev) for _ in range(7)]
    items = [(dYs[i], Xs[i], dWs[i], dbs[i], 256, n_in[i]) for i in range(7)]
    flop = 2.0 * M * 256 * sum(n_in)
    def run_new(): fo.mlp_wgrad(M, items)
    def run_old():
        for i in range(7): fo.gemm(fo.GEMM_TN, dYs[i], Xs[i], dWs[i], 256, ld_x[i], M)
    res = {}
    for fn in (run_new, run_old):
        for _ in range(3): fn()
    for rnd in range(5):
        for name, fn in (("wgrad", run_new), ("7xTN", run_old)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    for name, v in res.items():
        v = sorted(v)
        print(f"M={M} {name:6s} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f} us  -> {flop / (v[len(v)//2] * 1e-6) / 1e12:6.1f} TFLOP/s", flush=True)
