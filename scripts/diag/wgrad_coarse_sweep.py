"""k_mlp_wgrad alone at the coarse stage's two layers (192 x 90, 192 x 192) over a sweep of M: fixed vs per-sample cost."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fgs_nerf_amd import fused_ops as fo

dev = torch.device('cuda:0')
W, K0 = 192, 90
ld0 = 92
for M in (4096, 16384, 32768, 65536, 98304, 196608, 354000):
    X0 = torch.randn(M, ld0, device=dev)
    a0 = torch.randn(M, W, device=dev)
    dY1, dY0 = torch.randn(M, W, device=dev), torch.randn(M, W, device=dev)
    g0, g1, gb0 = torch.zeros(W, ld0, device=dev), torch.zeros(W, W, device=dev), torch.zeros(W, device=dev)
    for name, items in (("both", [(dY0, X0, g0, gb0, W, K0), (dY1, a0, g1, None, W, W)]), ("192x192", [(dY1, a0, g1, None, W, W)]),
                        ("192x90", [(dY0, X0, g0, gb0, W, K0)])):
        for _ in range(3):
            fo.mlp_wgrad(M, items)
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fo.mlp_wgrad(M, items)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        t = sorted(ts)[3]
        fl = 2.0 * M * W * sum(it[5] for it in items)
        print(f"M={M:7d} {name:8s} {t:7.1f} us  {fl / t / 1e6:6.1f} TF", flush=True)
