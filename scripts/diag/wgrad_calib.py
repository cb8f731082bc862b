"""Per-k-step cost of fgs_mlp_wgrad blocks by width: one item at a time, all workgroups on it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fgs_nerf_amd import fused_ops as fo
dev = torch.device('cuda:0')
M = 262144
for n_in, ld in ((256, 256), (128, 128), (106, 108), (64, 64), (307, 308)):
    X = torch.randn(M, ld, device=dev); dY = torch.randn(M, 256, device=dev)
    dW = torch.zeros(256, ld, device=dev); db = torch.zeros(256, device=dev)
    items = [(dY, X, dW, db, 256, n_in)]
    for _ in range(3): fo.mlp_wgrad(M, items)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fo.mlp_wgrad(M, items)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    # k-steps per workgroup: M / 2 / 256 workgroups (307 columns: two blocks share the chip by the cost model)
    print(f"n_in={n_in:4d}: {us:8.1f} us for M={M}; cycles per k-step per workgroup at 2.4 GHz ~ {us * 1e-6 * 2.4e9 / (M / 2 / 256):7.0f}"
          f"  ({2.0 * M * 256 * n_in / (us * 1e-6) / 1e12:5.1f} TFLOP/s)", flush=True)
