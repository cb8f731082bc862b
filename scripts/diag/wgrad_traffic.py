"""HBM fetch bytes of k_mlp_wgrad per item set (run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`): which operand shapes
over-fetch.  Each configuration is launched 3 times in the order printed; scripts/diag/wgrad_traffic_read.py reads the counter CSV."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fgs_nerf_amd import fused_ops as fo

dev = torch.device('cuda:0')
M = 57000
g = torch.Generator().manual_seed(0)
X256 = torch.randn(M, 256, generator=g).to(dev)
X108 = torch.randn(M, 108, generator=g).to(dev)
Z308 = torch.randn(M, 308, generator=g).to(dev)
Z320 = torch.randn(M, 320, generator=g).to(dev)
dY = torch.randn(M, 256, generator=g).to(dev)
configs = [
    ("256x256 (X ld 256)", [(dY, X256, 256, 256)]),
    ("256x106 (X ld 108)", [(dY, X108, 256, 106)]),
    ("256x307 (X ld 308)", [(dY, Z308, 256, 307)]),
    ("256x307 (X ld 320)", [(dY, Z320, 256, 307)]),
    ("dY strided ld 308 x X256", [(Z308[:, :256], X256, 256, 256)]),
]
for name, items in configs:
    full = []
    for (d, X, n_out, n_in) in items:
        full.append((d, X, torch.zeros(n_out, X.shape[1], device=dev), torch.zeros(n_out, device=dev), n_out, n_in))
    for _ in range(3):
        fo.mlp_wgrad(M, full)
    torch.cuda.synchronize()
    alg = sum(M * 4 * (n_out + n_in) for (_, _, n_out, n_in) in items)
    print(f"{name}: algorithmic {alg / 1e6:.1f} MB", flush=True)
