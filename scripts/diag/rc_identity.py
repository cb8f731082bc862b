"""Decode which weight chunk each output tile sees at each reduction step of fgs_mlp_rc_chain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fgs_nerf_amd import fused_ops as fo
dev = torch.device('cuda:0')
M = 64
K = 256
W = torch.arange(K, device=dev, dtype=torch.float32)[None, :].repeat(256, 1).contiguous()   # W[f][k] = k
for s in range(8):
    X0 = torch.zeros(M, K, device=dev)
    X0[:, 32 * s:32 * s + 32] = 1.0
    out = torch.full((M, 256), -7.0, device=dev)
    fo.rc_chain(False, M, X0, K, [dict(W=W, out=out, n_store=256)])
    torch.cuda.synchronize()
    o = out.cpu()
    # out[f] = sum_k' (32 c + k') = 1024 c + 496 if the tile saw chunk c at step s
    dec = ((o - 496) / 1024)
    print("step", s, "expected chunk", s, "| row 0 tiles:", [round(float(dec[0, 32 * t]), 2) for t in range(8)],
          "| row 40 tiles:", [round(float(dec[40, 32 * t + 5]), 2) for t in range(8)])
