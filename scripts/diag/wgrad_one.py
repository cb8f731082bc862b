import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fgs_nerf_amd import fused_ops as fo
dev = torch.device('cuda:0')
M = 262144
X = torch.randn(M, 256, device=dev); dY = torch.randn(M, 256, device=dev)
dW = torch.zeros(256, 256, device=dev); db = torch.zeros(256, device=dev)
for _ in range(5): fo.mlp_wgrad(M, [(dY, X, dW, db, 256, 256)])
torch.cuda.synchronize()
