"""Micro-bench of the three GEMM variants at the fine-stage MLP shapes (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgs_nerf_amd import fused_ops as fo

dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, ld = 64075, 256, 256
X = torch.randn(M, ld, device=dev)
W = torch.randn(N, ld, device=dev) * 0.1
dY = torch.randn(M, N, device=dev)
dW = torch.zeros(N, ld, device=dev)
Y = torch.empty(M, N, device=dev)
dX = torch.empty(M, ld, device=dev)
b = torch.randn(N, device=dev)


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


fl = 2.0 * M * N * ld
t_tn = timeit(lambda: fo.gemm(fo.GEMM_TN, dY, X, dW, N, ld, M))
t_nt = timeit(lambda: fo.gemm(fo.GEMM_NT, X, W, Y, M, N, ld, bias=b, relu=True))
t_nn = timeit(lambda: fo.gemm(fo.GEMM_NN, dY, W, dX, M, ld, N, mask=X))
print(f"TN_WGS={os.environ.get('FGS_TN_WGS')} TN {t_tn*1e6:.0f} us {fl/t_tn/1e12:.1f} TF/s | NT {t_nt*1e6:.0f} us "
      f"{fl/t_nt/1e12:.1f} TF/s | NN {t_nn*1e6:.0f} us {fl/t_nn/1e12:.1f} TF/s")
