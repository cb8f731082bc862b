"""Fixed overhead vs per-chunk cost of the NT GEMM: time over K at M = one / two tiles per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgs_nerf_amd import fused_ops as fo
dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for M in (16384, 32768, 49920, 65536, 105000):
    for K in (128, 192, 256, 1024):
        X = torch.randn(M, K, device=dev)
        W = torch.randn(256, K, device=dev) * 0.1
        b = torch.randn(256, device=dev)
        Y = torch.empty(M, 256, device=dev)
        t = timeit(lambda: fo.gemm(fo.GEMM_NT, X, W, Y, M, 256, K, bias=b, relu=True))
        print(f"M={M} K={K:5d}: {t*1e6:7.1f} us  {2.0*M*256*K/t/1e12:6.1f} TF/s  ({t*1e6/(K/32):6.2f} us/chunk)", flush=True)
