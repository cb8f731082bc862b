"""GPU micro-check + micro-bench of fgs_gemm_f32 (run on the GPU box): correctness vs float64 torch, TFLOP/s per shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgs_nerf_amd import fused_ops as fo

dev = torch.device("cuda:0")
torch.manual_seed(0)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


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


for M in (1000, 64075):
    for (K, N, ldx) in ((108, 256, 112), (256, 256, 256), (308, 256, 320)):
        X = torch.randn(M, ldx, device=dev)
        X[:, K:] = 0
        W = torch.randn(N, ldx, device=dev) * 0.1
        W[:, K:] = 0
        b = torch.randn(N, device=dev)
        Y = torch.empty(M, N, device=dev)
        cs = torch.zeros(N, device=dev)
        fo.gemm(fo.GEMM_NT, X, W, Y, M, N, ldx, bias=b, relu=True, colsum=cs)
        ref = torch.relu(X.double() @ W.double().T + b.double())
        e_nt, e_cs = rel(Y, ref), rel(cs, ref.sum(0))
        # NN: dX = dY @ W (masked by X>0 of a ReLU layer)
        dY = torch.randn(M, N, device=dev)
        act = torch.randn(M, ldx, device=dev)
        dX = torch.empty(M, ldx, device=dev)
        fo.gemm(fo.GEMM_NN, dY, W, dX, M, ldx, N, mask=act)
        refd = (dY.double() @ W.double()) * (act > 0)
        e_nn = rel(dX, refd)
        # TN: dW += dY^T @ X
        dW = torch.zeros(N, ldx, device=dev)
        fo.gemm(fo.GEMM_TN, dY, X, dW, N, ldx, M)
        e_tn = rel(dW, dY.double().T @ X.double())
        t_nt = timeit(lambda: fo.gemm(fo.GEMM_NT, X, W, Y, M, N, ldx, bias=b, relu=True))
        t_nn = timeit(lambda: fo.gemm(fo.GEMM_NN, dY, W, dX, M, ldx, N, mask=act))
        t_tn = timeit(lambda: fo.gemm(fo.GEMM_TN, dY, X, dW, N, ldx, M))
        t_ref = timeit(lambda: torch.relu(torch.addmm(b, X, W.T)))
        fl = 2.0 * M * N * ldx
        print(f"M={M} K={K}(ld {ldx}) N={N}: err nt {e_nt:.1e} colsum {e_cs:.1e} nn {e_nn:.1e} tn {e_tn:.1e} | "
              f"TF/s nt {fl/t_nt/1e12:.1f} nn {fl/t_nn/1e12:.1f} tn {fl/t_tn/1e12:.1f} | torch addmm+relu {fl/t_ref/1e12:.1f}", flush=True)
