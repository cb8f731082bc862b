"""fgs_mlp_wgrad: time vs M for one 256x256 item (all workgroups on it) and for the fine-stage set: intercept = fixed cost
(launch, prologue, atomics flush), slope = cycles per k-step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fgs_nerf_amd import fused_ops as fo
dev = torch.device('cuda:0')
def timeit(fn, n=8):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, n_in, ld_x in (("one 256", [256], [256]), ("fine set", [106, 256, 256, 256, 307, 256, 256], [108, 256, 256, 256, 308, 256, 256])):
    for M in (4096, 16384, 32768, 58430, 65536, 131072, 262144):
        Xs = [torch.randn(M, ld, device=dev) for ld in ld_x]
        dYs = [torch.randn(M, 256, device=dev) for _ in ld_x]
        dWs = [torch.zeros(256, ld, device=dev) for ld in ld_x]
        dbs = [torch.zeros(256, device=dev) for _ in ld_x]
        items = [(dYs[i], Xs[i], dWs[i], dbs[i], 256, n_in[i]) for i in range(len(ld_x))]
        us = timeit(lambda: fo.mlp_wgrad(M, items))
        fl = 2.0 * M * 256 * sum(n_in)
        print(f"{name:9s} M={M:7d}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s", flush=True)
