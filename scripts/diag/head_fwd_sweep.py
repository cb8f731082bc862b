"""k_head_fwd alone: rows of 192 / 256 floats, grid cap from FGS_HEAD_FWD_BLOCKS."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fgs_nerf_amd._lib import call, ptr, stream

dev = torch.device('cuda:0')
for W, M in ((192, 354000), (256, 57000)):
    R = torch.randn(M, W, device=dev)
    V, b = torch.randn(3, W, device=dev) * 0.1, torch.zeros(3, device=dev)
    rgb = torch.empty(M, 3, device=dev)
    for _ in range(3):
        call("fgs_head_fwd", ptr(R), W, W, M, ptr(V), ptr(b), ptr(rgb), None, stream())
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call("fgs_head_fwd", ptr(R), W, W, M, ptr(V), ptr(b), ptr(rgb), None, stream())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    t = sorted(ts)[3]
    print(f"W={W} M={M}: {t:6.1f} us  {M * W * 4 / t / 1e6:5.2f} TB/s  cap={os.environ.get('FGS_HEAD_FWD_BLOCKS', '2048')}", flush=True)
