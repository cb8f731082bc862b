"""Survivor counts of bench.py's 8 resident batches (fresh model) and what form 2's tile quantisation costs on them."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from fgs_nerf_amd import synth
dev = torch.device("cuda:0")
model = synth.build_model(160, synth.FINE_MODEL, device=dev)
tot_c = tot_x = tot_h = 0.0
for b in range(8):
    ro, rd, vd, _ = bench.make_batch(b, 0, dev)
    with torch.no_grad():
        res = model(ro, rd, vd, global_step=1000, **synth.RENDER_KWARGS)
    M = int(res['weights'].shape[0])
    T = math.ceil(M / 32); x = T / 256; c = math.ceil(x)
    T16 = math.ceil(M / 16); h = math.ceil(T16 / 256) / 2
    print(f"batch {b}: survivors {M}, tiles/CU {x:.2f} -> {c} (32-sample tiles), {h:.1f} (16-sample tiles), form 1 rounds {M / 128 / 256:.2f} -> {math.ceil(M / 128 / 256)}")
    tot_c += c; tot_x += x; tot_h += h
print(f"mean tiles/CU {tot_x / 8:.2f}; cost with 32-sample tiles {tot_c / 8:.2f} (+{(tot_c / tot_x - 1) * 100:.1f} %), with 16-sample tiles {tot_h / 8:.2f} (+{(tot_h / tot_x - 1) * 100:.1f} %), form 1 {sum(1 for _ in range(8)) and 8.0:.2f}")
