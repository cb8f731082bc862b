"""Phase times of fgs_mlp_wgrad workgroups (fgs_mlp_wgrad_debug_stamps) on the fine-stage set."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fgs_nerf_amd import fused_ops as fo
from fgs_nerf_amd._lib import call, ptr
dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 58430
n_in = [106, 256, 256, 256, 307, 256, 256]; ld_x = [108, 256, 256, 256, 308, 256, 256]
Xs = [torch.randn(M, ld, device=dev) for ld in ld_x]
dYs = [torch.randn(M, 256, device=dev) for _ in ld_x]
dWs = [torch.zeros(256, ld, device=dev) for ld in ld_x]
dbs = [torch.zeros(256, device=dev) for _ in ld_x]
items = [(dYs[i], Xs[i], dWs[i], dbs[i], 256, n_in[i]) for i in range(7)]
import time
t0 = time.time()
while time.time() - t0 < 2.5:            # clock management settles on sustained load (MI355X_MICROARCH.md, DVFS give-back 6)
    for _ in range(50): fo.mlp_wgrad(M, items)
    torch.cuda.synchronize()
st = torch.zeros(2048, dtype=torch.int64, device=dev)
call("fgs_mlp_wgrad_debug_stamps", ptr(st))
fo.mlp_wgrad(M, items)
torch.cuda.synchronize()
call("fgs_mlp_wgrad_debug_stamps", None)
s = st.cpu().numpy().reshape(256, 8).astype(np.float64)
s = s[s[:, 0] > 0]
clk = (s[:, 4] - s[:, 0]) / (s[:, 5] - s[:, 1]) * 100e6
print(f"workgroups {len(s)}; in-kernel clock {clk.mean() / 1e9:.3f} GHz (min {clk.min() / 1e9:.3f})")
t0 = s[:, 1].min()
print(f"wall: first start -> last end {(s[:, 5].max() - t0) / 100:.1f} us; start skew {(s[:, 1].max() - t0) / 100:.1f} us")
for blk in sorted(set(s[:, 7].astype(int))):
    r = s[s[:, 7] == blk]
    pro, loop, fl = r[:, 2] - r[:, 0], r[:, 3] - r[:, 2], r[:, 4] - r[:, 3]
    ch = r[:, 6]
    print(f"block {blk}: {len(r):3d} wgs, chunks {ch.mean():6.1f}; prologue {pro.mean():7.0f} cyc, loop {loop.mean():9.0f} cyc = "
          f"{(loop / ch).mean():7.0f} per chunk (max wg {loop.max():9.0f}), flush issue {fl.mean():7.0f} cyc; "
          f"end {((r[:, 5] - t0) / 100).mean():6.1f} us (max {((r[:, 5] - t0) / 100).max():6.1f})")
