"""One REAL coarse-stage iteration (config/shiny_blender.py coarse_train: 8192 rays, ori_tv = True -- the autograd sdf TV and
smooth-gradient TV terms every iteration --, weight_rgbper, orientation loss) as TrainStepper.run_captured runs it: ms per iteration
at a given grid size.  bench.py --stage coarse times the same path WITHOUT the ori_tv terms (TV add-grad only)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fgs_nerf_amd import nerf_training as nt
from fgs_nerf_amd import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 114
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ORI = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
dev = torch.device('cuda:0')
cfg = dict(N_iters=15000, N_rand=8192, lrate_k0=0.1, lrate_sdf=0.1, lrate_refnet=1e-3, lrate_decay=20, ray_sampler='flatten',
           weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.2, weight_tv_density=0.01, weight_tv_k0=0.0,
           sigmoid_rgb_loss=0.1, weight_orientation=1e-4, tv_every=1, tv_from=0, tv_end=40000, voxel_inc=False, pg_scale=[],
           scale_ratio=3.0, reset_iter=[], ori_tv=ORI, tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.05), tv_updates={},
           decay_step_module={}, tv_dense_before=40000, cosine_lr=True,
           cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0), skip_zero_grad_fields=['density', 'k0', 'sdf'])
R = 65536
rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=5))
target = torch.rand(R, 3, generator=torch.Generator().manual_seed(3)).to(dev)
model = synth.build_model(G, synth.COARSE_MODEL, device=dev)
st = nt.TrainStepper(model, cfg, {}, synth.RENDER_KWARGS, target, *rays, stage='coarse', seed=1)
st.run_captured(1, 8)                      # capture + warm
torch.cuda.synchronize()
t0 = time.perf_counter()
losses, overflow = st.run_captured(9, N)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"grid {G}^3, 8192 rays, ori_tv={ORI}: {dt / N * 1e3:.3f} ms per iteration over {N} iterations (incl. one capture: "
      f"see the second figure), overflow={overflow}, last loss {float(losses[-1]):.5f}", flush=True)
t0 = time.perf_counter()
losses, overflow = st.run_captured(9 + N, 4 * N)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"  {4 * N} more iterations: {dt / (4 * N) * 1e3:.3f} ms per iteration (capture amortised over 4 x as many)", flush=True)
