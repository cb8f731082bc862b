"""A REAL fine-stage iteration (config/shiny_blender.py fine_train: 8192 rays, the TV schedule -- sdf TV add-grad and the autograd
smooth-gradient TV term -- every third iteration) through TrainStepper.run_captured (two graphs, chosen per iteration; the batch
gather included): ms per iteration at a given grid size.  bench.py times the plain iteration with the TV add-grad in every step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fgs_nerf_amd import nerf_training as nt
from fgs_nerf_amd import synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 160
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device('cuda:0')
cfg = dict(N_iters=20000, N_rand=8192, lrate_k0=0.1, lrate_sdf=0.005, lrate_rgbnet=1e-3, lrate_refnet=1e-3, lrate_decay=20,
           ray_sampler='flatten', weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.0, weight_tv_density=0.01,
           weight_tv_k0=0.0, sigmoid_rgb_loss=0.02, weight_orientation=1e-4, tv_every=3, tv_from=0, tv_end=30000,
           voxel_inc=False, pg_scale=[], reset_iter=[], tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.05), tv_dense_before=20000,
           cosine_lr=True, cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0), decay_step_module={},
           skip_zero_grad_fields=['density', 'k0', 'k1'])
R = 65536
rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=5))
target = torch.rand(R, 3, generator=torch.Generator().manual_seed(3)).to(dev)
model = synth.build_model(G, synth.FINE_MODEL, device=dev)
st = nt.TrainStepper(model, cfg, {}, synth.RENDER_KWARGS, target, *rays, stage='fine', seed=1)
st.run_captured(1000, 9)
torch.cuda.synchronize()
t0 = time.perf_counter()
losses, overflow = st.run_captured(1009, N)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"grid {G}^3, 8192 rays, TV schedule every 3rd iteration: {dt / N * 1e3:.3f} ms per iteration over {N} iterations "
      f"(one capture of two graphs included), overflow={overflow}, last loss {float(losses[-1]):.5f}", flush=True)
