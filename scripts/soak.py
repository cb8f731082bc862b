"""Soak run of the training stepper at the bench size: N iterations, device memory and loss checked along the way."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgs_nerf_amd import nerf_training as nt, synth

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
stage = sys.argv[2] if len(sys.argv) > 2 else "fine"
cfg = synth.FINE_MODEL if stage == "fine" else synth.COARSE_MODEL
model = synth.build_model(160, cfg, device=dev)
R = 4096 * 16
rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=1))
target = torch.rand(R, 3, device=dev)
train = dict(N_iters=20000, N_rand=4096, lrate_k0=0.1, lrate_sdf=0.005, lrate_rgbnet=1e-3 if stage == "fine" else 0, lrate_refnet=1e-3,
             lrate_decay=20, ray_sampler='flatten', weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.0,
             weight_tv_density=0.01, weight_tv_k0=0.0, sigmoid_rgb_loss=0.02, weight_orientation=1e-4, tv_every=3, tv_from=0,
             tv_end=30000, voxel_inc=False, pg_scale=[], reset_iter=[], tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.05),
             tv_dense_before=20000, cosine_lr=True, cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0),
             decay_step_module={}, skip_zero_grad_fields=['density', 'k0', 'k1'])
st = nt.TrainStepper(model, train, {}, synth.RENDER_KWARGS, target, *rays, stage=stage, seed=0)
torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == "captured":
    # the same iterations as captured windows (TrainStepper.run_captured: two graphs, with / without the TV schedule active)
    W = 500
    t0 = time.perf_counter()
    for first in range(1, iters + 1, W):
        n = min(W, iters + 1 - first)
        tw = time.perf_counter()
        losses, overflow = st.run_captured(first, n)
        torch.cuda.synchronize()
        print(f"iters {first:6d}..{first + n - 1:6d}  {1e3 * (time.perf_counter() - tw) / n:6.3f} ms/iter (capture included)  "
              f"loss {float(losses[-1]):.5f}  overflow {overflow}  alloc {torch.cuda.memory_allocated() / 2**30:.2f} GiB  "
              f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
    print(f"total {1e3 * (time.perf_counter() - t0) / iters:6.3f} ms/iter")
    sys.exit(0)
t0 = time.perf_counter()
for g in range(1, iters + 1):
    loss = st.step(g)
    if g % 500 == 0 or g == iters:
        torch.cuda.synchronize()
        s = st.stats()
        print(f"iter {g:6d}  {1e3 * (time.perf_counter() - t0) / g:6.3f} ms/iter  loss {float(loss):.5f}  psnr {s['psnr']:.2f}  "
              f"alloc {torch.cuda.memory_allocated() / 2**30:.2f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB  "
              f"survivors {int(st.last_result['weights'].shape[0])}", flush=True)
assert torch.isfinite(loss)
