import sys, os
sys.path.insert(0, os.getcwd())
import torch
from fgs_nerf_amd import nerf_training as nt, synth
dev = torch.device("cuda:0")
cfg = dict(N_iters=20000, N_rand=512, lrate_k0=0.1, lrate_sdf=0.005, lrate_rgbnet=1e-3, lrate_refnet=1e-3, lrate_decay=20,
           ray_sampler='flatten', weight_main=1.0, weight_entropy_last=0.001, weight_rgbper=0.0, weight_tv_density=0.01,
           weight_tv_k0=0.0, sigmoid_rgb_loss=0.02, weight_orientation=1e-4, tv_every=3, tv_from=0, tv_end=30000,
           voxel_inc=False, pg_scale=[], reset_iter=[], tv_terms=dict(sdf_tv=0.1, smooth_grad_tv=0.05),
           tv_dense_before=20000, cosine_lr=True,
           cosine_lr_cfg=dict(warm_up_iters=0, const_warm_up=True, warm_up_min_ratio=1.0), decay_step_module={},
           skip_zero_grad_fields=['density', 'k0', 'k1'])
R, FIRST, N = 2048, 898, 7
rays = tuple(r.to(dev) for r in synth.random_rays(R, seed=31))
target = torch.rand(R, 3, generator=torch.Generator().manual_seed(7)).to(dev)
runs = []
for mode in ("steps", "steps", "captured", "captured"):
    model = synth.build_model(48, synth.FINE_MODEL, device=dev)
    st = nt.TrainStepper(model, cfg, {}, synth.RENDER_KWARGS, target, *rays, stage='fine', seed=13)
    if mode == "steps":
        for g in range(FIRST, FIRST + N): st.step(g)
    else:
        st.run_captured(FIRST, N)
    torch.cuda.synchronize()
    runs.append([p.detach().clone() for p in model.parameters()])
def worst(a, b): return max(float((x - y).norm() / x.norm().clamp_min(1e-30)) for x, y in zip(a, b))
print("eager vs eager      ", worst(runs[0], runs[1]))
print("captured vs captured", worst(runs[2], runs[3]))
print("eager vs captured   ", worst(runs[0], runs[2]), worst(runs[1], runs[3]))
