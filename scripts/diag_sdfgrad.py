import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgs_nerf_amd import synth
from fgs_nerf_amd.losses import render_losses
from oracle import oracle as O
dev = torch.device('cuda:0')
G, N = 48, 512
rays_c = synth.random_rays(N, seed=21)
rays = tuple(r.to(dev) for r in rays_c)
target_c = torch.rand(N, 3, generator=torch.Generator().manual_seed(4))
target = target_c.to(dev)
lossw = dict(synth.FINE_LOSS, weight_rgbper=0.05)
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
def step(model):
    for p in model.parameters(): p.grad = None
    res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
    render_losses(res, target, lossw, model).backward()
    return model.sdf.grid.grad.clone(), model.k0.grid.grad.clone()
a = synth.build_model(G, synth.FINE_MODEL, device=dev, fused=True)
b = synth.build_model(G, synth.FINE_MODEL, device=dev, fused=False)
fa1, ka1 = step(a); fa2, ka2 = step(a); fb1, kb1 = step(b); fb2, kb2 = step(b)
print('fused run-to-run sdf', rel(fa1, fa2), 'composed run-to-run', rel(fb1, fb2), 'fused vs composed', rel(fa1, fb1))
# CPU oracle (sequential fp32 accumulation)
P = synth.oracle_params(b)
P['sdf'].requires_grad_(True); P['k0'].requires_grad_(True)
res = O.forward_fine(P, *rays_c, global_step=1000, near=2.0, stepsize=0.5, bg=1)
render_losses(res, target_c, lossw).backward()
print('fused vs oracle', rel(fa1, P['sdf'].grad), 'composed vs oracle', rel(fb1, P['sdf'].grad), ' k0: fused', rel(ka1, P['k0'].grad), 'composed', rel(kb1, P['k0'].grad))
# float64 oracle for the torch part
P64 = synth.oracle_params(b)
for k in ('sdf','k0'): P64[k] = P64[k].double().requires_grad_(True)
for net in ('rgbnet','refnet'): P64[net] = [(w.double(), bb.double()) for w, bb in P64[net]]
for k in ('xyz_min','xyz_max','posfreq','viewfreq','reffreq'): P64[k] = P64[k].double()
P64['voxel_size'] = P64['voxel_size'].double()
try:
    r64 = O.forward_fine(P64, rays_c[0].double(), rays_c[1].double(), rays_c[2].double(), global_step=1000, near=2.0, stepsize=0.5, bg=1)
    render_losses(r64, target_c.double(), lossw).backward()
    print('vs float64 oracle: fused', rel(fa1, P64['sdf'].grad), 'composed', rel(fb1, P64['sdf'].grad), 'cpu fp32 oracle', rel(P['sdf'].grad, P64['sdf'].grad))
except Exception as e:
    print('float64 oracle failed:', repr(e)[:300])
