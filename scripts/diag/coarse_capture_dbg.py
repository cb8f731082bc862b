"""Capture the coarse step at a given grid / ray count (hunting a hipStreamEndCapture crash at bench size)."""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fgs_nerf_amd import synth, fused
from fgs_nerf_amd.graph_step import CapturedStep
G, N = int(sys.argv[1]), int(sys.argv[2])
skip = set(sys.argv[3:])
dev = torch.device('cuda:0')
model = synth.build_model(G, synth.COARSE_MODEL, device=dev)
opt = bench.make_optimizer(model)
ro, rd, vd = synth.random_rays(N, seed=1)
batch = tuple(t.to(dev).contiguous() for t in (ro, rd, vd, torch.rand(N, 3)))
if 'novol4' in skip: fused.FLAGS['coarse_vol4'] = False
if 'nobrick' in skip: fused.FLAGS['brick_adam'] = False
tv = None if 'notv' in skip else (1e-6, True)
step = CapturedStep(model, opt, synth.COARSE_LOSS, synth.RENDER_KWARGS, N, n_iters=4, global_step_of=lambda it: 300,
                    lr_of=lambda it, g: g['lr'], tv=tv, capacity=int(os.environ.get('CAP', 40 * N)))
if 'eager' in skip:
    from fgs_nerf_amd.dist import GradAverager
    av = GradAverager(model.parameters())
    for _ in range(3):
        bench.train_step(model, opt, av, batch, N)
    torch.cuda.synchronize()
    print('eager steps done', bench.STEP_STATS, flush=True)
print('capturing', G, N, skip, flush=True)
step.capture(batch)
print('captured', flush=True)
step.replay(batch); torch.cuda.synchronize()
print('replayed', step.check(), flush=True)
