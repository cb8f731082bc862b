"""Does CapturedFineStep.capture() leave the parameters untouched (with the inline early k0 update enabled)?"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from fgs_nerf_amd import fused, synth
from fgs_nerf_amd.graph_step import CapturedFineStep
dev = torch.device("cuda:0")
N = 512
for early in (False, True):
    model = synth.build_model(48, synth.FINE_MODEL, device=dev)
    opt = bench.make_optimizer(model)
    ro, rd, vd = synth.random_rays(N, seed=50)
    batch = tuple(t.to(dev).contiguous() for t in (ro, rd, vd, torch.rand(N, 3)))
    if early:
        fused.enable_early_update(model, opt, None, inline=True)
    base = {id(g): g['lr'] for g in opt.param_groups}
    step = CapturedFineStep(model, opt, synth.FINE_LOSS, synth.RENDER_KWARGS, N, n_iters=4, global_step_of=lambda it: 1000 + it,
                            lr_of=lambda it, g: base[id(g)], tv=(1e-6, True), capacity=8192)
    k0_before = model.k0.grid.detach().clone()
    step.capture(batch)
    torch.cuda.synchronize()
    print("early", early, "k0 changed by capture():", not torch.equal(k0_before, model.k0.grid.detach()), "opt._early:", dict(opt._early))
    step.replay(batch); torch.cuda.synchronize()
    k1 = model.k0.grid.detach().clone()
    print("   k0 changed by replay 1:", not torch.equal(k0_before, k1), " step counters:", sorted(set(st['step'] for st in opt.state.values())))
    step.replay(batch); torch.cuda.synchronize()
    print("   k0 changed by replay 2:", not torch.equal(k1, model.k0.grid.detach()))
