import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from fgs_nerf_amd import synth
from fgs_nerf_amd.dist import GradAverager
from fgs_nerf_amd.graph_step import CapturedFineStep
dev = torch.device('cuda:0')
def build():
    model = synth.build_model(160, synth.FINE_MODEL, device=dev)
    opt = bench.make_optimizer(model)
    return model, opt, GradAverager(model.parameters())
batches = []
for b in range(8):
    ro, rd, vd = synth.random_rays(4096, seed=synth.SEED + 97 * b)
    target = torch.rand(4096, 3, generator=torch.Generator().manual_seed(b))
    batches.append(tuple(t.to(dev).contiguous() for t in (ro, rd, vd, target)))
# eager trajectory
model, opt, av = build()
surv = []
for i in range(16):
    bench.STEP_STATS["survivors"] = 0
    bench.train_step(model, opt, av, batches[i % 8], 4096)
    surv.append(bench.STEP_STATS["survivors"])
print("eager survivors per step:", surv)
# graph trajectory: 8 eager then captured
model, opt, av = build()
for i in range(8):
    bench.train_step(model, opt, av, batches[i % 8], 4096)
cap = CapturedFineStep(model, opt, synth.FINE_LOSS, synth.RENDER_KWARGS, 4096, n_iters=32, global_step_of=lambda it: 1000,
                       lr_of=lambda it, g: g['lr'], tv=(0.01 * 0.1 / 4096, True), capacity=98304)
cap.capture(batches[0])
out = []
for i in range(8, 16):
    cap.clear_counters()
    cap.replay(batches[i % 8])
    out.append(cap.check()[1])
print("graph survivors per step (8..15):", out, "scalars", cap.scalars.cpu().tolist(), "counter", int(cap.counter))
print("table row 0..2", cap.table[:3].cpu().tolist())
