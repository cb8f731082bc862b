"""cProfile of the host side of bench.train_step (main thread only; the autograd thread runs the backward functions)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fgs_nerf_amd import synth
from fgs_nerf_amd.dist import GradAverager

dev = torch.device("cuda:0")
model = synth.build_model(bench.GRID, synth.FINE_MODEL, device=dev)
opt = bench.make_optimizer(model)
avg = GradAverager(model.parameters())
batches = []
for b in range(8):
    ro, rd, vd = synth.random_rays(4096, seed=synth.SEED + 97 * b)
    tgt = torch.rand(4096, 3, generator=torch.Generator().manual_seed(b))
    batches.append(tuple(t.to(dev).contiguous() for t in (ro, rd, vd, tgt)))
for i in range(10):
    bench.train_step(model, opt, avg, batches[i % 8], 4096)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    bench.train_step(model, opt, avg, batches[i % 8], 4096)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
