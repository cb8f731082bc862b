"""Where does the host lose time in a long run?  Per-phase host wall times of bench.train_step, slow steps listed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fgs_nerf_amd import synth
from fgs_nerf_amd.dist import GradAverager
from fgs_nerf_amd.losses import fused_render_losses

dev = torch.device("cuda:0")
model = synth.build_model(bench.GRID, synth.FINE_MODEL, device=dev)
opt = bench.make_optimizer(model)
avg = GradAverager(model.parameters())
batches = []
for b in range(8):
    ro, rd, vd = synth.random_rays(4096, seed=synth.SEED + 97 * b)
    tgt = torch.rand(4096, 3, generator=torch.Generator().manual_seed(b))
    batches.append(tuple(t.to(dev).contiguous() for t in (ro, rd, vd, tgt)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rec = []


def cpustat():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
        return {k: int(v) for k, v in d.items()}
    except Exception:
        return {}


try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except Exception as e:
    print("no cpu.max", e)
cs0 = cpustat()
t_start = time.perf_counter()
for i in range(steps):
    ro, rd, vd, tgt = batches[i % 8]
    t0 = time.perf_counter()
    res = model(ro, rd, vd, global_step=1000, **synth.RENDER_KWARGS)
    t1 = time.perf_counter()
    loss = fused_render_losses(res, tgt, synth.FINE_LOSS, model)
    opt.zero_grad(set_to_none=True)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    avg.average()
    model.sdf_total_variation_add_grad(0.01 * 0.1 / 4096, True)
    opt.step()
    t4 = time.perf_counter()
    rec.append((t0 - t_start, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    if i % 50 == 0:
        print(f'  mem step {i}: allocated {torch.cuda.memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB', flush=True)
torch.cuda.synchronize()
import statistics
for name, k in (("forward(+sync)", 1), ("loss", 2), ("backward", 3), ("tv+adam", 4)):
    v = [r[k] * 1e3 for r in rec[10:]]
    print(f"{name:14s} median {statistics.median(v):6.3f} ms  p95 {sorted(v)[int(0.95*len(v))]:6.3f}  max {max(v):7.3f}")
tot = [sum(r[1:]) * 1e3 for r in rec]
for i in range(0, steps, max(1, steps // 12)):
    r = rec[i]
    print(f"step {i:4d} t={r[0]:6.2f}s total {tot[i]:7.3f} ms  fwd {r[1]*1e3:6.3f} loss {r[2]*1e3:6.3f} bwd {r[3]*1e3:6.3f} opt {r[4]*1e3:6.3f}")
print("threads:", torch.get_num_threads(), "affinity:", len(os.sched_getaffinity(0)))
cs1 = cpustat()
wall = time.perf_counter() - t_start
print("loop wall %.2f s; cgroup deltas:" % wall, {k: cs1[k] - cs0[k] for k in cs1 if k in cs0})
