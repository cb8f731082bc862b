"""Host cost of one sync-free eager training step, phase by phase (the form `bench.py --gpus N` runs on N > 1 GPUs).

The problem is made tiny (16^3 grid, 256 rays) so that the GPU is never the limiter: what is measured is the time Python,
ctypes, torch's allocator and the autograd engine need to ENQUEUE a step -- the floor of an eager step whatever the GPU does.

    python scripts/host_phase.py [--force-dist] [--cprofile]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--force-dist", action="store_true", help="single-rank RCCL group: the exchange path runs too")
ap.add_argument("--cprofile", action="store_true")
ap.add_argument("--grid", type=int, default=16)
ap.add_argument("--rays", type=int, default=256)
args = ap.parse_args()
if args.force_dist:
    os.environ.update(FGS_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import torch
import bench
from fgs_nerf_amd import synth, fused
from fgs_nerf_amd.dist import GradAverager
from fgs_nerf_amd.losses import fused_render_losses

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if args.force_dist:
    import torch.distributed as dist
    dist.init_process_group(backend="nccl", device_id=dev)
bench.GRID = args.grid
model = synth.build_model(args.grid, synth.FINE_MODEL, device=dev)
opt = bench.make_optimizer(model)
avg = GradAverager(model.parameters(), force=args.force_dist)
avg.attach(model)
avg.attach_optimizer(opt)
if args.force_dist:
    fused.enable_early_update(model, opt, avg)
N = args.rays
batches = []
for b in range(8):
    ro, rd, vd = synth.random_rays(N, n_views=4, H=64, W=64, seed=synth.SEED + 97 * b)
    tgt = torch.rand(N, 3, generator=torch.Generator().manual_seed(b))
    batches.append(tuple(t.to(dev).contiguous() for t in (ro, rd, vd, tgt)))
for i in range(5):
    bench.train_step(model, opt, avg, batches[i % 8], N)
torch.cuda.synchronize()
fused.set_sync_free(model, 16384)
for i in range(5):
    bench.train_step(model, opt, avg, batches[i % 8], N)
torch.cuda.synchronize()

HOOK_T = {}
if args.force_dist:          # the exchange hooks run on the autograd thread, inside `backward`
    _early = avg.early

    def timed_early(kind, params, tensor=None):
        t = time.perf_counter()
        _early(kind, params, tensor)
        HOOK_T[kind] = HOOK_T.get(kind, 0.0) + time.perf_counter() - t
    model._fused_cache['grad_hook'] = timed_early

T = dict(forward=0.0, loss=0.0, hint=0.0, backward=0.0, average=0.0, tv=0.0, adam=0.0)


def step(batch):
    ro, rd, vd, target = batch
    t0 = time.perf_counter()
    res = model(ro, rd, vd, global_step=bench.GLOBAL_STEP, **synth.RENDER_KWARGS)
    t1 = time.perf_counter()
    loss = fused_render_losses(res, target, synth.FINE_LOSS, model)
    t2 = time.perf_counter()
    avg.hint_touched(model.k0.grid, res.get('survivor_pts'), model.xyz_min, model.xyz_max, count_ptr=res.get('survivor_count_ptr'))
    opt.zero_grad(set_to_none=True)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    avg.average()
    t5 = time.perf_counter()
    model.sdf_total_variation_add_grad(0.01 * 0.1 / N, True)
    t6 = time.perf_counter()
    opt.step()
    t7 = time.perf_counter()
    for k, d in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6)):
        T[k] += d


K = 300
if args.cprofile:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
t0 = time.perf_counter()
for i in range(K):
    step(batches[i % 8])
torch.cuda.synchronize()
el = time.perf_counter() - t0
if args.cprofile:
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
print(f"host-bound step: {el / K * 1e3:.3f} ms  (grid {args.grid}, {N} rays, force_dist={args.force_dist})")
for k, v in T.items():
    print(f"  {k:9s} {v / K * 1e3:.3f} ms")
for k, v in HOOK_T.items():
    print(f"    (inside backward) early('{k}') {v / K * 1e3:.3f} ms")
