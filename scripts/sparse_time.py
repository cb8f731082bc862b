import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from fgs_nerf_amd import synth
from fgs_nerf_amd.dist import GradAverager
from fgs_nerf_amd.losses import fused_render_losses
dev = torch.device('cuda:0')
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = synth.build_model(160, synth.FINE_MODEL, device=dev)
rays = tuple(r.to(dev) for r in synth.random_rays(4096)); target = torch.rand(4096, 3, device=dev)
res = model(*rays, global_step=1000, **synth.RENDER_KWARGS)
fused_render_losses(res, target, synth.FINE_LOSS, model).backward()
g = model.k0.grid.grad
avg = GradAverager(model.parameters()); avg.world_size = 2
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    avg._sparse(g, 1.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    h, flat = avg._dense(g, async_op=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: sparse {1e3*(t1-t0):.2f} ms (fill {avg.last_sparse_fill:.3f})  dense(1 rank) {1e3*(t2-t1):.2f} ms")
dist.destroy_process_group()
