"""Can two RCCL ranks share ONE GPU on this stack?  (A one-GPU box is all the build has: if yes, the captured multi-rank step
can be rehearsed with real inter-process collectives.)  Launch: python -m torch.distributed.run --nproc-per-node 2 ... this file.

Measured (round 3, RCCL 2.26.6): no -- "Duplicate GPU detected : rank 0 and rank 1 both on CUDA device 72000" (invalid usage), with or
without NCCL_IGNORE_DUPLICATE_GPU / RCCL_ENABLE_MULTIPLE_RANKS_PER_GPU.  The N > 1 path therefore stays covered by the gloo
world_size-2 tests plus the single-rank RCCL rehearsal (FGS_FORCE_DIST=1)."""
import os
import sys

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
try:
    dist.init_process_group("nccl", device_id=dev)
    t = torch.full((1024,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"[rank {rank}] all_reduce on a shared GPU: ok, value {float(t[0])}", flush=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    import time
    time.sleep(0.4)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            dist.all_reduce(t)
        g.replay()
    torch.cuda.synchronize()
    print(f"[rank {rank}] captured all_reduce replayed: value {float(t[0])}", flush=True)
    dist.destroy_process_group()
except Exception as e:      # noqa: BLE001
    print(f"[rank {rank}] FAILED: {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(1)
