"""Does a hipGraph capture carry RCCL collectives issued through torch.distributed on this stack?

Single-rank RCCL group (a one-GPU box has no peers): capture {kernel -> fork to a side stream -> all_reduce there ->
join -> all_reduce(async_op=True) + wait on the main stream}, replay it, check the values and time the replays against
the same sequence issued eagerly.  Prints one line per check; exit code 0 = every capture replayed with correct values.
"""
import os
import sys
import time

import torch
import torch.distributed as dist


def main() -> int:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", device_id=dev)
    second = dist.new_group(backend="nccl")
    n_big, n_small = 50 << 20 >> 2, 2 << 20 >> 2
    a = torch.ones(n_big, device=dev)
    b = torch.ones(n_small, device=dev)
    c = torch.ones(16 << 20 >> 2, device=dev)
    # communicators are created lazily by the first collective: outside any capture
    for t, g in ((a, second), (b, second), (c, None)):
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=g)
    torch.cuda.synchronize()
    prio = int(os.environ.get("PROBE_PRIORITY", "0"))
    side = torch.cuda.Stream(device=dev, priority=prio)

    def body():
        a.mul_(2.0)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            dist.all_reduce(a, op=dist.ReduceOp.AVG, group=second)
            dist.all_reduce(b, op=dist.ReduceOp.AVG, group=second)
            done = torch.cuda.Event()
            done.record()
        h = dist.all_reduce(c, op=dist.ReduceOp.AVG, async_op=True)
        c2 = b * 1.0          # main-stream work beside the exchange
        h.wait()
        torch.cuda.current_stream().wait_event(done)
        a.mul_(0.5)
        return c2

    ok = True
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        print("[probe] capture with RCCL collectives on a forked stream: ok", flush=True)
    except Exception as e:          # noqa: BLE001
        print(f"[probe] capture FAILED: {type(e).__name__}: {e}", flush=True)
        dist.destroy_process_group()
        return 1
    a.fill_(3.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    good = bool((a == 3.0).all()) and bool((b == 1.0).all()) and bool((c == 1.0).all())
    print(f"[probe] values after 3 replays: {'ok' if good else 'WRONG'} (a[0]={float(a[0])})", flush=True)
    ok &= good
    for name, fn in (("graph replay", g.replay), ("eager", body)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        print(f"[probe] {name}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per iteration (50 + 2 + 16 MB reduced)", flush=True)
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
