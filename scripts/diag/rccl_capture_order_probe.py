"""Are two collectives of ONE process group, issued under capture from two different forked streams, ORDERED in the captured
graph?  (RCCL requires every rank to launch the collectives of a communicator in the same order; a hipGraph may run independent
branches in any order.)  ProcessGroupNCCL enqueues collectives on its own internal stream in issue order, or -- for synchronous
calls in recent PyTorch -- on the caller's current stream: in the second case two branches of a captured step would be unordered.
(hipGraphDebugDotPrint writes nothing on this stack, so the probe TIMES it.)  Captured: branch A = [spin ~T] -> all_reduce(x),
branch B = all_reduce(y) -> [spin ~T], x's collective issued first.  If the process group keeps its collectives in issue order
(an internal stream, or edges between them), y's cannot start before x's, which waits for A's spin: the replay takes ~2 T.  If
each collective simply sits on its caller's branch, the two branches are independent: ~T."""
import os
import re
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x, y = torch.ones(1 << 20, device=dev), torch.ones(1 << 18, device=dev)
dist.all_reduce(x); dist.all_reduce(y)
torch.cuda.synchronize()
import time
time.sleep(0.4)
def spin_ms(cycles):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(cycles); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


cycles = 20_000_000
spin_ms(cycles)
T = spin_ms(cycles)
g = torch.cuda.CUDAGraph()
main, a, b = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(main):
    with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
        x.mul_(1.0)
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(a):
            a.wait_event(ev)
            torch.cuda._sleep(cycles)
            dist.all_reduce(x)                      # issued FIRST
            ea = torch.cuda.Event(); ea.record()
        with torch.cuda.stream(b):
            b.wait_event(ev)
            dist.all_reduce(y)                      # issued SECOND
            torch.cuda._sleep(cycles)
            eb = torch.cuda.Event(); eb.record()
        main.wait_event(ea); main.wait_event(eb)
        y.mul_(1.0)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1)
print(f"spin alone {T:.2f} ms; replay of the two-branch graph {t:.2f} ms -> the two collectives of one process group are "
      f"{'ORDERED (issue order kept)' if t > 1.6 * T else 'NOT ordered (each sits on its caller branch)'}", flush=True)
# the same with the second collective on a process group of its own
grp2 = dist.new_group(backend="nccl")
dist.all_reduce(y, group=grp2); torch.cuda.synchronize(); time.sleep(0.4)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    with torch.cuda.graph(g2, stream=main, capture_error_mode="thread_local"):
        x.mul_(1.0)
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(a):
            a.wait_event(ev)
            torch.cuda._sleep(cycles)
            dist.all_reduce(x)
            ea = torch.cuda.Event(); ea.record()
        with torch.cuda.stream(b):
            b.wait_event(ev)
            dist.all_reduce(y, group=grp2)
            torch.cuda._sleep(cycles)
            eb = torch.cuda.Event(); eb.record()
        main.wait_event(ea); main.wait_event(eb)
        y.mul_(1.0)
    g2.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
    t2 = e0.elapsed_time(e1)
print(f"second collective on its own process group: {t2:.2f} ms -> {'ordered' if t2 > 1.6 * T else 'independent'}", flush=True)
del g, g2
dist.destroy_process_group()
