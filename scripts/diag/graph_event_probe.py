"""Can a hipGraph replay carry timing events?  torch.cuda.Event(enable_timing=True, external=True) recorded during capture
becomes an event-record NODE (hipEventRecordExternal); after a replay, elapsed_time between two such events would give the
duration of the kernels between them inside the replay -- what bench.py's roofline needs."""
import torch

dev = torch.device("cuda", 0)
x = torch.randn(4096, 4096, device=dev)
y = torch.empty_like(x)
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
except TypeError as e:
    print("[probe] torch.cuda.Event has no `external` argument:", e)
    raise SystemExit(1)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    torch.mm(x, x, out=y)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        e0.record()
        torch.mm(x, x, out=y)
        e1.record()
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        print(f"[probe] replay {i}: elapsed between the two captured events = {e0.elapsed_time(e1):.3f} ms", flush=True)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(); g.replay(); t1.record(); torch.cuda.synchronize()
    print(f"[probe] the same replay bracketed by ordinary events: {t0.elapsed_time(t1):.3f} ms")
except Exception as e:      # noqa: BLE001
    print(f"[probe] FAILED: {type(e).__name__}: {e}")
    raise SystemExit(1)
