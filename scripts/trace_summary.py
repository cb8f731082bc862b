"""Per-step kernel table of bench.py's TIMED region from a rocprofv3 --kernel-trace (csv) run.

A step starts at a marker kernel (graph mode: k_scalars_tick, launched once per captured iteration; eager mode: the march
kernel).  In graph mode the run holds 1 eager warm-up tick + 2 untimed replays + `steps` timed replays; the timed ones are the
last `steps` markers that are followed by another marker or by the end of the replays.  Printed: per-kernel us/step over the
timed steps, the sum of kernel time per step, the marker-to-marker span per step (wall time as the GPU saw it), the idle time
inside a step, and any memory copies (from *_memory_copy_trace.csv when the run had --memory-copy-trace).

    python scripts/trace_summary.py gpurun_out/r02_fine 30 [graph|eager] [top_n]
"""
import csv
import glob
import sys
from collections import defaultdict

d, steps = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "graph"
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
f = max(glob.glob(d + "/*/*_kernel_trace.csv"), key=__import__("os").path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if mode == "graph":
    marks = [i for i, r in enumerate(rows) if "k_scalars_tick" in r["Kernel_Name"]]
else:
    marks = [i for i, r in enumerate(rows) if "k_march_fine_fwd" in r["Kernel_Name"] or "k_march_coarse_fwd" in r["Kernel_Name"]]
# graph mode: the warm-up tick and the two untimed replays come first, the timed replays are the last `steps` markers; the
# LAST timed step is left out of the statistics (its end is not delimited by a marker: the eager profiling steps follow).
# eager mode: the last steps + 1 markers delimit `steps` steps (bench.py --mode eager has no profiling steps behind them).
bounds = marks[-steps:] if mode == "graph" else marks[-(steps + 1):]
n = len(bounds) - 1
per = defaultdict(lambda: [0, 0.0])
busy_tot = span_tot = idle_tot = 0.0
spans = []
for s in range(n):
    seg = rows[bounds[s]:bounds[s + 1]]
    t0 = int(seg[0]["Start_Timestamp"])
    t1 = int(rows[bounds[s + 1]]["Start_Timestamp"])
    end_prev, idle = t0, 0.0
    for r in seg:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:90]
        per[name][0] += 1
        per[name][1] += b - a
        busy_tot += b - a
        idle += max(a - end_prev, 0)
        end_prev = max(end_prev, b)
    idle_tot += idle
    span_tot += t1 - t0
    spans.append(t1 - t0)
print(f"# {f}")
print(f"timed steps found: {n} ({mode} mode)")
print(f"kernel time per step (sum over kernels): {busy_tot / n / 1e6:.4f} ms")
print(f"step span (marker to marker):            {span_tot / n / 1e6:.4f} ms   (median {sorted(spans)[len(spans) // 2] / 1e6:.4f}; "
      f"a host hiccup under the profiler shows up as one long step)")
print(f"GPU idle inside a step:                  {idle_tot / n / 1e3:.1f} us")
print(f"kernel time hidden by overlap:           {(busy_tot - (span_tot - idle_tot)) / n / 1e3:.1f} us   (k_mlp_wgrad runs on a "
      f"graph branch of its own beside the scatter kernels: their durations below include the time they share the CUs)")
for name, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / n / 1e3:8.1f} us/step {c / n:6.2f} calls {t / c / 1e3:8.1f} us  {name}")
mc = sorted(glob.glob(d + "/*/*_memory_copy_trace.csv"))
if mc:
    cps = list(csv.DictReader(open(mc[-1])))
    lo, hi = int(rows[bounds[0]]["Start_Timestamp"]), int(rows[bounds[-1] - 1]["End_Timestamp"])
    inside = [c for c in cps if lo <= int(c["Start_Timestamp"]) <= hi]
    kinds = defaultdict(int)
    for c in inside:
        kinds[c.get("Direction", "?")] += 1
    print(f"memory copies inside the timed region ({n} steps): {dict(kinds)}")
