"""Timeline of one training step from a rocprofv3 kernel trace: start offset, duration, stream/queue, kernel.

    python scripts/trace_step.py gpurun_out/prof_x [step_index | median]
"""
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=__import__("os").path.getmtime)
which = sys.argv[2] if len(sys.argv) > 2 else "20"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_march_fine_fwd" in r["Kernel_Name"] or "k_march_coarse_fwd" in r["Kernel_Name"]]
if which == "median":      # the replay whose span is the median of the trace's second half (the timed steps)
    cand = list(range(len(starts) // 2, len(starts) - 1))
    spans = sorted((int(rows[starts[i + 1]]["Start_Timestamp"]) - int(rows[starts[i]]["Start_Timestamp"]), i) for i in cand)
    which = spans[len(spans) // 2][1]
    print(f"# replay {which} of {len(starts)}: the median span of the trace's second half")
else:
    which = int(which)
seg = rows[starts[which]:starts[which + 1]]
t0 = int(seg[0]["Start_Timestamp"])
qk = "Queue_Id" if "Queue_Id" in seg[0] else ("Stream_Id" if "Stream_Id" in seg[0] else None)
end_prev = 0
idle = 0.0
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
    gap = (s - end_prev) / 1e3
    idle += max(gap, 0.0)
    print(f"{s/1e3:9.1f} us  +{(e-s)/1e3:7.1f}  gap {gap:6.1f}  q={r.get(qk, '?') if qk else '?':>3}  {name}")
    end_prev = max(end_prev, e)
print("GPU idle inside the step: %.1f us" % idle)
print("span %.1f us" % ((int(rows[starts[which + 1]]["Start_Timestamp"]) - t0) / 1e3))
