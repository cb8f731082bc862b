"""Print a per-step summary of a rocprofv3 --kernel-trace --stats run of bench.py.

    python scripts/prof_summary.py gpurun_out/prof_x [steps_total] [top_n]
"""
import csv
import glob
import sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
top = int(sys.argv[3]) if len(sys.argv) > 3 else 22
f = max(glob.glob(d + "/*/*_kernel_stats.csv"), key=__import__("os").path.getmtime)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot / steps / 1e6:.3f} ms   ({f})")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e3:8.1f} us/step {int(r['Calls']) / steps:6.1f} calls "
          f"{float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:105]}")
