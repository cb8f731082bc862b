"""Per-step GPU busy time and wall span from a rocprofv3 kernel trace of bench.py: is a slowdown GPU time or host gaps?

    python scripts/trace_drift.py gpurun_out/prof_x
"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at each march kernel
starts = [i for i, r in enumerate(rows) if "k_march_fine_fwd" in r["Kernel_Name"] or "k_march_coarse_fwd" in r["Kernel_Name"]]
print("steps:", len(starts))
for s in range(0, len(starts) - 1, max(1, (len(starts) - 1) // 30)):
    seg = rows[starts[s]:starts[s + 1]]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    span = int(rows[starts[s + 1]]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    gem = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if "k_gemm" in r["Kernel_Name"]]
    print(f"step {s:4d}: span {span/1e6:7.3f} ms  busy {busy/1e6:7.3f} ms  kernels {len(seg):4d}  gemm avg {sum(gem)/max(1,len(gem))/1e3:7.1f} us")
