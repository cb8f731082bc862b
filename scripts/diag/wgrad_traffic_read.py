"""Reads the counter CSV of `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python scripts/diag/wgrad_traffic.py`."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
per = {}
for row in csv.DictReader(open(f)):
    if row.get("Counter_Name") != "FETCH_SIZE" or "k_mlp_wgrad" not in row["Kernel_Name"]:
        continue
    did = int(row["Dispatch_Id"])
    per[did] = per.get(did, 0.0) + float(row["Counter_Value"])
vals = [per[k] * 1024 * 2 / 1e6 for k in sorted(per)]          # MB (gfx950: 128-B requests tallied at 64 B)
for i in range(0, len(vals), 3):
    print(f"config {i // 3}: fetched {sum(vals[i:i + 3]) / len(vals[i:i + 3]):8.1f} MB per launch")
