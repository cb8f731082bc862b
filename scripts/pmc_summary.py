"""HBM traffic per launch of the MLP matrix-core kernels from two rocprofv3 PMC passes over bench.py (FETCH_SIZE and
WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes), with the guide's gfx950 corrections:
FETCH_SIZE is in KB and tallies 128-B requests at 64 B (x2); WRITE_SIZE (KB) is exact.

    python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_mlp.json
"""
import csv, glob, json, sys
from collections import defaultdict


def per_kernel(d, counter):
    f = max(glob.glob(d + "/*/*_counter_collection.csv"), key=__import__("os").path.getmtime)
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        for key in ("k_mlp_fwd", "k_linear_bwd", "k_gemm"):
            if name.startswith(key) or (" " + key) in name:
                acc[key if key != "k_gemm" else name.split("(")[0].replace("void ", "")][0] += 1
                acc[key if key != "k_gemm" else name.split("(")[0].replace("void ", "")][1] += float(r["Counter_Value"])
                break
    return {k: (n, v / n) for k, (n, v) in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python bench.py "
                  "--no-cpu-baseline --steps 6 --warmup 2   (two separate passes)",
       "workload": "bench.py default fine-stage step (160^3, 4096 rays, M_s ~ 50-64 K survivors)",
       "corrections": "bytes = FETCH_SIZE * 1024 * 2 (KB units; 128-B requests tallied at 64 B on gfx950), WRITE_SIZE * 1024 "
                      "(MI355X_MICROARCH.md, HBM section)",
       "kernels": {}}
for k in sorted(set(fetch) & set(write)):
    rd, wr = fetch[k][1] * 1024 * 2, write[k][1] * 1024
    out["kernels"][k] = {"launches_sampled": fetch[k][0], "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                         "traffic_bytes_per_launch": rd + wr}
    print(f"{k:28s} n={fetch[k][0]:4d}  read {rd/1e6:8.1f} MB  write {wr/1e6:8.1f} MB  total {(rd+wr)/1e6:8.1f} MB per launch")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
