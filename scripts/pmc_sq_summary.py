"""SQ counters of the MLP matrix-core kernels from one rocprofv3 PMC pass over bench.py --mode eager.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \\
        SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq -- \\
        python3 bench.py --no-cpu-baseline --no-pmc --mode eager --steps 6 --warmup 2
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_sq_trace -- python3 bench.py (same arguments)
    python scripts/pmc_sq_summary.py gpurun_out/pmc_sq gpurun_out/r_fine profiles/r02_pmc_sq.json   (durations: the default run)

SQ_VALU_MFMA_BUSY_CYCLES counts cycles (summed over SIMDs) in which the matrix pipe is busy; its share of the kernel's time is
busy / (4 SIMDs x CUs in use x kernel cycles).  Kernel cycles come from the UN-profiled kernel trace of the same command
(durations under --pmc are inflated) at the measured in-kernel clock (2.39 GHz, scripts/rc_bench.py stamps).
"""
import csv, glob, json, sys
from collections import defaultdict

KEYS = (("k_mlp_rc2<false", "k_mlp_rc2 forward chain"), ("k_mlp_rc2<true", "k_mlp_rc2 backward chain"),
        ("k_mlp_rc<false", "k_mlp_rc forward chain"), ("k_mlp_rc<true", "k_mlp_rc backward chain"), ("k_mlp_wgrad", "k_mlp_wgrad"))
CLOCK_GHZ = 2.39


def label_of(name):
    name = name.replace("(anonymous namespace)::", "")
    for key, lab in KEYS:
        if key in name:
            return lab
    return None


f = max(glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True), key=__import__("os").path.getmtime)
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # label -> counter -> dispatch -> value
for r in csv.DictReader(open(f)):
    lab = label_of(r["Kernel_Name"])
    if lab:
        acc[lab][r["Counter_Name"]][r.get("Dispatch_Id", "0")] += float(r["Counter_Value"])
dur = defaultdict(list)
t = max(glob.glob(sys.argv[2] + "/**/*_kernel_trace.csv", recursive=True), key=__import__("os").path.getmtime)
for r in csv.DictReader(open(t)):
    lab = label_of(r["Kernel_Name"])
    if lab:
        dur[lab].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY "
                  "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -- python3 bench.py "
                  "--no-cpu-baseline --no-pmc --mode eager --steps 6 --warmup 2; durations from an un-profiled --kernel-trace run of "
                  "the same command",
       "units": "SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*: quad-cycles summed "
                "over waves (MI355X_MICROARCH.md)",
       "kernels": {}}
for lab, ctr in acc.items():
    d = sorted(dur[lab])
    med_ns = d[len(d) // 2] if d else None
    k = {"launches_sampled": len(next(iter(ctr.values()))), "median_duration_us_unprofiled": None if med_ns is None else round(med_ns / 1e3, 1)}
    for c, v in ctr.items():
        k[c] = sum(v.values()) / len(v)
    if med_ns and "SQ_VALU_MFMA_BUSY_CYCLES" in k:
        cyc = med_ns * CLOCK_GHZ
        k["mfma_busy_fraction_of_kernel_time"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc), 3)
    if "SQ_WAVE_CYCLES" in k:
        k["fraction_of_wave_cycles"] = {c: round(k[c] / k["SQ_WAVE_CYCLES"], 3) for c in k
                                        if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES")}
    out["kernels"][lab] = k
    print(lab, json.dumps({a: b for a, b in k.items() if not isinstance(b, dict)}))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
