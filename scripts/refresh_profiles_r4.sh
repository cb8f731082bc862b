#!/bin/bash
# Regenerates the round's evidence on the GPU box (run through gpurun from the repo root; ~6 min) into gpurun_out/r4p/, in the
# form it is committed under profiles/r04_*:
#   fine          rocprofv3 --kernel-trace --memory-copy-trace --stats of the default bench.py (one hipGraph replay per step)
#   forced        the same with FGS_FORCE_DIST=1: the N > 1 step (RCCL collectives + device-counted brick exchange inside the graph)
#                 rehearsed with a single-rank RCCL group
#   coarse, 320   --stage coarse, --grid 320
#   collapse      FGS_MLP_COLLAPSE=1 (labelled mode)
#   bench_*.json  un-profiled bench lines: default (with cpu_baseline and live PMC traffic), 8192 rays at 160^3 and 256^3,
#                 forced-dist captured / eager, collapse
#   pmc_sq        SQ counters of the MLP kernels (one --pmc pass; durations from the un-profiled trace)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv"
$P -d $OUT/fine -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --warmup 8 > $OUT/fine.log 2>&1
FGS_FORCE_DIST=1 $P -d $OUT/forced -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --warmup 8 > $OUT/forced.log 2>&1
$P -d $OUT/coarse -- python3 $ROOT/bench.py --stage coarse --no-cpu-baseline --no-pmc --warmup 8 > $OUT/coarse.log 2>&1
$P -d $OUT/g320 -- python3 $ROOT/bench.py --grid 320 --no-cpu-baseline --no-pmc --steps 20 --warmup 8 > $OUT/g320.log 2>&1

cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --rays 8192 --no-cpu-baseline --no-pmc > $OUT/bench_160_8192.json 2> /dev/null
python3 bench.py --grid 256 --rays 8192 --no-cpu-baseline --no-pmc > $OUT/bench_256_8192.json 2> /dev/null
python3 bench.py --grid 320 --no-cpu-baseline --no-pmc > $OUT/bench_320.json 2> /dev/null
python3 bench.py --stage coarse --no-cpu-baseline --no-pmc > $OUT/bench_coarse.json 2> /dev/null
FGS_FORCE_DIST=1 python3 bench.py --no-cpu-baseline > $OUT/bench_forced_graph.json 2> /dev/null
FGS_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --mode eager > $OUT/bench_forced_eager.json 2> /dev/null
FGS_FORCE_DIST=1 python3 bench.py --grid 320 --no-cpu-baseline > $OUT/bench_forced_graph_320.json 2> /dev/null

for t in fine forced coarse; do python3 scripts/trace_summary.py $OUT/$t 30 graph > $OUT/sum_$t.txt; done
python3 scripts/trace_summary.py $OUT/g320 20 graph > $OUT/sum_g320.txt
python3 scripts/trace_step.py $OUT/fine median > $OUT/timeline_fine.txt
python3 scripts/trace_step.py $OUT/forced median > $OUT/timeline_forced.txt
for t in fine forced coarse g320; do
  cp $(ls $OUT/$t/*/*_kernel_stats.csv | head -1) $OUT/kernel_stats_$t.csv
  cp $(ls $OUT/$t/*/*_memory_copy_stats.csv 2>/dev/null | head -1) $OUT/memory_copy_stats_$t.csv 2>/dev/null || true
done
cd /tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --mode eager --steps 6 --warmup 2 > $OUT/pmc_sq.log 2>&1
cd $ROOT
python3 scripts/pmc_sq_summary.py $OUT/pmc_sq $OUT/fine $OUT/pmc_sq.json > $OUT/pmc_sq.txt 2>&1 || true
rm -rf $OUT/fine $OUT/forced $OUT/coarse $OUT/g320 $OUT/pmc_sq
# the real iterations (shipped train bodies through TrainStepper.run_captured) and the coarse one's timeline
python3 scripts/fine_real_iter.py > $OUT/fine_real_iter.txt 2>&1 || true
bash scripts/r4_coarse_real.sh r4p > /dev/null 2>&1 || true
python3 - <<P
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"))
    except Exception as e: print(f, "unreadable", e)
P
