#!/bin/bash
# Profiled run of scripts/coarse_real_iter.py (114^3): per-kernel table of one real coarse-stage iteration (gpurun_out/$1/).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4c}
mkdir -p $OUT
python3 $ROOT/scripts/coarse_real_iter.py 114 60 > $OUT/coarse_real.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cr -- python3 $ROOT/scripts/coarse_real_iter.py 114 30 > $OUT/cr.log 2>&1
cd $ROOT
cp $(ls $OUT/cr/*/*_kernel_stats.csv | head -1) $OUT/kernel_stats_coarse_real.csv
python3 - $OUT/kernel_stats_coarse_real.csv > $OUT/coarse_real_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iters = 8 + 30 + 120 + 3      # warm + timed + 4x + capture passes (approximate: per-iteration = total / calls * calls-per-iteration)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"]) / tot * 100:6.2f} %  {int(r["Calls"]):6d} calls  {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:90]}')
PY
python3 scripts/trace_step.py $OUT/cr median > $OUT/coarse_real_timeline.txt
rm -rf $OUT/cr
cat $OUT/coarse_real.log
cut -c1-130 $OUT/coarse_real_timeline.txt
