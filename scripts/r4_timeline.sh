#!/bin/bash
# One profiled run of the default bench.py: per-step kernel table and timeline of the captured step (gpurun_out/r4t/).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r4t}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/fine -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --warmup 8 > $OUT/fine.log 2>&1
cd $ROOT
python3 scripts/trace_summary.py $OUT/fine 30 graph > $OUT/sum_fine.txt
python3 scripts/trace_step.py $OUT/fine median > $OUT/timeline_fine.txt
cp $(ls $OUT/fine/*/*_kernel_stats.csv | head -1) $OUT/kernel_stats_fine.csv
rm -rf $OUT/fine
tail -3 $OUT/fine.log | cut -c1-300
