#!/bin/bash
# Regenerates the round's evidence on the GPU box (run through gpurun from the repo root):
#   gpurun_out/r_fine, r_coarse : rocprofv3 --kernel-trace --stats of bench.py --warmup 8 (fine / coarse stage; a warm-up
#                                 that covers the 8-batch cycle, so no allocator priming pass: 38 identical steps)
#   gpurun_out/r_bench.json     : the default bench.py line (with cpu_baseline)
# Copy what should be judged into profiles/ afterwards (scripts/prof_summary.py prints the per-step tables).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r_fine -- python3 $ROOT/bench.py --no-cpu-baseline --warmup 8 > $ROOT/gpurun_out/r_fine.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r_coarse -- python3 $ROOT/bench.py --stage coarse --no-cpu-baseline --warmup 8 > $ROOT/gpurun_out/r_coarse.log 2>&1
cd $ROOT
python3 bench.py > gpurun_out/r_bench.json 2> gpurun_out/r_bench.err
python3 scripts/prof_summary.py gpurun_out/r_fine 38 40 > gpurun_out/r_sum_fine.txt
python3 scripts/prof_summary.py gpurun_out/r_coarse 38 40 > gpurun_out/r_sum_coarse.txt
tail -1 gpurun_out/r_bench.json | cut -c1-400
