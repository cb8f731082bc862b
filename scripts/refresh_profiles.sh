#!/bin/bash
# Regenerates the round's evidence on the GPU box (run through gpurun from the repo root; ~4 min):
#   gpurun_out/r_fine    : rocprofv3 --kernel-trace --memory-copy-trace --stats of the default bench.py (fine stage, one
#                          hipGraph replay per step)
#   gpurun_out/r_eager   : the same with --mode eager (host-driven launches, one survivor-count read per step)
#   gpurun_out/r_coarse  : --stage coarse
#   gpurun_out/r_320     : --grid 320
#   gpurun_out/r_bench.json : the default bench.py line (with cpu_baseline), no profiler attached
# Copy what should be judged into profiles/ afterwards (scripts/trace_summary.py prints the per-step tables).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv"
$P -d $ROOT/gpurun_out/r_fine -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --warmup 8 > $ROOT/gpurun_out/r_fine.log 2>&1
$P -d $ROOT/gpurun_out/r_eager -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --warmup 8 --mode eager > $ROOT/gpurun_out/r_eager.log 2>&1
$P -d $ROOT/gpurun_out/r_coarse -- python3 $ROOT/bench.py --stage coarse --no-cpu-baseline --no-pmc --warmup 8 > $ROOT/gpurun_out/r_coarse.log 2>&1
$P -d $ROOT/gpurun_out/r_320 -- python3 $ROOT/bench.py --grid 320 --no-cpu-baseline --no-pmc --steps 20 --warmup 8 > $ROOT/gpurun_out/r_320.log 2>&1
cd $ROOT
python3 bench.py > gpurun_out/r_bench.json 2> gpurun_out/r_bench.err
python3 scripts/trace_summary.py gpurun_out/r_fine 30 graph > gpurun_out/r_sum_fine.txt
python3 scripts/trace_summary.py gpurun_out/r_eager 30 eager > gpurun_out/r_sum_eager.txt
python3 scripts/trace_summary.py gpurun_out/r_coarse 30 graph > gpurun_out/r_sum_coarse.txt
python3 scripts/trace_summary.py gpurun_out/r_320 20 graph > gpurun_out/r_sum_320.txt
tail -1 gpurun_out/r_bench.json | cut -c1-400
# SQ counters of the MLP kernels (one --pmc pass; durations from an un-profiled trace of the same command)
cd /tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $ROOT/gpurun_out/pmc_sq -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --mode eager --steps 6 --warmup 2 > $ROOT/gpurun_out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_sq_trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --mode eager --steps 6 --warmup 2 > $ROOT/gpurun_out/pmc_sq_trace.log 2>&1
cd $ROOT
python3 scripts/pmc_sq_summary.py gpurun_out/pmc_sq gpurun_out/r_fine gpurun_out/r_pmc_sq.json > gpurun_out/r_pmc_sq.txt 2>&1 || true
