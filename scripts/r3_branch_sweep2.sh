#!/bin/bash
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3_sweep2}
mkdir -p $OUT
cd $ROOT
run() { local label=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-pmc --steps 60 --warmup 8 > $OUT/$label.json 2> $OUT/$label.err || echo "$label failed"; }
run base FGS_X=0
run late FGS_K0_ADAM_LATE=1
run late_m1 FGS_K0_ADAM_LATE=1 FGS_PRIO_MARCH_BWD=1
run late_m3t3 FGS_K0_ADAM_LATE=1 FGS_PRIO_MARCH_BWD=3 FGS_PRIO_TAPS_BWD=3
run end FGS_EARLY_ADAM=0
run base2 FGS_X=0
run late2 FGS_K0_ADAM_LATE=1
run m1 FGS_PRIO_MARCH_BWD=1
cd /tmp && export TMPDIR=/tmp
FGS_K0_ADAM_LATE=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_late -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --warmup 8 > $OUT/trace_late.log 2>&1
cd $ROOT
python3 scripts/trace_step.py $OUT/trace_late 25 > $OUT/step_late.txt 2>&1
rm -rf $OUT/trace_late
python3 - <<P
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "unreadable", e)
P
tail -n 16 $OUT/step_late.txt
