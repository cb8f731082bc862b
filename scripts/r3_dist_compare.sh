#!/bin/bash
# Same-box comparison of the captured step with and without the gradient exchange inside it (round 3):
#   plain        : python bench.py                      (one GPU, no exchange)
#   forced       : FGS_FORCE_DIST=1 python bench.py     (single-rank RCCL group: every collective of the N > 1 step is issued)
#   forced eager : ... --mode eager                      (round 2's N > 1 form)
# plus kernel traces of both captured forms (timeline of one step each).  Run through gpurun from the repo root.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3_dist}
mkdir -p $OUT
cd $ROOT
for i in 1 2; do
  python3 bench.py --no-cpu-baseline --no-pmc > $OUT/plain_$i.json 2> $OUT/plain_$i.err
  FGS_FORCE_DIST=1 python3 bench.py --no-cpu-baseline > $OUT/forced_$i.json 2> $OUT/forced_$i.err
done
FGS_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --mode eager > $OUT/forced_eager.json 2> $OUT/forced_eager.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_plain -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --warmup 8 > $OUT/trace_plain.log 2>&1
FGS_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_forced -- python3 $ROOT/bench.py --no-cpu-baseline --no-pmc --warmup 8 > $OUT/trace_forced.log 2>&1
cd $ROOT
python3 scripts/trace_step.py $OUT/trace_plain 25 > $OUT/step_plain.txt 2>&1
python3 scripts/trace_step.py $OUT/trace_forced 25 > $OUT/step_forced.txt 2>&1
rm -rf $OUT/trace_plain $OUT/trace_forced
python3 - <<P
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["step_mode"][:60])
P
