#!/bin/bash
# Sweep of the co-scheduling knobs of the backward pass's two graph branches, plain captured step:
#   FGS_PRIO_MARCH_BWD / FGS_PRIO_TAPS_BWD : s_setprio level of the two VALU-heavy sdf kernels beside k_mlp_wgrad
#   FGS_WGRAD_CUS                          : workgroups of k_mlp_wgrad (whole CUs left to the other branch)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3_sweep}
mkdir -p $OUT
cd $ROOT
run() { # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-pmc --steps 60 --warmup 8 > $OUT/$label.json 2> $OUT/$label.err || echo "$label failed"
}
run base FGS_X=0
run base2 FGS_X=0
run m1 FGS_PRIO_MARCH_BWD=1
run t1 FGS_PRIO_TAPS_BWD=1
run m1t1 FGS_PRIO_MARCH_BWD=1 FGS_PRIO_TAPS_BWD=1
run m2t2 FGS_PRIO_MARCH_BWD=2 FGS_PRIO_TAPS_BWD=2
run m3t3 FGS_PRIO_MARCH_BWD=3 FGS_PRIO_TAPS_BWD=3
run cu248 FGS_WGRAD_CUS=248
run cu240 FGS_WGRAD_CUS=240
run cu224 FGS_WGRAD_CUS=224
run cu240_m1t1 FGS_WGRAD_CUS=240 FGS_PRIO_MARCH_BWD=1 FGS_PRIO_TAPS_BWD=1
run base3 FGS_X=0
python3 - <<P
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "unreadable", e)
P
