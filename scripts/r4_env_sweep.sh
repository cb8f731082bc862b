#!/bin/bash
# Round-3 placement switches measured again under round 4's balance (the scatter branch now outlives the weight-gradient launch).
run() { env "$@" python bench.py --no-cpu-baseline --no-pmc --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$*', d['ms_per_step'], r['frac'], {k[:14]:v['avg_us'] for k,v in r['chains'].items()})"; }
run FGS_NOOP=1
run FGS_WGRAD_CUS=248
run FGS_WGRAD_CUS=240
run FGS_WGRAD_CUS=224
run FGS_K0_ADAM_LATE=1
run FGS_MARCH_FIRST=1
run FGS_K0_ADAM_DEFER=1
run FGS_NOOP=1
