#!/bin/bash
# Priority levels of the two vector-bound scatter kernels beside k_mlp_wgrad (FGS_PRIO_MARCH_BWD x FGS_PRIO_TAPS_BWD), 60-step runs.
for m in 1 2 3; do for t in 0 1 2 3; do
  FGS_PRIO_MARCH_BWD=$m FGS_PRIO_TAPS_BWD=$t python bench.py --no-cpu-baseline --no-pmc --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('march',$m,'taps',$t, d['ms_per_step'], r['frac'], {k[:14]:v['avg_us'] for k,v in r['chains'].items()})"
done; done
