#!/bin/bash
# A/B of one environment variable on one box: scripts/ab_env.sh VAR "v1 v2 ..." [bench.py args]; alternates the values twice.
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$VAR=$v', d['ms_per_step'], d['value'], r['frac'], {k[:22]:v['avg_us'] for k,v in r['chains'].items()})"
done; done
