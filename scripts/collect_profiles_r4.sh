#!/bin/bash
# gpurun_out/r4p/* (scripts/refresh_profiles_r4.sh) -> profiles/r04_* under the names profiles/r04_README.md lists.
set -e
S=gpurun_out/r4p; D=profiles
for f in $S/bench_*.json; do cp $f $D/r04_$(basename $f); done
for t in fine forced coarse g320; do
  cp $S/sum_$t.txt $D/r04_${t}_step_summary.txt
  cp $S/kernel_stats_$t.csv $D/r04_${t}_kernel_stats.csv
  [ -f $S/memory_copy_stats_$t.csv ] && cp $S/memory_copy_stats_$t.csv $D/r04_${t}_memory_copy_stats.csv
done
cp $S/timeline_fine.txt $D/r04_fine_graph_step_timeline.txt
cp $S/timeline_forced.txt $D/r04_forced_graph_step_timeline.txt
cp $S/pmc_sq.json $D/r04_pmc_sq.json
[ -f $S/coarse_real_timeline.txt ] && cp $S/coarse_real_timeline.txt $D/r04_coarse_real_iteration_timeline.txt
[ -f $S/coarse_real.log ] && cp $S/coarse_real.log $D/r04_coarse_real_iteration.txt
[ -f $S/fine_real_iter.txt ] && cp $S/fine_real_iter.txt $D/r04_fine_real_iteration.txt
ls $D | grep -c r04_
