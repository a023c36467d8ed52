#!/bin/bash
# Latency leg in isolation: timings with and without graph replay, then a kernel trace of each for the gap analysis.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/latency_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in 1 0 1 0; do timeout 200 python $R/scripts/latency_leg.py --graph $g 2>&1 | grep graph=; done | tee $O/timings.txt
for g in 1 0; do
  timeout 300 rocprofv3 --kernel-trace -d $O/trace_g$g -o t -- python $R/scripts/latency_leg.py --graph $g --reps 2 > $O/under_rocprof_g$g.txt 2> $O/trace_g$g.err
  db=$(find $O/trace_g$g -name "*.db" | head -1)
  python $R/scripts/rocpd_gaps.py $db 0.3 > $O/gaps_g$g.md
  python $R/scripts/rocpd_stats.py $db 0.3 > $O/kernel_stats_g$g.md
  find $O/trace_g$g -name "*.db" -size +40M -delete
done
head -25 $O/gaps_g1.md; head -12 $O/gaps_g0.md
