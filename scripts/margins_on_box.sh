#!/bin/bash
# tests/golden/make_bench_margins.py for frames [$1, $2) in sub-blocks of 8 frames, one single-threaded process each, on the
# GPU box's host cores (CPU only; the outputs come back through gpurun_out/ and are merged into 64-frame files by
# tests/golden/merge_bench_margins.py)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/margins
for lo in $(seq $1 8 $(($2-1))); do
  ( UOC_MARGINS_OUT=$R/gpurun_out/margins timeout 1500 python tests/golden/make_bench_margins.py $lo $((lo+8)) 1 > gpurun_out/margins/log_$lo.txt 2>&1 ) &
done
wait
ls gpurun_out/margins | wc -l; tail -qn1 gpurun_out/margins/log_*.txt | cut -c1-60 | sort | uniq -c | head
