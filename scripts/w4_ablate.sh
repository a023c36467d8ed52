#!/bin/bash
# Timing ablations of the plane GEMM (dev builds -DUOC_W4_ABLATE=n under csrc/build/abl<n>/): which part of the loop costs what
R=$GRAFT_REPO_ROOT; cd $R
for n in 0 1 2 4; do
  lib=$R/unseenobjectclustering_amd/libuoc_hip.so; [ $n -gt 0 ] && lib=$R/unseenobjectclustering_amd/csrc/build/abl$n/libuoc_hip.so
  echo "== ablate $n"; UOC_LIB_PATH=$lib WINO4_BENCH_ONLY="8 waves auto" timeout 120 python scripts/wino4_bench.py 2>&1 | grep -E "^s[12]|8 waves auto"
done
