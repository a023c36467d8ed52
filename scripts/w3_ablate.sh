#!/bin/bash
# Timing ablations of the split-precision plane GEMM (libraries built with -DW3_ABLATE=n under gpurun_tmp_abl/): which part of the
# loop costs what.  Build in the container:  scripts/w3_ablate.sh build ; run through gpurun: scripts/w3_ablate.sh run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
if [ "$1" = build ]; then
  mkdir -p gpurun_tmp_abl
  for n in ${ABL:-1 2 3 4 5 6}; do
    objs=""
    for f in unseenobjectclustering_amd/csrc/*.hip; do b=$(basename $f .hip)
      if [ $b = wino4_split ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DW3_ABLATE=$n -c $f -o gpurun_tmp_abl/$b.$n.o; objs="$objs gpurun_tmp_abl/$b.$n.o"
      else objs="$objs unseenobjectclustering_amd/csrc/build/$b.o"; fi
    done
    hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_tmp_abl/lib$n.so $objs
  done; ls -la gpurun_tmp_abl/*.so; exit 0
fi
for n in ${ABL:-0 1 2 3 4 5 6}; do
  lib=$R/unseenobjectclustering_amd/libuoc_hip.so; [ $n -gt 0 ] && lib=$R/gpurun_tmp_abl/lib$n.so
  echo "== ablate $n"; UOC_LIB_PATH=$lib SPLIT_BENCH_SHAPES="${2:-layer4}" timeout 120 python scripts/split_bench.py 2>&1 | grep -E "^s[12]|bf16x3"
done
