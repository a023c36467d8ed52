#!/bin/bash
# Frames-in-flight sweep on the GPU box: bench.py at inflight 1/2/3 (cooperative and plain launch of the sampling kernel)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/inflight; mkdir -p $O
cd $R
for coop in 1 0; do for n in 1 2 3; do
  UOC_FPS_COOP=$coop timeout 240 python bench.py --steps 40 --warmup 5 --inflight $n --cpu-frames 0 --profile-steps 0 --sustained-seconds 0 \
     > $O/coop${coop}_n$n.json 2> $O/coop${coop}_n$n.err
  echo "coop=$coop inflight=$n: $(python -c "import json;d=json.load(open('$O/coop${coop}_n$n.json'));print(d['value'],'fps',d['ms_per_step'],'ms')" 2>&1 | tail -1)"
done; done
