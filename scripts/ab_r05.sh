#!/bin/bash
# Same-box A/B of the working tree against an exported older tree (gpurun_tmp_r05/, built in the container) and of the
# flow knobs of round 6.  Usage through gpurun: scripts/ab_r05.sh [reps]
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ab_r05; mkdir -p $O
reps=${1:-2}
run() { # tag dir env...
  tag=$1; dir=$2; shift 2
  ( cd $dir && env "$@" UOC_BENCH_FULL=$O/$tag.full.json timeout 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --profile-steps 0 --sustained-seconds 4 --skip-pcie > $O/$tag.json 2> $O/$tag.err )
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "value", d["value"], "sustained", d.get("sustained_frames_per_s"), "latency", d.get("latency"), "host", d.get("per_rank"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for i in $(seq 1 $reps); do
  run r05_$i $R/gpurun_tmp_r05 A=1
  run head_$i $R A=1
  run head_hostorder_$i $R UOC_HOST_ORDER=1
  run head_nograph_$i $R UOC_GRAPH_REPLAY=0
  run head_q8_$i $R GPU_MAX_HW_QUEUES=8
done
