#!/bin/bash
# Same-box A/B of library builds through the driver-form bench (20 steps) + sustained 4 s + the latency leg.
# Usage: scripts/ab_lib.sh reps libA.so libB.so ...   ("-" = the shipped library)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ab_lib; mkdir -p $O
reps=$1; shift
for i in $(seq 1 $reps); do n=0
  for lib in "$@"; do n=$((n+1)); [ "$lib" = "-" ] && a="A=1" || a="UOC_LIB_PATH=$R/$lib"
    env $a UOC_BENCH_FULL=$O/$n.full.json timeout 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --profile-steps 0 --sustained-seconds 4 --skip-pcie > $O/$n.json 2> $O/$n.err
    python - $O/$n.json "$lib" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"[{sys.argv[2]}] value {d['value']} sustained {d.get('sustained_frames_per_s')} latency {d.get('latency')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
