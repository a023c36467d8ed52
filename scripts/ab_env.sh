#!/bin/bash
# Same-box A/B of environment knobs through the driver-form bench (20 steps) + sustained 4 s.  Usage: scripts/ab_env.sh reps "A=1 B=2" "-" ...
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/ab_env; mkdir -p $O
reps=$1; shift
for i in $(seq 1 $reps); do n=0
  for arm in "$@"; do n=$((n+1)); [ "$arm" = "-" ] && a="A=1" || a="$arm"
    env $a UOC_BENCH_FULL=$O/$n.full.json timeout 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --profile-steps 0 --sustained-seconds 4 --skip-latency > $O/$n.json 2> $O/$n.err
    python - $O/$n.json "$arm" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"[{sys.argv[2]}] value {d['value']} sustained {d.get('sustained_frames_per_s')} pcie {d.get('pcie_inclusive_frames_per_s')} host {d.get('per_rank')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
done
