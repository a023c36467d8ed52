#!/bin/bash
# Same-box A/B of environment knobs through the whole benchmark (sustained rate over ~6 s per arm, plus the per-kernel
# table of the profiled pass).  Usage (through gpurun): scripts/ab.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   ("-" = no knobs)
#   -> gpurun_out/ab_<tag>/<i>.json (the full record of each arm) and a summary on stdout
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/ab_$T; mkdir -p $O
cd $R
i=0
for arm in "$@"; do i=$((i+1))
  [ "$arm" = "-" ] && arm=""
  env $arm UOC_BENCH_FULL=$O/$i.json timeout 300 python bench.py --steps 64 --warmup 4 --cpu-frames 0 --profile-steps 2 --sustained-seconds 6 --skip-pcie --skip-latency > $O/$i.line 2> $O/$i.err
  python - "$O/$i.json" "$arm" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = {r["kernel"]: r for r in d.get("kernels") or []}
    pick = lambda n: (f"{k[n]['avg_us']:.0f}us/{100 * k[n]['gpu_time_share']:.1f}%" if n in k else "-")
    print(f"[{sys.argv[2] or 'baseline'}] timed {d['value']} fps, sustained {d['sustained']['frames_per_s']} fps | "
          f"gemm {pick('wino4_gemm')} hc {pick('hc_iter')} fps {pick('fps_step')} w4in {pick('wino4_input')} w4out {pick('wino4_output')}")
except Exception as e:
    print(f"[{sys.argv[2]}] FAILED: {e}")
PY
done
