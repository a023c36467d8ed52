cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for m in 128 64; do UOC_WINOGRAD_MIN_CIN=$m timeout 200 python bench.py --steps 48 --cpu-frames 0 --sustained-seconds 5 --skip-pcie > gpurun_out/r3j_bench_m$m.json 2> gpurun_out/r3j_bench_m$m.err; python - <<PY
import json
d=json.load(open("gpurun_out/r3j_bench_m$m.json")); print("mincin$m", d["value"], d["sustained"]["frames_per_s"], d["latency"]["frames_per_s"])
for x in d["kernels"][:9]: print("   ", x["kernel"], x["launches_per_frame"], x["avg_us"], x["gpu_time_share"])
PY
done
