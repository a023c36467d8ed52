cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_meanshift_gpu.py tests/test_pipeline_gpu.py tests/test_twostage_gpu.py -x -q > gpurun_out/r3c_ms.log 2>&1; echo "ms rc=$?"; tail -2 gpurun_out/r3c_ms.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r3c_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3c_smoke.log
timeout 600 python bench.py > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err; echo "bench rc=$?"
for cfg in "3 2" "4 2" "3 3" "2 4" "4 4" "3 6" "2 6"; do set -- $cfg; timeout 200 python bench.py --steps 48 --cpu-frames 0 --sustained-seconds 5 --skip-pcie --profile-steps 0 --inflight $1 --frames-per-launch $2 > gpurun_out/r3c_sched_$1x$2.json 2> gpurun_out/r3c_sched_$1x$2.err; echo "sched $1x$2 rc=$?"; done
python - <<'PY'
import json,glob
d=json.load(open("gpurun_out/r3c_bench.json"))
print("bench", d["value"], d["sustained"]["frames_per_s"], "latency", d["latency"], "pcie", d["pcie_inclusive_frames_per_s"])
print("parity", {k:v for k,v in d["parity"].items() if k not in ("note","against")})
print("cpu", d["cpu_baseline"])
r=d["roofline"]; print("roofline", {k:v for k,v in r.items() if k!="by_shape"})
for x in d["kernels"][:8]: print(x)
for f in sorted(glob.glob("gpurun_out/r3c_sched_*.json")):
    try:
        x=json.load(open(f)); print(f, x["value"], x["sustained"]["frames_per_s"])
    except Exception as e: print(f,"failed",e)
PY
