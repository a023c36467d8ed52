cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python scripts/wino4_bench.py > gpurun_out/r3b_wino4_bench.log 2>&1; echo "w4bench rc=$?"
timeout 300 python bench.py --steps 24 --cpu-frames 0 --sustained-seconds 6 --skip-pcie > gpurun_out/r3b_bench_f4.json 2> gpurun_out/r3b_bench_f4.err; echo "bench f4 rc=$?"
UOC_WINOGRAD_F=2 timeout 300 python bench.py --steps 24 --cpu-frames 0 --sustained-seconds 6 --skip-pcie > gpurun_out/r3b_bench_f2.json 2> gpurun_out/r3b_bench_f2.err; echo "bench f2 rc=$?"
for pk in 2 3 4; do UOC_FPS_PACK=$pk timeout 300 python bench.py --steps 24 --cpu-frames 0 --sustained-seconds 6 --skip-pcie --profile-steps 0 > gpurun_out/r3b_bench_f4_pack$pk.json 2> gpurun_out/r3b_bench_f4_pack$pk.err; echo "bench pack$pk rc=$?"; done
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r3b_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^FAILED" gpurun_out/r3b_gpu_tests.log | tail -20
python - <<'PY'
import json
for n in ("f4","f2","f4_pack2","f4_pack3","f4_pack4"):
    try:
        d=json.load(open(f"gpurun_out/r3b_bench_{n}.json"))
        print(n, d["value"], d["sustained"]["frames_per_s"] if d.get("sustained") else None, d.get("roofline"))
    except Exception as e: print(n,"failed",e)
PY
