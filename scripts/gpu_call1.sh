cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_backbone_gpu.py -x -q > gpurun_out/r3a_backbone.log 2>&1; echo "backbone rc=$?"
timeout 300 python scripts/wino4_bench.py > gpurun_out/r3a_wino4_bench.log 2>&1; echo "w4bench rc=$?"
timeout 300 python bench.py --steps 24 --cpu-frames 0 --sustained-seconds 6 --skip-pcie > gpurun_out/r3a_bench_f4.json 2> gpurun_out/r3a_bench_f4.err; echo "bench f4 rc=$?"
UOC_WINOGRAD_F=2 timeout 300 python bench.py --steps 24 --cpu-frames 0 --sustained-seconds 6 --skip-pcie > gpurun_out/r3a_bench_f2.json 2> gpurun_out/r3a_bench_f2.err; echo "bench f2 rc=$?"
UOC_FPS_PACK=4 timeout 300 python bench.py --steps 24 --cpu-frames 0 --sustained-seconds 6 --skip-pcie --profile-steps 0 > gpurun_out/r3a_bench_f4_pack4.json 2> gpurun_out/r3a_bench_f4_pack4.err; echo "bench pack rc=$?"
timeout 600 python -m pytest tests/test_meanshift_gpu.py tests/test_twostage_gpu.py tests/test_pipeline_gpu.py -x -q > gpurun_out/r3a_ms.log 2>&1; echo "ms rc=$?"
timeout 700 python -m pytest tests/test_headline_parity_gpu.py -x -q -s > gpurun_out/r3a_parity.log 2>&1; echo "parity rc=$?"
tail -3 gpurun_out/r3a_backbone.log; tail -3 gpurun_out/r3a_ms.log; tail -5 gpurun_out/r3a_parity.log
python - <<'PY'
import json
for n in ("f4","f2","f4_pack4"):
    try:
        d=json.load(open(f"gpurun_out/r3a_bench_{n}.json"))
        print(n, d["value"], d["sustained"]["frames_per_s"] if d.get("sustained") else None)
    except Exception as e: print(n,"failed",e)
PY
