cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for t in 16 24; do UOC_HC_VB_TILES=$t timeout 200 python bench.py --steps 24 --cpu-frames 0 --sustained-seconds 5 --skip-pcie > gpurun_out/r3i_bench_t$t.json 2> gpurun_out/r3i_bench_t$t.err; python - <<PY
import json
d=json.load(open("gpurun_out/r3i_bench_t$t.json")); print("tiles$t", d["value"], d["sustained"]["frames_per_s"], d["latency"]["frames_per_s"])
for x in d["clustering_by_shape"]: print("   ", x)
PY
done
timeout 600 python -m pytest tests/test_meanshift_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -2
