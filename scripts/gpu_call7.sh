cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for v in 4 2 1; do echo "== VEC $v"; UOC_W4_VEC=$v WINO4_BENCH_ONLY=auto timeout 200 python scripts/wino4_bench.py 2>&1 | grep -E "^s|auto"; done > gpurun_out/r3g_vec.log 2>&1
cat gpurun_out/r3g_vec.log
for v in 4 2 1 4 2; do UOC_W4_VEC=$v timeout 200 python bench.py --steps 48 --cpu-frames 0 --sustained-seconds 6 --skip-pcie --profile-steps 0 > gpurun_out/r3g_bench_v$v.json 2> gpurun_out/r3g_bench_v$v.err; python - <<PY
import json
d=json.load(open("gpurun_out/r3g_bench_v$v.json")); print("vec$v", d["value"], d["sustained"]["frames_per_s"], d["latency"]["frames_per_s"])
PY
done
UOC_W4_VEC=2 timeout 300 python -m pytest tests/test_backbone_gpu.py -q 2>&1 | tail -2
