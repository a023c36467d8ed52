cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for tg in 0 2 3 4; do echo "== TGRID $tg"; UOC_W4_TGRID=$tg WINO4_BENCH_ONLY=auto timeout 200 python scripts/wino4_bench.py 2>&1 | grep -E "^s|auto"; done > gpurun_out/r3f_tgrid.log 2>&1
cat gpurun_out/r3f_tgrid.log
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_bench_dist_gpu.py -x -q > gpurun_out/r3f_pipe.log 2>&1; echo "pipe rc=$?"; tail -3 gpurun_out/r3f_pipe.log
for tg in 0 2; do UOC_W4_TGRID=$tg timeout 200 python bench.py --steps 48 --cpu-frames 0 --sustained-seconds 6 --skip-pcie --profile-steps 0 > gpurun_out/r3f_bench_tg$tg.json 2> gpurun_out/r3f_bench_tg$tg.err; echo "bench tg$tg rc=$?"; done
python - <<'PY'
import json
for n in ("tg0","tg2"):
    d=json.load(open(f"gpurun_out/r3f_bench_{n}.json")); print(n, d["value"], d["sustained"]["frames_per_s"], d["latency"])
PY
