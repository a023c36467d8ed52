cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r3k_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^FAILED" gpurun_out/r3k_gpu_tests.log | tail -10
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
