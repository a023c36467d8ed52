cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 400 python scripts/oracle_thread_sensitivity.py 0 1 > gpurun_out/r3e_sens.log 2>&1; echo "sens rc=$?"; cat gpurun_out/r3e_sens.log | tail -3
timeout 600 python -m pytest tests/test_headline_parity_gpu.py -q -s > gpurun_out/r3e_parity.log 2>&1; echo "parity rc=$?"; grep -E "^\{|passed|failed" gpurun_out/r3e_parity.log | cut -c1-600
