cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
UOC_PARITY_E2E_FRAMES=1024 timeout 900 python -m pytest tests/test_headline_parity_gpu.py -q -s -k histogram > gpurun_out/r3m_hist.log 2>&1; echo "hist rc=$?"; grep -E "^\{|passed|failed" gpurun_out/r3m_hist.log | cut -c1-400
cp gpurun_out/parity_histogram.json gpurun_out/r3m_parity_histogram_1024.json
timeout 600 python bench.py --frames 1024 --cpu-frames 0 --sustained-seconds 0 --skip-pcie --skip-latency --profile-steps 0 > gpurun_out/r3m_strong1024.json 2> gpurun_out/r3m_strong1024.err; echo "strong rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3m_strong1024.json')); print(d['value'], d['ms_per_step'], d['config']['total_frames'], d['per_rank'])"
