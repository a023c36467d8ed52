cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
UOC_PARITY_FRAME_LIST=445,845,639,705 timeout 600 python -m pytest tests/test_headline_parity_gpu.py -q -s -k separately > gpurun_out/r3n_outliers.log 2>&1; echo rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/parity_decomposed.json"))
print({k:v for k,v in d.items() if k!="per_frame"})
for r in d["per_frame"]: print(r)
PY
cp gpurun_out/parity_decomposed.json gpurun_out/r3n_outliers.json
