cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
export UOC_PARITY_FRAMES=8
for v in "A" "B UOC_FPS_PACK=0" "C UOC_HC_VARIANT=0" "D UOC_FPS_PACK=0 UOC_HC_VARIANT=0"; do set -- $v; tag=$1; shift
  env "$@" timeout 300 python -m pytest tests/test_headline_parity_gpu.py -q -s -k separately > gpurun_out/r3d_parity_$tag.log 2>&1; echo "$tag ($*) rc=$?"
  cp gpurun_out/parity_decomposed.json gpurun_out/r3d_parity_$tag.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r3d_parity_$tag.json"))
print({k:v for k,v in d.items() if k!="per_frame"})
for r in d["per_frame"]:
    g=r["given_oracle_embeddings"]
    if not (g["stage1_identical_ids"] and g["final_identical_ids"]): print(r)
PY
done
