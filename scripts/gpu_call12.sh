cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
for t in 24 20 12 30; do UOC_HC_VB_TILES=$t timeout 200 python bench.py --steps 16 --cpu-frames 0 --sustained-seconds 0 --skip-pcie --skip-latency > /tmp/b.json 2> /tmp/b.err; python - <<PY
import json
d=json.load(open("/tmp/b.json")); print("tiles$t", d["value"])
for x in d["clustering_by_shape"]:
    if x["kernel"]=="hc_iter": print("   ", x)
PY
done
