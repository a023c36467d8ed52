#!/bin/bash
# L2 behaviour of the fp32 and the split-precision plane GEMM on one layer shape (rocprofv3 --pmc, kernel trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w3_pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE TCC_REQ_sum TCC_READ_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  SPLIT_BENCH_SHAPES="${1:-s1 layer4 512 d4 4x}" timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -o p -- python $R/scripts/split_bench.py > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "wino4_gemm" not in k: continue
    name = "gemm3" if "gemm3" in k else "gemm_fp32"
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, c in agg.items():
    print(name, {k: round(sum(v) / len(v)) for k, v in c.items()}, "dispatches", len(next(iter(c.values()))))
PY
done
