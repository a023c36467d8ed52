#!/bin/bash
# Counters behind profiles/r05_ab_gemm_variants.md: the plane GEMM as shipped (8 waves per block), with ONE wave per SIMD
# (UOC_W4_WAVES=4) and with the 256-wide block tile (UOC_WINO4_TILE=128256), development library, on the two layer-4 launch
# shapes of the pipeline.  Separate rocprofv3 passes with --kernel-trace only next to --pmc (MI355X_MICROARCH.md).
#   usage (through gpurun): scripts/pmc_gemm_variants.sh <tag>  -> gpurun_out/pmc_gemm_<tag>/<arm>/{A,B}, <arm>.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export UOC_LIB_PATH=$R/unseenobjectclustering_amd/libuoc_hip_dev.so WINO4_BENCH_SHAPES="layer4 512 d4"
arm() { name=$1; only=$2; shift 2
  for p in A B; do
    if [ $p = A ]; then C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE";
    else C="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; fi
    WINO4_BENCH_ONLY="$only" timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$name/$p -o p -- python $R/scripts/wino4_bench.py > $O/$name.$p.log 2>&1
    echo "$name $p rc=$?"
  done
  (cd $R && python scripts/pmc_mfma_table.py $O/$name > $O/$name.md)
  echo "== $name"; grep wino4_gemm $O/$name.md
}
arm waves8 "8 waves auto"
arm waves4 "4 waves auto"
arm wide "128x256 wide"
find $O -name "*counter_collection.csv" -size +30M -delete; find $O -name "*.db" -size +40M -delete
