#!/bin/bash
# MFMA-utilisation counters for the matrix kernels: wino4_gemm_kernel<*>, conv_glds_kernel<*>,
# hc_iter_reg1_kernel, assign_kernel.  Each pass is its own rocprofv3 run with --kernel-trace only (no other trace
# domain beside --pmc), over the bench's launch shapes one launch set at a time on one stream.
#   usage (through gpurun): scripts/pmc_mfma.sh <tag>   -> gpurun_out/pmc_<tag>/{A,B,C,D}/..., pmc_mfma.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export UOC_CONV_TUNE_CACHE=/tmp/uoc_tune_pmc_$1.txt
QUIET="--cpu-frames 0 --sustained-seconds 0 --skip-pcie --skip-latency --profile-steps 0"
timeout 200 python $R/bench.py --steps 4 --warmup 1 --inflight 1 $QUIET > /dev/null 2>&1     # fills the tune cache un-profiled
run() { n=$1; shift; timeout 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python $R/bench.py --steps 8 --warmup 2 --inflight 1 $QUIET > $O/$n.log 2>&1; echo "$n rc=$?"; }
run A SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run B SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run C SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run D SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE
cd $R
python scripts/pmc_summary.py $O wino4_gemm_kernel conv_glds_kernel conv_mfma_kernel hc_iter_reg1_kernel assign_kernel stem > $O/pmc_mfma_raw.md
python scripts/pmc_mfma_table.py $O > $O/pmc_mfma.md
find $O -name "*counter_collection.csv" -size +30M -delete; find $O -name "*.db" -size +40M -delete
head -60 $O/pmc_mfma.md
