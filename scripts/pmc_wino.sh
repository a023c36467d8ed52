#!/bin/bash
# PMC passes over the Winograd GEMM (dev).  Usage (through gpurun): UOC_WINO_TMT=5 scripts/pmc_wino.sh <outdir>
# Each pass is its own rocprofv3 run with --kernel-trace only.  NOTE: on this image the derived "*_sum" TCP/TCC
# counters and the TA_* counters make rocprofv3 abort (signal 6) or hang until the timeout; the raw ones below work.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/$1; mkdir -p $O
run() { n=$1; shift; timeout 70 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python $R/scripts/wino_one.py > $O/$n.log 2>&1; echo "$n rc=$?"; }
run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run B SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD
run E TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ
find $O -name "*counter_collection.csv" | head
