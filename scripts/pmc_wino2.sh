#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/$1; mkdir -p $O
run() { n=$1; shift; timeout 70 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python $R/scripts/wino_one.py > $O/$n.log 2>&1; echo "$n rc=$?"; }
run E TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ
run F TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TA_BUSY TA_FLAT_READ_LDS_WAVEFRONTS
find $O -name "*counter_collection.csv" | head
