#!/bin/bash
# hill-climbing kernel study on the GPU box: ablation timings + SQ counters of the shipped kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hc_study; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for a in 0 1 2 3 4 5; do echo "ablate $a: $(UOC_HC_ABLATE=$a timeout 120 python $R/scripts/time_meanshift.py 2>&1 | grep hc_ms)"; done | tee $O/ablate.txt
echo "lds-kernel: $(UOC_HC_VARIANT=0 timeout 120 python $R/scripts/time_meanshift.py 2>&1 | grep hc_ms)" | tee -a $O/ablate.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM"
P3="SQ_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for v in 1 0; do i=0; for P in "$P1" "$P2" "$P3"; do i=$((i+1));
  UOC_HC_VARIANT=$v timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/v${v}_p$i -o p -- python $R/scripts/time_meanshift.py > /dev/null 2> $O/v${v}_p$i.err
done; done
cd $R
for v in 1 0; do python scripts/pmc_summary.py "$O" hc_iter > /dev/null; done
python - <<'PY'
import glob,os,subprocess,sys
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","hc_study")
for v in (1,0):
    out=open(os.path.join(O,f"pmc_v{v}.md"),"w")
    for p in sorted(glob.glob(os.path.join(O,f"v{v}_p*"))):
        if os.path.isdir(p):
            out.write(subprocess.run([sys.executable,"scripts/pmc_summary.py",p,"hc_iter"],capture_output=True,text=True).stdout)
    out.close()
    print(open(os.path.join(O,f"pmc_v{v}.md")).read())
PY
find $O -name "*.csv" -size +1M -delete
