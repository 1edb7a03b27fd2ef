#!/bin/bash
# rocprofv3 PMC passes of the fused warp -> down-scale launch through tools/ablate_warp_resize.py:  bash tools/profile_wr.sh <tag> "<plan>"
set -u
TAG=$1; PLAN=$2; KSUB=${3:-k_warp_resize}
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/ablate_warp_resize.py --rot 1 --launches 10 --plans $PLAN"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python tools/summarize_op_prof.py $OUT "$KSUB" 1990656000 "fused warp -> 4x resize, plan $PLAN" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
