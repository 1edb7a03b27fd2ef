#!/bin/bash
# rocprofv3 counters of the one-channel warpAffine kernels (run on the GPU box from the repo root): the four-frames-per-pass kernel
# and, with RCV_WARP_GRAY4=0, the one-frame kernel it replaces.  Same passes as profile_ops_r03.sh plus the LDS counters.
set -u
REPO=$PWD
OUTROOT=$REPO/gpurun_out/prof_ops
mkdir -p $OUTROOT
cd /tmp && export TMPDIR=/tmp
run_op() {  # tag, --only pattern, kernel substring, algorithmic bytes per launch
  local TAG=$1 PAT=$2 KSUB=$3 ALG=$4
  local OUT=/tmp/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
  local CMD="python $REPO/tools/bench_ops.py --steps 5 --warmup 2 --only $PAT --out $OUT/bench.json"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LEVEL_WAVES"; do
    name=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
  done
  python $REPO/tools/summarize_op_prof.py $OUT "$KSUB" $ALG "$PAT" > $OUTROOT/$TAG.txt 2>&1
  cat $OUTROOT/$TAG.txt
}
PX8K=$((32*4320*7680))
run_op warp_gray4 "warpAffine_bilinear_(rot_7deg)_on_a_GRAY" "k_warp_gray_lds4" $((PX8K*2))
RCV_WARP_GRAY4=0 run_op warp_gray1 "warpAffine_bilinear_(rot_7deg)_on_a_GRAY" "k_warp_affine_lds<1" $((PX8K*2))
