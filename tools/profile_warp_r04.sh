#!/bin/bash
# round 4: rocprofv3 evidence for the three LDS-staged warpAffine kernels after the rotated frame loop (run on the GPU box from the
# repo root).  Same passes as profile_ops.sh; writes gpurun_out/prof_ops/<tag>.txt (copied into profiles/r04_op_<tag>.txt).
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
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
    name=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
  done
  python $REPO/tools/summarize_op_prof.py $OUT "$KSUB" $ALG "$PAT" > $OUTROOT/$TAG.txt 2>&1
  cat $OUTROOT/$TAG.txt
}
PX8K=$((32*4320*7680))
run_op warp_8k "warpAffine_bilinear_(rot_7deg)_@_8K_batch" "k_warp_affine_lds<3" $((PX8K*6))
run_op warp_gray_8k "warpAffine_bilinear_(rot_7deg)_on_a_GRAY" "k_warp_gray_lds4" $((PX8K*2))
run_op warp_f32_8k "warpAffine_bilinear_f32" "k_warp_f32_lds" $((8*4320*7680*8))
