#!/bin/bash
# rocprofv3 evidence for the other ops of the path (run on the GPU box from the repo root):  bash tools/profile_ops.sh
# One --kernel-trace --stats run and separate --pmc passes per op (never a --pmc together with a trace domain other than
# --kernel-trace), on `tools/bench_ops.py --only <op>`; writes gpurun_out/prof_ops/<tag>.txt (copy into profiles/).
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
PX4K=$((64*2160*3840)); PX8K=$((32*4320*7680))
run_op bgr2gray_4k "BGR2GRAY" "k_bgr2gray16" $((PX4K*4))
run_op sobel_4k "Sobel_3x3_->_dx,dy_i16_@_4K" "k_sobel_rows<0, false" $((PX4K*5))
run_op harris_4k "Harris_pipeline_(BGR->mask)_@_4K" "k_harris_fused<false, 0," $((PX4K*4))
run_op harris_b3_4k "Harris_pipeline_blockSize_3" "k_harris_blocks_fused" $((PX4K*4))
run_op warp_8k "warpAffine_bilinear_(rot_7deg)" "k_warp_affine_lds<3" $((PX8K*6))
PXO=$((32*1080*1920))
run_op warp_resize_fused "warpAffine_+_resize" "k_warp_resize_box" $((PXO*30))
run_op resize_5k "resize_8K_->_5K" "k_resize_bgr" $((32*2880*5120*975/100))
run_op filter7_gray_4k "filter2D_7x7_i8_on_a_GRAY" "k_filter_rows_mfma<7, 3, 0, 0, 2>" $((PX4K*2))
run_op nms_4k "NMS_3x3" "k_nms3x3_rows" $((PX4K*5))
run_op filter_sobel_4k "filter2D_7x7_i8_->_gray_->_Sobel" "k_filter_rows_mfma<7, 3, 0, 0, 0, 1>" $((PX4K*7))
