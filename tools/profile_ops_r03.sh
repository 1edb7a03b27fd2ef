#!/bin/bash
# round 3: rocprofv3 evidence for the rows VERDICT r2 asked about (run on the GPU box from the repo root): the two f32 filters
# (VALU roofline), the Sobel / NMS kernels after the store fix, config 2's register-window kernel.  Same passes as profile_ops.sh.
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
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
    name=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
  done
  python $REPO/tools/summarize_op_prof.py $OUT "$KSUB" $ALG "$PAT" > $OUTROOT/$TAG.txt 2>&1
  cat $OUTROOT/$TAG.txt
}
PX4K=$((64*2160*3840))
run_op filter_f32_7 "filter2D_7x7_f32" "k_filter_f32_stream" $((PX4K*6))
run_op gauss_sigma_7 "GaussianBlur_7x7_(sigma=1.5)" "k_filter_f32_stream" $((PX4K*6))
run_op sobel_4k "Sobel_3x3_->_dx,dy_i16_@_4K" "k_sobel_rows<0, false" $((PX4K*5))
run_op nms_4k "NMS_3x3" "k_nms3x3_rows" $((PX4K*5))
