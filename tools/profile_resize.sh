#!/bin/bash
# rocprofv3 counters of the general (non-box) resize kernels, u8 BGR and one-channel f32 (run on the GPU box from the repo root)
set -u
REPO=$PWD
OUTROOT=$REPO/gpurun_out/prof_ops
mkdir -p $OUTROOT
cd /tmp && export TMPDIR=/tmp
run_op() {
  local TAG=$1 PAT=$2 KSUB=$3 ALG=$4
  local OUT=/tmp/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
  local CMD="python $REPO/tools/bench_ops.py --steps 5 --warmup 2 --only $PAT --out $OUT/bench.json"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
    name=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
  done
  python $REPO/tools/summarize_op_prof.py $OUT "$KSUB" $ALG "$PAT" > $OUTROOT/$TAG.txt 2>&1
  cat $OUTROOT/$TAG.txt
}
run_op resize_5k_u8 "resize_8K_->_5K_bilinear_(general_1.5x)_@_8K_batch=32" "k_resize" $((32*5120*2880*39/4))
run_op resize_5k_f32 "resize_f32_8K_->_5K" "k_resize_f32" $((8*5120*2880*13))
