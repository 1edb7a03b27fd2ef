#!/bin/bash
# rocprofv3 kernel stats + PMC passes of ONE BASELINE config as bench.py runs it (one stream):
#   bash tools/profile_bench_config.sh <tag> <config 3|4|5> <kernel substring> <algorithmic bytes per launch> "<title>"
set -u
TAG=$1; CFG=$2; KSUB=$3; ALG=$4; TITLE=$5
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --config $CFG --in-flight 1 --steps 10 --warmup 3 --no-cpu --no-verify --no-ceiling --no-probe --no-others --sustained 60"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python tools/summarize_op_prof.py $OUT "$KSUB" $ALG "$TITLE" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
