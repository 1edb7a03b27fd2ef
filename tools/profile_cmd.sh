#!/bin/bash
# rocprofv3 kernel stats + PMC passes of ANY command (run on the GPU box from the repo root):
#   bash tools/profile_cmd.sh <tag> "<kernel substring>" <algorithmic bytes per launch> "<title>" -- <command ...>
# (tools/profile_bench_config.sh is this for `bench.py --config N`); writes gpurun_out/prof_<tag>/summary.txt
set -u
TAG=$1; KSUB=$2; ALG=$3; TITLE=$4; shift 5
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p $OUT; REPO=$PWD
export PYTHONPATH=$REPO
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- "$@" > $OUT/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- "$@" > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python tools/summarize_op_prof.py $OUT "$KSUB" $ALG "$TITLE" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
