#!/bin/bash
# rocprofv3 PMC passes for one op of tools/bench_ops.py:  bash tools/profile_op.sh <tag> <--only pattern>
set -u
TAG=$1; PAT=$2
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/bench_ops.py --steps 5 --warmup 2 --only $PAT --out $OUT/bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
