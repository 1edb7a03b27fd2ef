#!/bin/bash
# rocprofv3 evidence for the north-star kernel.  Run on the GPU box from the repo root:
#   bash tools/profile_filter7.sh <tag>
# Writes CSV summaries under gpurun_out/prof_<tag>/ (copy the ones to keep into profiles/).
# Kernel-trace/stats and each PMC set are SEPARATE runs (never --pmc with a trace domain).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu --no-verify --no-ceiling --sustained 100"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python tools/summarize_prof.py $OUT $OUT/pmc_traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
