#!/bin/bash
# rocprofv3 SQ counters of the north-star launch on ONE stream: chained-band kernel (default) against the one-band-per-wave
# kernel (RCV_FR_CHAIN=0).  Run on the GPU box from the repo root:  bash tools/profile_chain.sh <tag>
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_chain_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu --no-verify --no-ceiling --no-probe --no-others --in-flight 1 --sustained 60"
for mode in chain plain; do
  if [ $mode = plain ]; then export RCV_FR_CHAIN=0; else unset RCV_FR_CHAIN; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${mode}_stats -o stats -- $CMD > $OUT/${mode}_stats.log 2>&1
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $set | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${mode}_pmc_$name -o pmc -- $CMD > $OUT/${mode}_pmc_$name.log 2>&1
  done
done
cd $REPO
python - <<PY > $OUT/summary.txt 2>&1
import csv, glob, os
from collections import defaultdict
out = "$OUT"
for mode in ("chain", "plain"):
    print("====", mode)
    for f in glob.glob(os.path.join(out, mode + "_stats", "**", "*kernel_stats.csv"), recursive=True):
        for i, row in enumerate(csv.reader(open(f))):
            if i < 4:
                print("  ", ",".join(row))
    for d in sorted(glob.glob(os.path.join(out, mode + "_pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(list)
            for row in csv.DictReader(open(f)):
                if "k_filter_rows" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            for c, v in acc.items():
                print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
cat $OUT/summary.txt
