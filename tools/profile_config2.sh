#!/bin/bash
# rocprofv3 kernel durations of BASELINE config 2 (one 1080p BGR frame, 5x5 integer Gaussian):  bash tools/profile_config2.sh
set -u
REPO=$PWD; OUT=$REPO/gpurun_out/prof_config2; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python $REPO/tools/config2_latency.py --quick > $OUT/run.log 2>&1
cd $REPO
cat $OUT/run.log | grep -v "^W2" | tail -12
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_config2/stats/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if any(k in row["Name"] for k in ("k_gauss_rows", "k_filter7_mfma", "k_nop", "k_filter_rows")):
            print(f"{row['Name'][:70]:70s} calls={row['Calls']:>6s} avg={float(row['AverageNs'])/1e3:6.2f} us  min={float(row['MinNs'])/1e3:6.2f}  max={float(row['MaxNs'])/1e3:6.2f}")
PY
