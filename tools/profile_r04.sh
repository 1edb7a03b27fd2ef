#!/bin/bash
# Round-4 rocprofv3 evidence for bench.py's line.  Run on the GPU box from the repo root:  bash tools/profile_r04.sh <tag>
#   (1) --kernel-trace --stats of the DEFAULT bench command (two batches in flight): kernel durations + start / end timestamps, from
#       which tools/summarize_r04.py shows the overlap and reconciles bytes / wall time with the per-kernel durations;
#   (2) the same with --in-flight 1 (one stream: the per-kernel duration IS the launch time);
#   (3) PMC passes (separate runs, never with a trace domain beyond --kernel-trace): FETCH_SIZE / WRITE_SIZE of the dominant kernel of
#       configs 3, 4 and 5 -> pmc_traffic.json.
set -u
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu --no-verify --no-ceiling --no-probe --no-others --sustained 100"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/inflight2 -o t -- $B > $OUT/inflight2.json 2> $OUT/inflight2.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/inflight1 -o t -- $B --in-flight 1 > $OUT/inflight1.json 2> $OUT/inflight1.log
for cfg in 3 4 5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_c${cfg}_$c -o pmc -- $B --config $cfg --in-flight 1 --sustained 40 > $OUT/pmc_c${cfg}_$c.log 2>&1
  done
done
cd $REPO
python tools/summarize_r04.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
