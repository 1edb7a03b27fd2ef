#!/bin/bash
# Round-6 rocprofv3 evidence for bench.py's line.  Run on the GPU box from the repo root:  bash tools/profile_r06.sh <tag>
#   (1) --kernel-trace --stats of the DEFAULT bench command (two batches in flight; other_configs 3s / 3f / 4 / 5 in the same process):
#       kernel durations + start / end timestamps (tools/summarize_r05.py: overlap of the two streams, duration of every dominant kernel);
#   (2) the same with --in-flight 1 (one stream: the per-kernel duration IS the launch time);
#   (3) two PMC passes of the one-stream command (separate runs, --kernel-trace only beside --pmc): FETCH_SIZE, WRITE_SIZE of the dominant
#       kernel of the headline and of every other_configs record -> pmc_traffic.json.
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu --no-verify --no-ceiling --no-probe --sustained 100"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/inflight2 -o t -- $B > $OUT/inflight2.json 2> $OUT/inflight2.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/inflight1 -o t -- $B --in-flight 1 > $OUT/inflight1.json 2> $OUT/inflight1.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $B --in-flight 1 --sustained 40 > $OUT/pmc_$c.log 2>&1
done
cd $REPO
python tools/summarize_r05.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
