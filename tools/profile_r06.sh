#!/bin/bash
# Round-6 rocprofv3 evidence for bench.py's line.  Run on the GPU box from the repo root:  bash tools/profile_r06.sh <tag>
#   (1) onelaunch: --kernel-trace --stats of the bench command with RCV_FR_SPLIT=0 (every call ONE launch: a kernel's duration is the launch time; the
#       headline kernel and the dominant kernel of every other_configs record);
#   (2) split: the same in the library's default form (a 64-frame filter2D call = two 32-frame launches on the context's two streams, never joined per
#       call): tools/summarize_r06.py reads the overlap of the two queues and the time per CALL from the start / end timestamps;
#   (3) two PMC passes of the one-launch command (separate runs, --kernel-trace only beside --pmc): FETCH_SIZE, WRITE_SIZE per launch -> pmc_traffic.json.
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu --no-verify --no-ceiling --no-probe --sustained 100"
RCV_FR_SPLIT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/onelaunch -o t -- $B > $OUT/onelaunch.json 2> $OUT/onelaunch.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/split -o t -- $B --no-others > $OUT/split.json 2> $OUT/split.log
for c in FETCH_SIZE WRITE_SIZE; do
  RCV_FR_SPLIT=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $B --sustained 40 > $OUT/pmc_$c.log 2>&1
done
cd $REPO
python tools/summarize_r06.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
