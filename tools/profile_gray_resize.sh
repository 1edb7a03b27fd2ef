#!/bin/bash
# counters of the one-channel u8 resize (run on the GPU box from the repo root)
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg; rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d /tmp/pg -o pmc -- python $REPO/tools/bench_geom_misc.py > /tmp/pg.log 2>&1
rm -rf /tmp/ps; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o st -- python $REPO/tools/bench_geom_misc.py > /tmp/ps.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pg/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "resize" in k or "warp" in k:
        print(k, {c: "%.3g" % (sum(x)/len(x)) for c,x in v.items()}, len(list(v.values())[0]))
f=glob.glob("/tmp/ps/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if "resize" in r["Name"] or "warp" in r["Name"]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
