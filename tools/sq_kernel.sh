#!/bin/bash
# SQ counters of one kernel on one shape of tools/sweep.py:
#   gpurun -- 'bash tools/sq_kernel.sh <kernel substring> <data> <order> <block size> <blocks>'  -> gpurun_out/sqk/summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/sqk; rm -rf $O; mkdir -p $O
K=$1; export DATA=$2 ORDER=$3 BS=$4; N=$5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    -d $O/p1 --output-format csv -- python3 $R/tools/sweep.py $N > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    -d $O/p2 --output-format csv -- python3 $R/tools/sweep.py $N > $O/p2.log 2>&1
cd $R
python3 - <<PY > $O/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "$K" not in k: continue
        agg[(k[:44], r["Dispatch_Id"], f.split("/")[-3])][r["Counter_Name"]] += float(r["Counter_Value"])
best = {}
for (key, did, p), c in agg.items():
    score = sum(c.values())
    if (key, p) not in best or score >= best[(key, p)][0]: best[(key, p)] = (score, did, c)
out = collections.defaultdict(dict)
for (key, p), (_, did, c) in best.items(): out[key].update(c)
for key, c in sorted(out.items()):
    print(key)
    for n, v in sorted(c.items()): print("   %-24s %16.0f" % (n, v))
PY
cat $O/summary.txt
