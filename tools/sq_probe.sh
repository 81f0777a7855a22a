#!/bin/bash
# SQ counters of the chain kernels on one small batch (default: ONE 1 MiB q40 block): where a lone wave's cycles go.
#   gpurun -- 'bash tools/sq_probe.sh [blocks] [block size] [data] [order]'   -> gpurun_out/sqp/summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/sqp; rm -rf $O; mkdir -p $O
A="${1:-1} ${2:-1048576} ${3:-q40+dir} ${4:-1}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY \
    -d $O/p1 --output-format csv -- python3 $R/tools/trace_probe.py $A > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    -d $O/p2 --output-format csv -- python3 $R/tools/trace_probe.py $A > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_EXP_GDS \
    -d $O/p3 --output-format csv -- python3 $R/tools/trace_probe.py $A > $O/p3.log 2>&1 || true
cd $R
python3 - <<PY > $O/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "chain" not in k: continue
        key = k[:44]
        agg[(key, r["Dispatch_Id"], f.split("/")[-3])][r["Counter_Name"]] += float(r["Counter_Value"])
# keep, per kernel name and pass, the dispatch with the most wave cycles / instructions (the working launch of the LAST pass)
best = {}
for (key, did, p), c in agg.items():
    score = sum(c.values())
    if (key, p) not in best or score >= best[(key, p)][0]: best[(key, p)] = (score, did, c)
out = collections.defaultdict(dict)
for (key, p), (_, did, c) in best.items(): out[key].update(c)
for key, c in sorted(out.items()):
    if c.get("SQ_INSTS_VALU", 0) < 1e6: continue
    print(key)
    for n, v in sorted(c.items()): print("   %-24s %16.0f" % (n, v))
PY
cat $O/summary.txt
