#!/bin/bash
# rocprofv3 kernel statistics of ONE shape: tools/shape_one.sh <name> <data> <order> <block size> <blocks>
#   -> gpurun_out/shape_one/<name>/ and a per-kernel summary of the working launches on stdout
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/shape_one; mkdir -p "$O"; rm -rf "$O/$1"
cd /tmp && export TMPDIR=/tmp
DATA=$2 ORDER=$3 BS=$4 rocprofv3 --kernel-trace --stats -d $O/$1 --output-format csv -- python3 $R/tools/sweep.py $5 > $O/$1.log 2>&1
find $O/$1 -name "*agent_info.csv" -delete; find $O/$1 -name "*domain_stats.csv" -delete
grep -h nblk $O/$1.log
python3 $R/tools/summarize_trace.py $(find $O/$1 -name "*kernel_trace.csv" | head -1) 100 | head -14
