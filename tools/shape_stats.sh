#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE shapes (tools/sweep.py under --kernel-trace --stats), one directory
# per shape under gpurun_out/shape_stats/.  Afterwards, in the container: python3 tools/shape_stats_summary.py <tag>.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/shape_stats
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
prof() { # name data order bs nblk
  DATA=$2 ORDER=$3 BS=$4 rocprofv3 --kernel-trace --stats -d $O/$1 --output-format csv -- python3 $R/tools/sweep.py $5 > $O/$1.log 2>&1
  find $O/$1 -name "*agent_info.csv" -delete; find $O/$1 -name "*domain_stats.csv" -delete
}
prof o0_q40_1MiB q40+dir 0 1048576 15360
prof o1_q8_1MiB q8 1 1048576 8192
prof o193_q4_1MiB q4 193 1048576 4096
prof o1_mixed_64KiB mixed 1 65536 32768
prof o1_q40_64KiB q40+dir 1 65536 61440
grep -h nblk $O/*.log
