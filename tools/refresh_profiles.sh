#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the rocprofv3 kernel statistics of the same command and
# the two PMC passes for HBM traffic; everything lands under gpurun_out/refresh/.  Post-process with
# tools/summarize_trace.py and tools/pmc_traffic.py, then copy the summaries into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/refresh   # (delete the local copy first: gpurun merges new files into it)
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --no-cpu > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $O/write.log 2>&1
tail -n 1 $O/bench.json
