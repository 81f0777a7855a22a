#!/bin/bash
# Closing sequence of round 4, from the container: the full GPU suite, then the two refresh calls, then the summaries.
# Every step must succeed (pipefail: a failing stage of a pipeline fails the script - ADVICE r3).
set -e
set -o pipefail
cd "$(dirname "$0")/.."
# (exit code 3 = no box or slot free right now, nothing ran and nothing was charged: wait and ask again - a step that
#  RAN and failed is never repeated)
G() { local i; for i in 1 2 3 4 5 6; do /usr/local/graft/bin/gpurun "$@" && return 0; [ $? -eq 3 ] || return 1; sleep 150; done; return 1; }
if [ "${1:-all}" = all ]; then
rm -rf gpurun_out/refresh
G --timeout 1100 -- 'timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04_final_tests.txt 2>&1; tail -n 3 gpurun_out/r04_final_tests.txt' | tail -n 6
grep -q " passed" gpurun_out/r04_final_tests.txt
fi   # (any argument: the refresh alone)
G --timeout 1150 -- 'timeout -k 10 1100 bash tools/refresh_profiles.sh a' | tail -n 4
G --timeout 1150 -- 'timeout -k 10 1100 bash tools/refresh_profiles.sh b' | tail -n 12
make -C htscodecs_amd/csrc asm > /dev/null
python3 tools/isa_count.py --json profiles/r04_isa_counts.json > /dev/null
python3 tools/make_profiles.py r04 > /dev/null
ls -la profiles | grep r04
