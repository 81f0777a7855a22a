#!/bin/bash
# Closing sequence of round 4, from the container: the full GPU suite, then the two refresh calls, then the summaries.
# Every step must succeed (pipefail: a failing stage of a pipeline fails the script - ADVICE r3).
set -e
set -o pipefail
cd "$(dirname "$0")/.."
rm -rf gpurun_out/refresh
G=/usr/local/graft/bin/gpurun
$G --timeout 1100 -- 'timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04_final_tests.txt 2>&1; tail -n 3 gpurun_out/r04_final_tests.txt' | tail -n 6
grep -q " passed" gpurun_out/r04_final_tests.txt
$G --timeout 1150 -- 'timeout -k 10 1100 bash tools/refresh_profiles.sh a' | tail -n 4
$G --timeout 1150 -- 'timeout -k 10 1100 bash tools/refresh_profiles.sh b' | tail -n 12
make -C htscodecs_amd/csrc asm > /dev/null
python3 tools/isa_count.py --json profiles/r04_isa_counts.json > /dev/null
python3 tools/make_profiles.py r04 > /dev/null
ls -la profiles | grep r04
