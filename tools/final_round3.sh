#!/bin/bash
# Round 3's closing run on the GPU box: the profile refresh, the host-buffer rates, the GPU suite.
#   gpurun --timeout 1200 -- 'bash tools/final_round3.sh'   then, in the container:  python3 tools/make_profiles.py r03
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
bash tools/refresh_profiles.sh > gpurun_out/final_refresh.log 2>&1 || { tail -5 gpurun_out/final_refresh.log; exit 1; }
tail -n 1 gpurun_out/refresh/bench.json | cut -c1-300
H=gpurun_out/refresh/host_batch_rate.txt
python3 tools/threads_single_call.py 2>&1 | tail -1 > $H
python3 tools/latency_single_call.py 2>&1 | tail -3 >> $H
bash tools/host_batch_rate.sh 3 1 2>&1 | grep "MB/s enc" >> $H
cat $H
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
