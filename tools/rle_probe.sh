#!/bin/bash
# the run-length shapes at several batch sizes, workgroup-per-block expansion forced on (99999) and off (0)
cd ${GRAFT_REPO_ROOT:-.}
for wg in 0 99999; do
  echo "== R4X16_BACK_WG_PER_CU=$wg"
  for sh in "q8 65 1048576 64" "q8 65 1048576 1024" "q8 65 1048576 2048" "q8 65 1048576 4096" "q4 193 1048576 1024" "q4 193 1048576 4096"; do
    set -- $sh
    R4X16_BACK_WG_PER_CU=$wg DATA=$1 ORDER=$2 BS=$3 python3 tools/sweep.py $4 2>&1 | grep nblk
  done
done
