#!/bin/bash
# sweep.py on the shapes the side streams matter for, with the fork threshold at 8 and 16 blocks per CU
cd ${GRAFT_REPO_ROOT:-.}
for per in 8 16; do
  echo "== R4X16_FORK_PER_CU=$per"
  for sh in "q4 193 1048576 4096" "q8 65 1048576 4096" "mixed 1 65536 4096" "q40+dir 1 1048576 4096" "q8 1 1048576 4096" "mixed 1 1048576 2048"; do
    set -- $sh
    R4X16_FORK_PER_CU=$per DATA=$1 ORDER=$2 BS=$3 python3 tools/sweep.py $4 2>&1 | grep nblk
  done
done
