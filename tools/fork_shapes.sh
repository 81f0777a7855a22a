#!/bin/bash
# chain kernels side by side (R4X16_FORK_PER_CU) on shapes with two stream kinds per block, with one kind, and on the
# batches whose direct-row workgroups take all of a CU's LDS
cd ${GRAFT_REPO_ROOT:-.}
for f in ${FORKS:-2 16}; do
  echo "== R4X16_FORK_PER_CU=$f"
  IFS=";" read -ra SH <<< "${SHAPES:-q40+dir 1 1048576 1024;q40+dir 1 1048576 2048;q40+dir 1 1048576 3072;mixed 1 65536 4096;mixed 1 65536 2048;q8 1 1048576 2048}"
  for sh in "${SH[@]}"; do
    set -- $sh
    R4X16_FORK_PER_CU=$f DATA=$1 ORDER=$2 BS=$3 python3 tools/sweep.py $4 2>&1 | grep nblk
  done
done
