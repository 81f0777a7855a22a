#!/bin/bash
# the one-wave run-length expansion under the soaks (small batches take the workgroup route unless told otherwise)
cd ${GRAFT_REPO_ROOT:-.}
export R4X16_BACK_WG_PER_CU=0
L=gpurun_out/soak_rle_wave.log; : > $L
for seed in 401 402 403 404; do
  python3 tests/soak/fuzz_damaged_gpu.py 1200 10 $seed 2>&1 | tail -1 | sed "s/^/damaged, one-wave expansion, seed $seed: /" | tee -a $L
done
for seed in 501 502; do
  python3 tests/soak/fuzz_gpu.py 8000 $seed 2>&1 | tail -1 | sed "s/^/random, one-wave expansion, seed $seed: /" | tee -a $L
done
