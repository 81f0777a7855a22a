#!/bin/bash
# Round 3's long soak on the final library: damaged streams over many seeds (small batches: the direct rows and the
# workgroup run-length expansion; and with the short-step routes off), random inputs x random flag sets, rANS 4x8, the host
# batch pipeline, stripe batches, big blocks.  Prints one summary line per run; gpurun_out/soak_r03.log keeps them.
cd ${GRAFT_REPO_ROOT:-.}
L=gpurun_out/soak_r03.log; : > $L
for seed in 101 102 103 104 105 106 107 108 109 110 111 112; do
  python3 tests/soak/fuzz_damaged_gpu.py 1200 10 $seed 2>&1 | tail -1 | sed "s/^/damaged seed $seed: /" | tee -a $L
done
for seed in 201 202 203 204; do
  R4X16_DEC_DIRECT=0 R4X16_ENC_DIRECT=0 python3 tests/soak/fuzz_damaged_gpu.py 1200 10 $seed 2>&1 | tail -1 | sed "s/^/damaged (compressed rows) seed $seed: /" | tee -a $L
done
for seed in 301 302 303 304 305 306; do
  python3 tests/soak/fuzz_gpu.py 8000 $seed 2>&1 | tail -1 | sed "s/^/random seed $seed: /" | tee -a $L
done
python3 tests/soak/fuzz_4x8_gpu.py 10 77 2>&1 | tail -2 | tee -a $L
python3 tests/soak/soak_host_batch.py 60000 65536 9 2>&1 | tail -6 | tee -a $L
python3 tests/soak/stripe_batch_rate.py 2>&1 | tail -2 | tee -a $L
python3 tests/soak/big_block_check.py 2>&1 | tail -3 | tee -a $L
