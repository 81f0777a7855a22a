#!/bin/bash
# Round 4's soak on the final library: round 3's runs (tools/soak_round3.sh) with the scheduling of the chain kernels as
# shipped (sorted class lists, claimed shares, classes side by side), with every piece of it off, with the short-step
# routes off and with the mid rows on.  One summary line per run; gpurun_out/soak_r04.log keeps them.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
B=${1:-0}                 # added to every seed: a second call with another value is another soak
L=gpurun_out/soak_r04.log; [ "$B" = 0 ] && : > $L
sha256sum htscodecs_amd/librans4x16_hip.so | cut -c1-16 | sed "s/^/library /" | tee -a $L
OFF="R4X16_SCHED_SORT=0 R4X16_SCHED_CLAIM=0 R4X16_SCHED_CONCURRENT=0"
for seed in $((401+B)) $((402+B)) $((403+B)) $((404+B)) $((405+B)) $((406+B)); do
  python3 tests/soak/fuzz_damaged_gpu.py 1200 10 $seed 2>&1 | tail -1 | sed "s/^/damaged seed $seed: /" | tee -a $L
done
for seed in $((411+B)) $((412+B)); do
  env $OFF python3 tests/soak/fuzz_damaged_gpu.py 1200 10 $seed 2>&1 | tail -1 | sed "s/^/damaged (scheduling off) seed $seed: /" | tee -a $L
done
for seed in $((421+B)) $((422+B)); do
  R4X16_DEC_DIRECT=0 R4X16_ENC_DIRECT=0 python3 tests/soak/fuzz_damaged_gpu.py 1200 10 $seed 2>&1 | tail -1 | sed "s/^/damaged (compressed rows) seed $seed: /" | tee -a $L
done
for seed in $((431+B)) $((432+B)); do
  R4X16_DEC_MID=1 python3 tests/soak/fuzz_damaged_gpu.py 1200 10 $seed 2>&1 | tail -1 | sed "s/^/damaged (mid rows on) seed $seed: /" | tee -a $L
done
for seed in $((501+B)) $((502+B)) $((503+B)) $((504+B)); do
  python3 tests/soak/fuzz_gpu.py 8000 $seed 2>&1 | tail -1 | sed "s/^/random seed $seed: /" | tee -a $L
done
for seed in $((511+B)) $((512+B)); do
  env $OFF python3 tests/soak/fuzz_gpu.py 8000 $seed 2>&1 | tail -1 | sed "s/^/random (scheduling off) seed $seed: /" | tee -a $L
done
R4X16_DEC_MID=1 python3 tests/soak/fuzz_gpu.py 8000 $((521+B)) 2>&1 | tail -1 | sed "s/^/random (mid rows on) seed $((521+B)): /" | tee -a $L
python3 tests/soak/fuzz_4x8_gpu.py 10 $((78+B)) 2>&1 | tail -2 | tee -a $L
python3 tests/soak/soak_host_batch.py 60000 65536 9 2>&1 | tail -6 | tee -a $L
python3 tests/soak/stripe_batch_rate.py 2>&1 | tail -2 | tee -a $L
python3 tests/soak/big_block_check.py 2>&1 | tail -3 | tee -a $L
