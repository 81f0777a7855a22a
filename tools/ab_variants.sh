#!/bin/bash
# A/B of library builds on one GPU box (box-to-box variation is larger than most effects worth measuring).
# Build the variants first, in the container: tools/build_variant.sh <name> [-D... / -mllvm ...]
# (htscodecs_amd/variants/*.so is git-ignored and travels with gpurun), then: gpurun -- 'bash tools/ab_variants.sh'
# The variant is selected through R4X16_LIB, which htscodecs_amd/lib.py honours: the shipped library is never replaced.
cd ${GRAFT_REPO_ROOT:-.}
for rep in $(seq 1 ${1:-1}); do
for v in base $(ls htscodecs_amd/variants 2>/dev/null | sed "s/^lib//; s/\.so$//"); do
  if [ $v = base ]; then lib=$PWD/htscodecs_amd/librans4x16_hip.so; else lib=$PWD/htscodecs_amd/variants/lib$v.so; fi
  echo "== $v"
  R4X16_LIB=$lib BS=1048576 python3 tools/sweep.py 23040 2>&1 | grep nblk | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('O1 q40  enc', d['enc_chain_ms'], 'dec', d['dec_chain_ms'], 'step', d['step_ms'], d['ok'])"
  R4X16_LIB=$lib BS=1048576 ORDER=0 python3 tools/sweep.py 15360 2>&1 | grep nblk | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('O0 q40  enc', d['enc_chain_ms'], 'dec', d['dec_chain_ms'], 'step', d['step_ms'], d['ok'])"
done
done
