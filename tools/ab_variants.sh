#!/bin/bash
# A/B of library builds on one GPU box (box-to-box variation is larger than most effects worth measuring).
# Build the variants first, in the container, e.g. for compiler scheduling strategies:
#   cd htscodecs_amd/csrc && mkdir -p ../variants && for v in "A:-mllvm -enable-post-misched=false" \
#     "B:-mllvm -amdgpu-sched-strategy=max-ilp"; do n=${v%%:*}; f=${v#*:}; hipcc --offload-arch=gfx950 -O3 -std=c++17 \
#     -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math $f -shared -Wl,--version-script=exports.map \
#     -o ../variants/lib$n.so r4x16_*.hip; done
# (htscodecs_amd/variants/*.so is git-ignored and travels with gpurun), then: gpurun -- 'bash tools/ab_variants.sh'
cd $GRAFT_REPO_ROOT
cp htscodecs_amd/librans4x16_hip.so /tmp/base.so
for v in base $(ls htscodecs_amd/variants 2>/dev/null | sed "s/^lib//; s/\.so$//"); do
  if [ $v = base ]; then cp /tmp/base.so htscodecs_amd/librans4x16_hip.so; else cp htscodecs_amd/variants/lib$v.so htscodecs_amd/librans4x16_hip.so; fi
  echo "== $v"
  BS=1048576 python3 tools/sweep.py 23040 2>&1 | grep nblk | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('O1 q40  enc', d['enc_chain_ms'], 'dec', d['dec_chain_ms'], 'step', d['step_ms'], d['ok'])"
  BS=1048576 ORDER=0 python3 tools/sweep.py 15360 2>&1 | grep nblk | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('O0 q40  enc', d['enc_chain_ms'], 'dec', d['dec_chain_ms'], 'step', d['step_ms'], d['ok'])"
done
cp /tmp/base.so htscodecs_amd/librans4x16_hip.so
