#!/bin/bash
# one shape of tools/sweep.py on the built library and on every library under htscodecs_amd/variants, alternating:
#   tools/ab_shape.sh <data> <order> <block size> <blocks> [repeats]
cd ${GRAFT_REPO_ROOT:-.}
cp htscodecs_amd/librans4x16_hip.so /tmp/base.so
for rep in $(seq 1 ${5:-2}); do
for v in base $(ls htscodecs_amd/variants 2>/dev/null | sed "s/^lib//; s/\.so$//"); do
  if [ $v = base ]; then cp /tmp/base.so htscodecs_amd/librans4x16_hip.so; else cp htscodecs_amd/variants/lib$v.so htscodecs_amd/librans4x16_hip.so; fi
  echo -n "$v: "; DATA=$1 ORDER=$2 BS=$3 python3 tools/sweep.py $4 2>&1 | grep nblk
done
done
cp /tmp/base.so htscodecs_amd/librans4x16_hip.so
