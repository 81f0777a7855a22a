#!/bin/bash
# one shape of tools/sweep.py on the built library and on every library under htscodecs_amd/variants, alternating:
#   tools/ab_shape.sh <data> <order> <block size> <blocks> [repeats]
# (variants are selected through R4X16_LIB, htscodecs_amd/lib.py: the shipped library is never replaced)
cd ${GRAFT_REPO_ROOT:-.}
for rep in $(seq 1 ${5:-2}); do
for v in base $(ls htscodecs_amd/variants 2>/dev/null | sed "s/^lib//; s/\.so$//"); do
  if [ $v = base ]; then lib=$PWD/htscodecs_amd/librans4x16_hip.so; else lib=$PWD/htscodecs_amd/variants/lib$v.so; fi
  echo -n "$v: "; R4X16_LIB=$lib DATA=$1 ORDER=$2 BS=$3 python3 tools/sweep.py $4 2>&1 | grep nblk
done
done
