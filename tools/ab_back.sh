#!/bin/bash
# k_dec_back of every library under htscodecs_amd/variants against the built one: tools/ab_back.sh [blocks [wg_per_cu]]
cd ${GRAFT_REPO_ROOT:-.}
NB=${1:-4096}; WG=${2:-0}
cp htscodecs_amd/librans4x16_hip.so /tmp/base.so
for v in base $(ls htscodecs_amd/variants 2>/dev/null | sed "s/^lib//; s/\.so$//"); do
  if [ $v = base ]; then cp /tmp/base.so htscodecs_amd/librans4x16_hip.so; else cp htscodecs_amd/variants/lib$v.so htscodecs_amd/librans4x16_hip.so; fi
  echo "== $v"
  R4X16_BACK_WG_PER_CU=$WG bash tools/shape_one.sh ab_$v q8 65 1048576 $NB | grep -E "nblk|k_dec_back"
done
cp /tmp/base.so htscodecs_amd/librans4x16_hip.so
