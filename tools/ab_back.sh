#!/bin/bash
# k_dec_back of every library under htscodecs_amd/variants against the built one: tools/ab_back.sh [blocks [wg_per_cu]]
cd ${GRAFT_REPO_ROOT:-.}
NB=${1:-4096}; WG=${2:-0}

for v in base $(ls htscodecs_amd/variants 2>/dev/null | sed "s/^lib//; s/\.so$//"); do
  if [ $v = base ]; then export R4X16_LIB=$PWD/htscodecs_amd/librans4x16_hip.so; else export R4X16_LIB=$PWD/htscodecs_amd/variants/lib$v.so; fi   # (htscodecs_amd/lib.py: the shipped library is never replaced)
  echo "== $v"
  R4X16_BACK_WG_PER_CU=$WG bash tools/shape_one.sh ab_$v q8 65 1048576 $NB | grep -E "nblk|k_dec_back"
done

