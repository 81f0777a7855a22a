#!/bin/bash
# experiment: streams per wave / waves per CU of the rANS 4x8 chain kernels (library built with -DX8_EXP)
set -o pipefail
export R4X16_LIB=$PWD/htscodecs_amd/variants/libx8exp.so
N=${1:-11520}
for q in 12 14 15; do echo "dec q1a=$q"; X8D_Q1A=$q timeout -k 10 120 python3 tools/rate_4x8.py $N 2>/dev/null | grep "order 1" || exit 1; done
for c in "30 8" "28 7" "24 6" "24 8" "30 10" "30 15" "16 8" "15 8"; do set -- $c; echo "enc qpw=$1 spw=$2"; X8E_QPW=$1 X8E_SPW=$2 timeout -k 10 120 python3 tools/rate_4x8.py $N 2>/dev/null | grep "order 1" || exit 1; done
