#!/bin/bash
# streams per wave of the packed-row decode class on partly filled rounds (1 MiB q40 blocks, order 1)
set -o pipefail
export BS=1048576
for n in 2048 4096; do
 for o in "dec_direct=0" "dec_direct=0,dec_qpw_pk=12" "dec_direct=0,dec_qpw_pk=8" "dec_direct=0,dec_qpw_pk=6" "dec_direct=0,dec_qpw_pk=4" "dec_direct=0,dec_qpw_pk=3" "dec_direct=0,dec_qpw_pk=2"; do
  echo "n=$n $o"; OPTS=$o timeout -k 10 120 python3 tools/sweep.py $n 2>/dev/null || exit 1
 done
done
