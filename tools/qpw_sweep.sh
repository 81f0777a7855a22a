#!/bin/bash
# Streams-per-wave sweep of the packed-row decode class (46-symbol order-1 tables): each setting is timed on a batch
# of three full rounds of its own resident stream count.  usage: tools/qpw_sweep.sh  (on the GPU box)
cd "$(dirname "$0")/.."
for cfg in "14 10752" "13 9984" "12 9216" "10 10240" "8 10240" "7 10752" "16 8192"; do
  set -- $cfg
  echo "R4X16_DEC_QPW_PK=$1 resident=$2"
  R4X16_DEC_QPW_PK=$1 BS=262144 python3 tools/sweep.py $((3 * $2)) || exit 1
done
