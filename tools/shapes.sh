#!/bin/bash
# The other shapes of BASELINE.json through tools/sweep.py (chain-kernel times from HIP events, whole-step time,
# round trip checked), one JSON line each -> profiles/<tag>_shapes.jsonl.  Runs on the GPU box.
cd "$(dirname "$0")/.."
run() { DATA=$1 ORDER=$2 BS=$3 python3 tools/sweep.py $4 2>/dev/null | grep nblk; }
run q40+dir 1 1048576 23040        # headline shape
run q40+dir 0 1048576 15360        # configs[1]: order-0, 1 MiB q40
run q40+dir 0 262144 122880
run q8 1 1048576 8192              # configs[2]: q8 order-1
run q8 1 1048576 32768
run q4 1 262144 65536
run q4 193 1048576 4096            # configs[3]: q4 with X_PACK|X_RLE
run q4 193 1048576 16384
run mixed 1 65536 32768            # configs[4]: one GPU's share
run mixed 1 65536 98304
run q40+dir 1 65536 61440          # small q40 blocks
run q8 65 1048576 4096             # X_RLE without X_PACK
run q8 65 1048576 16384
