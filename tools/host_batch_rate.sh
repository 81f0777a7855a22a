#!/bin/bash
# PCIe-inclusive rate of the host-buffer batch entry points: the reference's `-t` benchmark
# loop (tests/rANS_static4x16pr_test.c:176-224) over a tiled q40 file, through rans4x16_hip_*_batch.
# usage: host_batch_rate.sh [GiB (<4)] [order]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
GIB=${1:-3}; ORDER=${2:-1}
cd $R
python3 - "$GIB" <<'P'
import sys, numpy as np
sys.path.insert(0, sys.argv[0] if False else "tests")
import datagen
g = float(sys.argv[1]); n = int(g * (1 << 30))
b = datagen.base_text("q40+dir")
np.tile(b, n // len(b) + 1)[:n].tofile("/tmp/host_batch.q40")
print("wrote", n, "bytes")
P
LD_LIBRARY_PATH=$R/htscodecs_amd:$LD_LIBRARY_PATH $R/tools/rans4x16pr_hip -t -b -o$ORDER /tmp/host_batch.q40 2>&1
rm -f /tmp/host_batch.q40
