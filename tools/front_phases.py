#!/usr/bin/env python3
"""Phase times of k_enc_front from a variant build (hipcc ... -DR4X16_PROF_FRONT -> htscodecs_amd/variants/libprof.so):
usage (GPU box): DATA=q4 ORDER=193 BS=1048576 python3 tools/front_phases.py 4096"""
import ctypes as C, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from htscodecs_amd import lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "htscodecs_amd", "variants", "libprof.so")
import sweep
L = _lib.load()
nblk = int(sys.argv[1])
r = sweep.run(nblk, int(os.environ.get("BS", 1 << 20)), os.environ.get("DATA", "q4"), int(os.environ.get("ORDER", 193)), reps=1)
buf = (C.c_ulonglong * 16)()
fn = C.CDLL(_lib.LIB_PATH).rans4x16_hip_debug_front_prof
assert fn(buf, 0) == 0
names = ["header", "present8 (for PACK)", "pack", "to RLE", "hist8 (packed)", "rle split", "to order-1", "pass 1 (present8 / hist8)", "alphabet + zeroing", "pair histogram", "hand-over"]
tot = sum(buf[:11])
calls = 2 * nblk            # warm step + timed step
print(r)
for k, nm in enumerate(names):
    print(f"{nm:28s} {buf[k] / calls / 100.0:9.1f} us per block  {100.0 * buf[k] / max(tot, 1):5.1f} %")   # wall_clock64: 100 MHz
for k, nm in zip(range(11, 16), ["rle: repeat counts", "rle: symbol choice", "rle: walk A", "rle: scans", "rle: walk B"]):
    print(f"  {nm:26s} {buf[k] / calls / 100.0:9.1f} us per block")

fn2 = C.CDLL(_lib.LIB_PATH).rans4x16_hip_debug_tables_prof
assert fn2(buf, 0) == 0
print("k_enc_tables (order-1 path), us per block:")
for k, nm in enumerate(["load counters, totals", "compute_shift", "normalise rows", "serialise table", "encoder image", "nested table coding"]):
    print(f"  {nm:26s} {buf[k] / calls / 100.0:9.1f}")
