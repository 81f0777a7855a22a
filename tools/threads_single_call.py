#!/usr/bin/env python3
"""The literal drop-in case: T host threads (a CRAM writer's pool) each calling rans_compress_to_4x16 /
rans_uncompress_to_4x16 on its own 1 MiB blocks, one block per call.  Aggregate MB/s over all threads.
usage: threads_single_call.py [threads=32] [blocks per thread=6]"""
import ctypes as C, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen
import htscodecs_amd as H
L = H.load()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
bs, order = 1 << 20, 1
cap = L.rans_compress_bound_4x16(bs, order)
srcs = [np.ascontiguousarray(datagen.tile("q40+dir", bs, t)) for t in range(T)]
comps = [np.zeros(cap, dtype=np.uint8) for _ in range(T)]
backs = [np.zeros(bs, dtype=np.uint8) for _ in range(T)]
sizes = [0] * T
def enc(t, reps):
    for _ in range(reps):
        n = C.c_uint(cap)
        assert L.rans_compress_to_4x16(srcs[t].ctypes.data, bs, comps[t].ctypes.data, C.byref(n), order)
        sizes[t] = n.value
def dec(t, reps):
    for _ in range(reps):
        n = C.c_uint(bs)
        assert L.rans_uncompress_to_4x16(comps[t].ctypes.data, sizes[t], backs[t].ctypes.data, C.byref(n))
def run(fn, reps):
    th = [threading.Thread(target=fn, args=(t, reps)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    return time.perf_counter() - t0
run(enc, 1); run(dec, 1)                                  # contexts, pinned buffers
te = run(enc, K); td = run(dec, K)
assert all((b == s).all() for b, s in zip(backs, srcs))
tot = T * K * bs
print(f"{T} threads x {K} calls of 1 MiB: encode {tot/te/1e6:.1f} MB/s aggregate ({te/K*1e3:.1f} ms per call), "
      f"decode {tot/td/1e6:.1f} MB/s aggregate ({td/K*1e3:.1f} ms per call)")
