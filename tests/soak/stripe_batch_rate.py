#!/usr/bin/env python3
"""Rate of X_STRIPE blocks through the host batch calls (they are expanded into device batches of their planes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, htscodecs_amd as H, datagen, cpu_libs
orc = cpu_libs.oracle()
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
order = int(sys.argv[3]) if len(sys.argv) > 3 else 9
datas = [datagen.tile("q40+dir", sz, i).tobytes() for i in range(nblk)]
for rep in range(3):
    t0 = time.time(); enc, st = H.compress_batch(datas, [order] * nblk); t1 = time.time()
    dec, st2 = H.uncompress_batch(enc, [sz] * nblk); t2 = time.time()
    print("stripe order %d: %d x %d B  enc %.1f ms (%.2f ms/block)  dec %.1f ms (%.2f ms/block)" % (order, nblk, sz, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / nblk, (t2 - t1) * 1e3, (t2 - t1) * 1e3 / nblk), flush=True)
assert dec == datas
bad = sum(1 for d, e in zip(datas[:100], enc[:100]) if orc.compress(d, order) != e)
print("mismatches in the first 100:", bad)
