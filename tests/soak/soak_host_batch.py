#!/usr/bin/env python3
"""Soak of the host-buffer batch calls at scale (the staged pipeline of r4x16_host.hip): many blocks of irregular
sizes and per-block orders, every result compared with the oracle.
usage: soak_host_batch.py [blocks] [max block bytes] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import htscodecs_amd as H
import cpu_libs, datagen

def main():
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    maxb = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    rs = np.random.RandomState(int(sys.argv[3]) if len(sys.argv) > 3 else 5)
    orc = cpu_libs.oracle()
    texts = {nm: np.tile(np.ascontiguousarray(datagen.base_text(nm)), 3) for nm in ("q4", "q8", "q40+dir", "qvar")}
    rnd = rs.randint(0, 256, 1 << 22).astype(np.uint8)
    datas, orders = [], []
    flagsets = [0, 1, 64, 65, 128, 129, 192, 193, 16, 17, 32]
    for i in range(nblk):
        n = int(rs.choice([rs.randint(0, 200), rs.randint(200, 5000), rs.randint(5000, maxb + 1)], p=[.1, .3, .6]))
        kind = rs.randint(0, 10)
        if kind == 0:
            o = int(rs.randint(0, len(rnd) - n)); d = rnd[o:o + n]
        else:
            t = texts[("q4", "q8", "q40+dir", "qvar")[kind % 4]]
            n = min(n, len(t) - 1); o = int(rs.randint(0, len(t) - n)); d = t[o:o + n]
        datas.append(d.tobytes()); orders.append(int(rs.choice(flagsets)))
    tot = sum(len(d) for d in datas)
    print("blocks", nblk, "bytes", tot, flush=True)
    t0 = time.time(); enc, st = H.compress_batch(datas, orders); t1 = time.time()
    print("compress_batch %.2f s (%.2f GB/s incl. python marshalling)" % (t1 - t0, tot / (t1 - t0) / 1e9), flush=True)
    bad = 0
    want = []
    for d, o, e in zip(datas, orders, enc):
        w = orc.compress(d, o); want.append(w)
        if e != w: bad += 1
    print("encode mismatches", bad, flush=True)
    t0 = time.time(); dec, st = H.uncompress_batch(want, [len(d) for d in datas]); t1 = time.time()
    print("uncompress_batch %.2f s" % (t1 - t0), flush=True)
    badd = sum(1 for d, x in zip(datas, dec) if x != d)
    print("decode mismatches", badd)
    # try-K mode on a slice
    sl = datas[:4000]
    methods = [0, 1, 128, 129, 64, 65, 192, 193, 193 + 8]           # tokenise_name3.c:1260, level 9
    t0 = time.time(); got, chosen, st = H.compress_best_batch(sl, methods); t1 = time.time()
    print("compress_best_batch, 9 methods, %d blocks, %.1f MB: %.2f s" % (len(sl), sum(len(d) for d in sl) / 1e6, t1 - t0), flush=True)
    badb = 0
    for d, g, c in zip(sl, got, chosen):
        best, bm = None, None
        for m in methods:
            if len(d) % 4 != 0 and (m & 8): continue
            w = orc.compress(d, m)
            if best is None or len(w) < len(best): best, bm = w, m
        if g != best or c != bm: badb += 1
    print("best-of-9 mismatches", badb)
    return 1 if bad or badd or badb else 0

if __name__ == "__main__":
    sys.exit(main())
