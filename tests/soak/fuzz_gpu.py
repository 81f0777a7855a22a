#!/usr/bin/env python3
"""One-off differential fuzz on the GPU box: random inputs x random order flags, device library vs the oracle.
usage: fuzz_gpu.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import htscodecs_amd as H
import cpu_libs, datagen

def gen(rs):
    kind = int(rs.randint(0, 7))
    n = int(rs.choice([0, 1, 3, 7, 8, 21, 100, 1000, 4097, 20000, 70000, 300000], p=[.02,.03,.03,.03,.03,.05,.15,.2,.15,.15,.11,.05]))
    n = max(0, n + int(rs.randint(-3, 4)))
    seed = int(rs.randint(0, 1 << 30))
    if kind == 0: a = datagen.rand(n, seed, int(rs.randint(1, 257)), 0)
    elif kind == 1: a = datagen.runs(n, int(rs.randint(1, 40)), int(rs.randint(1, 200)), seed, int(rs.randint(0, 200)))
    elif kind == 2: a = datagen.weighted(n, [int(rs.randint(1, 5000))] + [1] * int(rs.randint(1, 255)), seed)
    elif kind == 3: a = datagen.tile(str(rs.choice(["q4", "q8", "q40+dir", "qvar"])), n, int(rs.randint(0, 50)))
    elif kind == 4: a = datagen.markov(min(n, 60000), int(rs.randint(2, 220)), seed, 0, float(rs.random_sample()))
    elif kind == 5: a = datagen.const(n, int(rs.randint(0, 256)))
    else: a = datagen.markov(min(n, 60000), int(rs.randint(40, 160)), seed, int(rs.randint(0, 90)), 0.3)
    return np.ascontiguousarray(a).tobytes()

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    orc = cpu_libs.oracle()
    datas = [gen(rs) for _ in range(cases)]
    flagsets = [0, 1, 64, 65, 128, 129, 192, 193, 16, 17, 32, 8, 9, 8 | 64 | 1, 8 | 128 | 1 | (3 << 8), 8 | (2 << 8)]
    orders = [int(rs.choice(flagsets)) for _ in datas]
    enc, st = H.compress_batch(datas, orders)
    bad = 0
    comps = []
    for d, o, e in zip(datas, orders, enc):
        want = orc.compress(d, o)
        comps.append(want)
        if e != want:
            bad += 1
            if bad < 10: print("ENC MISMATCH len", len(d), "order", o, None if e is None else len(e), len(want))
    dec, st = H.uncompress_batch(comps, [len(d) for d in datas])
    for d, o, x in zip(datas, orders, dec):
        if x != d:
            bad += 1
            if bad < 20: print("DEC MISMATCH len", len(d), "order", o)
    print("cases", cases, "mismatches", bad)
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
