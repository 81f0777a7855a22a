#!/usr/bin/env python3
"""Hostile-input fuzz on the GPU box (the reference's harness is
tests/rANS_static4x16pr_fuzz.c:71): valid streams of every flag combination are damaged (bit flips,
byte smashes, truncation, splices, header-targeted edits) and decoded by the device library and by
the oracle.  Contract checked per case:
    oracle rejects             -> device rejects
    both accept                -> identical bytes
    device rejects, oracle not -> only with one of the documented stricter statuses (6, 7, 8)
usage: fuzz_damaged_gpu.py [valid streams] [mutants per stream] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import htscodecs_amd as H
import cpu_libs, datagen

FLAGSETS = [0, 1, 64, 65, 128, 129, 192, 193, 16, 17, 32, 8, 9, 8 | 64 | 1, 8 | 128 | 1 | (3 << 8), 8 | (2 << 8)]


def gen(rs):
    kind = int(rs.randint(0, 6))
    n = int(rs.choice([5, 21, 100, 1000, 4097, 20000, 70000], p=[.05, .1, .2, .25, .2, .15, .05])) + int(rs.randint(0, 4))
    seed = int(rs.randint(0, 1 << 30))
    if kind == 0: a = datagen.rand(n, seed, int(rs.randint(1, 257)), 0)
    elif kind == 1: a = datagen.runs(n, int(rs.randint(1, 40)), int(rs.randint(1, 200)), seed, int(rs.randint(0, 200)))
    elif kind == 2: a = datagen.weighted(n, [int(rs.randint(1, 5000))] + [1] * int(rs.randint(1, 255)), seed)
    elif kind == 3: a = datagen.tile(str(rs.choice(["q4", "q8", "q40+dir", "qvar"])), n, int(rs.randint(0, 50)))
    elif kind == 4: a = datagen.markov(min(n, 60000), int(rs.randint(2, 220)), seed, 0, float(rs.random_sample()))
    else: a = datagen.const(n, int(rs.randint(0, 256)))
    return np.ascontiguousarray(a).tobytes()


def mutate(rs, comp, others):
    bad = bytearray(comp)
    mode = int(rs.randint(0, 7))
    if mode == 0:                                   # one bit anywhere
        p = int(rs.randint(0, len(bad))); bad[p] ^= 1 << int(rs.randint(0, 8))
    elif mode == 1:                                 # truncate
        bad = bad[:int(rs.randint(1, len(bad) + 1))]
    elif mode == 2:                                 # smash a byte in the header / table region
        p = int(rs.randint(0, min(len(bad), 64))); bad[p] = int(rs.randint(0, 256))
    elif mode == 3:                                 # several byte smashes anywhere
        for _ in range(int(rs.randint(1, 6))):
            p = int(rs.randint(0, len(bad))); bad[p] = int(rs.randint(0, 256))
    elif mode == 4:                                 # splice the tail of another stream
        o = others[int(rs.randint(0, len(others)))]
        cut = int(rs.randint(1, len(bad) + 1))
        bad = bad[:cut] + bytearray(o[int(rs.randint(0, len(o))):])
    elif mode == 5:                                 # extreme values where sizes and counts live
        p = int(rs.randint(0, min(len(bad), 24))); bad[p] = int(rs.choice([0, 0x7f, 0x80, 0xff]))
    else:                                           # insert or delete a few bytes
        p = int(rs.randint(0, len(bad)))
        if rs.randint(0, 2): del bad[p:p + int(rs.randint(1, 4))]
        else: bad[p:p] = bytes(rs.randint(0, 256, int(rs.randint(1, 4))).astype(np.uint8))
    return bytes(bad) if len(bad) else b"\x00"


def main():
    nvalid = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rs = np.random.RandomState(int(sys.argv[3]) if len(sys.argv) > 3 else 11)
    orc = cpu_libs.oracle()
    datas = [gen(rs) for _ in range(nvalid)]
    comps = [orc.compress(d, int(rs.choice(FLAGSETS))) for d in datas]
    bads, caps, refs = [], [], []
    for d, c in zip(datas, comps):
        for _ in range(per):
            b = mutate(rs, c, comps)
            cap = len(d) + 64
            bads.append(b); caps.append(cap)
            refs.append(orc.uncompress(b, capacity=cap, out_size_hint=cap))
    print("cases", len(bads), "oracle accepts", sum(r is not None for r in refs), flush=True)
    dec, st = H.uncompress_batch(bads, caps)
    agree = stricter = wrong = 0
    by_status = {}
    for b, x, s, r in zip(bads, dec, st, refs):
        if r is None:
            if x is not None or s == 0:
                wrong += 1
                if wrong < 10: print("DEVICE ACCEPTED what the oracle rejects: flags %#x len %d" % (b[0], len(b)))
        elif x is not None:
            if x == r: agree += 1
            else:
                wrong += 1
                if wrong < 10:
                    print("DIFFERENT BYTES: flags %#x len %d\n  in  %s\n  dev %s\n  ref %s" % (b[0], len(b), b[:80].hex(), x[:60].hex(), r[:60].hex()), "lens", len(x), len(r))
        else:
            by_status[int(s)] = by_status.get(int(s), 0) + 1
            if s in (6, 7, 8): stricter += 1
            else:
                wrong += 1
                if wrong < 10: print("DEVICE REJECTED (status %d) what the oracle accepts: flags %#x len %d" % (s, b[0], len(b)))
    # the library must still work after all that
    good, _ = H.uncompress_batch(comps[:50], [len(d) for d in datas[:50]])
    alive = all(g == d for g, d in zip(good, datas[:50]))
    print("both accept, same bytes", agree, "| device stricter", stricter, by_status, "| violations", wrong, "| alive", alive)
    return 1 if wrong or not alive else 0


if __name__ == "__main__":
    sys.exit(main())
