#!/usr/bin/env python3
"""rANS 4x8 on the GPU box, at length: random inputs x both orders against the oracle's restatement (bytes identical both
ways), then damaged streams (what the oracle rejects the device rejects; what both accept decodes identically; the
device may refuse more - status 3 / 7 - never less).
usage: fuzz_4x8_gpu.py [rounds of 300 inputs] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import htscodecs_amd as H
import cpu_libs, datagen
from test_oracle4x8 import Codec8, _inputs

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 48)
    H.load()
    orc = Codec8(cpu_libs.oracle().lib, "orc8_")
    cases = mism = dam = agree = stricter = viol = 0
    for r in range(rounds):
        datas = _inputs(rs, 300)
        if r % 3 == 0:
            datas += [datagen.tile(str(rs.choice(["q4", "q8", "q40+dir", "qvar"])), int(rs.randint(100000, 1500000)), int(rs.randint(0, 50))).tobytes()
                      for _ in range(4)]
        orders = [int(rs.randint(0, 2)) for _ in datas]
        enc, st = H.compress_batch_4x8(datas, orders)
        want = [orc.compress(d, o) for d, o in zip(datas, orders)]
        mism += sum(1 for e, w in zip(enc, want) if e != w)
        dec, st = H.uncompress_batch_4x8(want, [len(d) for d in datas])
        mism += sum(1 for x, d in zip(dec, datas) if x != d)
        cases += len(datas)
        bads, caps, refs = [], [], []
        for d, w in zip(datas, want):
            if len(w) < 12:
                continue
            for _ in range(3):
                bad = bytearray(w)
                mode = int(rs.randint(0, 4))
                if mode == 0: bad[int(rs.randint(9, len(bad)))] ^= int(rs.randint(1, 256))
                elif mode == 1:
                    bad = bad[:int(rs.randint(9, len(bad) + 1))]; bad[1:5] = int(len(bad) - 9).to_bytes(4, "little")
                elif mode == 2: bad[int(rs.randint(0, len(bad)))] ^= 1 << int(rs.randint(0, 8))
                else:
                    a = int(rs.randint(9, len(bad))); b = min(len(bad), a + int(rs.randint(1, 40)))
                    bad[a:b] = bytes(rs.randint(0, 256, size=b - a).astype(np.uint8))
                bads.append(bytes(bad)); caps.append(len(d) + 64); refs.append(orc.uncompress(bytes(bad)))
        dec, st = H.uncompress_batch_4x8(bads, caps)
        for x, s, rf, cap in zip(dec, st, refs, caps):
            dam += 1
            if rf is None: viol += x is not None
            elif len(rf) > cap: viol += not (x is None and s == 1)
            elif x is not None: agree += 1; viol += x != rf
            else: stricter += 1; viol += s not in (3, 7)
        print(f"round {r}: cases {cases} mismatches {mism} | damaged {dam} both accept {agree} device stricter {stricter} violations {viol}", flush=True)
    print(f"cases {cases} mismatches {mism} | damaged {dam} both accept, same bytes {agree} | device stricter {stricter} | violations {viol}")
    return 1 if (mism or viol) else 0

if __name__ == "__main__":
    sys.exit(main())
