#!/usr/bin/env python3
"""Generate tests/golden/.  Run in the build container only; the outputs are committed so that the GPU box (which has
no /root/reference) can pin the oracle and the HIP path against them.

Two kinds of vectors, and they do not have the same standing:
  * REFERENCE-HELD: the inputs and the compressed fixtures are copied byte for byte from the reference's own test data
    (tests/dat/, tests/dat/r4x16, tests/dat/r4x8).  These pin the oracle (tests/test_oracle.py: decode + byte-identical
    re-encode of all of them).
  * edge.json: generated edge cases with the ORACLE's outputs.  They are regression vectors for the HIP path against the
    oracle at sizes and shapes the fixtures do not have; they pin nothing about the reference by themselves.  (Rounds
    1-3 generated them with a build of the reference sources against an empty stand-in config.h; that build is not the
    reference's own and was removed in round 4 - see oracle/Makefile.  The oracle reproduces every one of those vectors
    byte for byte, which is how this script was checked when it was switched over.)

What is written (data only — no reference source):
  tests/golden/dat/<name>.nl      the reference's own test inputs, first column with newlines
                                  removed, i.e. exactly what tests/rans4x16.test:11 feeds the codec
  tests/golden/r4x16/<name>.<o>   the reference's committed compressed fixtures (tests/dat/r4x16)
  tests/golden/edge.json          generated edge cases: [input spec, order] -> reference output
                                  (base64 when small, md5+length always)
"""
import base64
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/tests/dat"
GOLD = os.path.join(ROOT, "tests", "golden")


def strip_first_column(path):
    out = bytearray()
    with open(path, "rb") as f:
        for line in f.read().split(b"\n"):
            out += line.split(b"\t")[0]
    return bytes(out)


def main():
    os.makedirs(os.path.join(GOLD, "dat"), exist_ok=True)
    os.makedirs(os.path.join(GOLD, "r4x16"), exist_ok=True)
    for name in ("q4", "q8", "q40+dir", "qvar"):
        with open(os.path.join(GOLD, "dat", name + ".nl"), "wb") as f:
            f.write(strip_first_column(os.path.join(REF, name)))
    for fn in sorted(os.listdir(os.path.join(REF, "r4x16"))):
        shutil.copyfile(os.path.join(REF, "r4x16", fn), os.path.join(GOLD, "r4x16", fn))
        os.chmod(os.path.join(GOLD, "r4x16", fn), 0o644)

    import cpu_libs
    import datagen
    ref = cpu_libs.oracle()

    all_orders = [0, 1, 64, 65, 128, 129, 192, 193]
    cases = []

    def add(spec, orders):
        data = datagen.make(spec).tobytes()
        for o in orders:
            comp = ref.compress(data, o)
            assert comp is not None, (spec, o)
            back = ref.uncompress(comp, capacity=len(data),
                                  out_size_hint=len(data))
            assert back == data, (spec, o)
            e = {"in": spec, "n": len(data), "order": o, "len": len(comp),
                 "md5": hashlib.md5(comp).hexdigest()}
            if len(comp) <= 24576:
                e["out"] = base64.b64encode(comp).decode()
            cases.append(e)

    # tiny inputs: CAT fall-backs, order-1 forced to order-0 below 8 bytes, n % 4 coverage
    for n in (0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 20, 21, 27, 33):
        add(["tile", "q8", n, 0, 5], all_orders + [8, 9])
    for n in (1, 21, 1000):
        add(["const", n, 65], all_orders + [8, 9, 32, 33])
    # n % 4 and quarter boundaries on real data, all three alphabets
    for name in ("q4", "q8", "q40+dir"):
        for n in (1021, 1022, 1023, 1024, 4099):
            add(["tile", name, n, 0, 17], all_orders)
    # stripes: default N=4, explicit N=2,3,5; methods limited by the order bits
    for n in (22, 100, 4097):
        for o in (8, 9, 0x48, 0xc9, (2 << 8) | 9, (3 << 8) | 0xc9, (5 << 8) | 8):
            add(["tile", "q40+dir", n, 0, 3], [o])
    # pack alphabet sizes 1,2,3,4,5,16,17 and the n==256 wrap quirk (stays packed, copy)
    for nsym in (1, 2, 3, 4, 5, 16, 17):
        add(["rand", 3001, 7, nsym, 40], [128, 129, 192, 193])
    add(["rand", 100000, 3, 256, 0], [0, 1, 128, 129, 192])
    add(["rand", 5000, 3, 250, 0], [128, 193])
    # run-heavy data: RLE kept, meta compressed vs raw (short input -> raw meta)
    add(["runs", 50000, 6, 12, 1, 48], all_orders)
    add(["runs", 600, 3, 9, 2, 48], [64, 65, 192, 193])
    add(["runs", 20000, 200, 5, 3, 20], [64, 65, 193])
    # order-1 table variants: raw table (q4/q8), compressed table (q40), shift 12
    add(["tile", "q40+dir", 65536, 0, 0], [1, 193])
    add(["markov", 30000, 40, 4, 33, 0.55], [0, 1, 65])
    add(["weighted", 1 << 20, [3000] + [1] * 255, 1], [0, 1])         # -> 12-bit order-1 table
    add(["weighted", 200000, [100000] + [1] * 255, 2], [1, 193])
    # hist1_4 switches implementation at 500000 bytes (utils.h:145): same result expected
    for n in (499999, 500000, 500001):
        add(["tile", "q8", n, 0, 11], [1])
    # headline block shapes: 1 MiB tiles (sizes must match BASELINE.md) and a 64 KiB tile
    for name in ("q4", "q8", "q40+dir"):
        add(["tile", name, 1 << 20, 0, 0], [0, 1, 193])
        add(["tile", name, 1 << 16, 1, None], [0, 1, 193])
    # explicit no-size streams
    add(["tile", "q8", 5000, 0, 0], [16, 17, 0xd1])

    with open(os.path.join(GOLD, "edge.json"), "w") as f:
        json.dump({"generator": "oracle/make_golden.py", "cases": cases}, f, indent=0)
    tot = sum(len(c.get("out", "")) for c in cases)
    print(f"{len(cases)} edge cases, {tot/1e6:.2f} MB of base64")


if __name__ == "__main__":
    main()
