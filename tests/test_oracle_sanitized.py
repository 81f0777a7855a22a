"""The oracle under AddressSanitizer + UBSan (oracle/Makefile: liboracle4x16_asan.so; the reference's CI does the
same with its own library, .cirrus.yml:35-42): every fixture both ways, edge sizes and a few thousand damaged and
random streams through both restatements (4x16 and 4x8) in a child process with the sanitizer runtime preloaded.
A finding aborts the child."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import datagen
lib = C.CDLL(sys.argv[2])
free = C.CDLL(None).free; free.argtypes = [C.c_void_p]
def call(name, data, *extra):
    fn = getattr(lib, name); fn.restype = C.c_void_p
    src = np.frombuffer(bytes(data) + bytes(32), dtype=np.uint8)      # slack: the table readers look ahead by design
    n = C.c_uint(0)
    p = fn(C.c_void_p(src.ctypes.data), C.c_uint(len(data)), C.byref(n), *extra)
    if not p: return None
    out = C.string_at(p, n.value); free(p); return out
rs = np.random.RandomState(11)
gold = os.path.join(sys.argv[1], "golden")
count = 0
for codec, sub, enc, dec in (("4x16", "r4x16", "orc_rans_compress_4x16", "orc_rans_uncompress_4x16"),
                             ("4x8", "r4x8", "orc8_rans_compress", "orc8_rans_uncompress")):
    for fn in sorted(os.listdir(os.path.join(gold, sub))):
        name, order = fn.rsplit(".", 1)
        comp = open(os.path.join(gold, sub, fn), "rb").read()
        plain = datagen.base_text(name).tobytes()
        assert call(dec, comp) == plain, fn
        assert call(enc, plain, C.c_int(int(order))) == comp, fn
    orders = [0, 1, 64, 65, 128, 129, 192, 193, 9, 0x48, 32] if codec == "4x16" else [0, 1]
    for n in list(range(0, 40)) + [255, 256, 257, 1000, 4095, 4096, 65536, 70001]:
        for kind in range(3):
            d = (datagen.rand(n, n + kind, 1 + (n * 7 + kind) % 256, 0) if kind == 0 else
                 datagen.runs(n, 5, 9, n + 1, 40) if kind == 1 else datagen.tile("q40+dir", n, n)).tobytes()
            for o in orders:
                c = call(enc, d, C.c_int(o))
                if c is None:
                    assert n == 0 and codec == "4x8"
                    continue
                if o & 0x10:
                    continue                                             # X_NOSZ streams need a caller buffer
                assert call(dec, c) == d, (codec, n, o)
                for _ in range(4):                                       # damaged copies must fail cleanly or decode
                    b = bytearray(c)
                    lo = 6 if codec == "4x16" else 9                     # (size fields stay: a flipped size bit can ask
                    if len(b) > lo + 1:                                  #  for gigabytes, legitimately)
                        b[int(rs.randint(lo, len(b)))] ^= 1 << int(rs.randint(0, 8))
                    call(dec, bytes(b))
                    count += 1
    for n in (0, 1, 2, 5, 9, 16, 27, 40, 200):
        for _ in range(60):
            call(dec, bytes(rs.randint(0, 256, size=n).astype(np.uint8)))
            count += 1
print("sanitized cases:", count)
'''


def test_oracle_under_asan_ubsan():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle4x16_asan.so"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available: " + r.stdout[-300:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD, HERE, os.path.join(ROOT, "oracle", "liboracle4x16_asan.so")],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "sanitized cases:" in r.stdout
