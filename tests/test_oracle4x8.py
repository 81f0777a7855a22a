"""rANS 4x8 (CRAM 3.0's codec): the oracle's restatement (oracle/rans4x8_oracle.c) pinned by the
reference's own eight fixtures (tests/golden/r4x8/ = tests/dat/r4x8/ of the reference, what tests/rans4x8.test
decodes): decode, and byte-identical re-encode."""
import ctypes as C
import os

import numpy as np
import pytest

import cpu_libs
import datagen

GOLD = os.path.join(datagen.GOLDEN, "r4x8")
FIXTURES = sorted(os.listdir(GOLD))


class Codec8:
    """rans_compress / rans_uncompress (htscodecs/rANS_static.h:41-44) of one library; results are malloc'd."""

    def __init__(self, lib, prefix):
        self.c = getattr(lib, prefix + "rans_compress")
        self.c.restype = C.c_void_p
        self.c.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint), C.c_int]
        self.u = getattr(lib, prefix + "rans_uncompress")
        self.u.restype = C.c_void_p
        self.u.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint)]
        self.free = C.CDLL(None).free
        self.free.argtypes = [C.c_void_p]

    def _take(self, p, n):
        if not p:
            return None
        res = C.string_at(p, n.value)
        self.free(p)
        return res

    def compress(self, data, order):
        src = np.frombuffer(bytes(data), dtype=np.uint8)
        n = C.c_uint(0)
        return self._take(self.c(src.ctypes.data, len(src), C.byref(n), order), n)

    def uncompress(self, comp):
        src = np.frombuffer(bytes(comp) + b"\0" * 16, dtype=np.uint8)     # (slack: the reference's table reader looks ahead)
        n = C.c_uint(0)
        return self._take(self.u(src.ctypes.data, len(comp), C.byref(n)), n)


@pytest.fixture(scope="module")
def orc8():
    return Codec8(cpu_libs.oracle().lib, "orc8_")


@pytest.mark.parametrize("fn", FIXTURES)
def test_fixture_decodes_and_reencodes_byte_identically(orc8, fn):
    name, order = fn.rsplit(".", 1)
    with open(os.path.join(GOLD, fn), "rb") as f:
        comp = f.read()
    plain = datagen.base_text(name).tobytes()
    assert orc8.uncompress(comp) == plain
    assert orc8.compress(plain, int(order)) == comp


def _inputs(rs, count):
    out = []
    for _ in range(count):
        kind = rs.randint(0, 5)
        n = int(rs.choice([rs.randint(1, 40), rs.randint(40, 3000), rs.randint(3000, 70000)]))
        seed = int(rs.randint(1, 1 << 30))
        if kind == 0:
            a = datagen.rand(n, seed, int(rs.randint(1, 257)), 0)
        elif kind == 1:
            a = datagen.runs(n, int(rs.randint(1, 40)), int(rs.randint(2, 60)), seed, 30)
        elif kind == 2:
            w = rs.random_sample(int(rs.randint(2, 256))) ** int(rs.randint(1, 12))
            a = datagen.weighted(n, w + 1e-9, seed)
        elif kind == 3:
            a = datagen.tile(str(rs.choice(["q4", "q8", "q40+dir", "qvar"])), n, 0, seed)
        else:
            a = datagen.const(n, int(rs.randint(0, 256)))
        out.append(a.tobytes())
    return out


def test_round_trips_and_damaged_streams_do_not_crash(orc8):
    """Random inputs of every kind round-trip through the restatement; damaged streams and garbage of every small length
    are refused or decoded, never crashed on (the sanitizer build runs the same cases: test_oracle_sanitized.py)."""
    rs = np.random.RandomState(48)
    for d in _inputs(rs, 250) + [b"a", b"ab", b"abc", b"abcd", b"abcde", bytes(range(256)) * 3,
                                 datagen.tile("q40+dir", 1 << 20, 3).tobytes()]:
        for order in (0, 1):
            comp = orc8.compress(d, order)
            assert comp is not None and orc8.uncompress(comp) == d, (len(d), order)
    rs = np.random.RandomState(84)
    for d in _inputs(rs, 120):
        comp = bytearray(orc8.compress(d, int(rs.randint(0, 2))))
        for _ in range(6):
            bad = bytearray(comp)
            mode = rs.randint(0, 3)
            if mode == 0:
                bad[int(rs.randint(9, len(bad)))] ^= int(rs.randint(1, 256))
            elif mode == 1:
                cut = int(rs.randint(9, len(bad) + 1))
                bad = bad[:cut]
                bad[1:5] = int(len(bad) - 9).to_bytes(4, "little")          # keep the size field consistent
            else:
                bad[int(rs.randint(9, len(bad)))] ^= 1 << int(rs.randint(0, 8))
            orc8.uncompress(bytes(bad))
    for n in (0, 1, 8, 9, 10, 25, 26, 27, 40, 300):
        for _ in range(30):
            junk = bytes(rs.randint(0, 256, size=n).astype(np.uint8))
            orc8.uncompress(junk)
