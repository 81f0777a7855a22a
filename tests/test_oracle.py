"""Pins the CPU oracle (oracle/rans4x16_oracle.c) to the reference's own golden data:
the 24 committed fixtures of tests/dat/r4x16 (decode AND byte-identical re-encode, as
SURVEY.md §8c established for the reference itself), the varint known-answer values of
tests/varint_test.c:145-155, and the generated edge vectors in tests/golden/edge.json
(made from the real reference by oracle/make_golden.py)."""
import base64
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import datagen

GOLD = datagen.GOLDEN
FIXTURES = sorted(os.listdir(os.path.join(GOLD, "r4x16")))


def _fixture(fn):
    name, order = fn.rsplit(".", 1)
    with open(os.path.join(GOLD, "r4x16", fn), "rb") as f:
        comp = f.read()
    return name, int(order), comp, datagen.base_text(name).tobytes()


@pytest.mark.parametrize("fn", FIXTURES)
def test_fixture_decode(oracle, fn):
    name, order, comp, plain = _fixture(fn)
    assert oracle.uncompress(comp) == plain


@pytest.mark.parametrize("fn", FIXTURES)
def test_fixture_encode_bit_exact(oracle, fn):
    name, order, comp, plain = _fixture(fn)
    got = oracle.compress(plain, order)
    assert got is not None
    assert hashlib.md5(got).hexdigest() == hashlib.md5(comp).hexdigest()
    assert got == comp


def _edge_cases():
    with open(os.path.join(GOLD, "edge.json")) as f:
        return json.load(f)["cases"]


EDGE = _edge_cases()


@pytest.mark.parametrize("idx", range(len(EDGE)))
def test_edge_vectors(oracle, idx):
    e = EDGE[idx]
    data = datagen.make(e["in"]).tobytes()
    assert len(data) == e["n"]
    got = oracle.compress(data, e["order"])
    assert got is not None
    assert len(got) == e["len"], (e["in"], e["order"])
    assert hashlib.md5(got).hexdigest() == e["md5"], (e["in"], e["order"])
    if "out" in e:
        assert got == base64.b64decode(e["out"])
    back = oracle.uncompress(got, capacity=len(data), out_size_hint=len(data))
    assert back == data


# tests/varint_test.c:145-155 exercises 7-bit boundaries; the encodings below are the
# big-endian 7-bit groups defined by varint.h:85-104.
VARINT_KAT = [
    (0, "00"), (1, "01"), (127, "7f"), (128, "8100"), (255, "817f"), (256, "8200"),
    (16383, "ff7f"), (16384, "818000"), (2097151, "ffff7f"), (2097152, "81808000"),
    (268435455, "ffffff7f"), (268435456, "8180808000"), (0x7FFFFFFF, "87ffffff7f"),
    (0xFFFFFFFF, "8fffffff7f"),
]


@pytest.mark.parametrize("value,hexstr", VARINT_KAT)
def test_varint_kat(oracle, value, hexstr):
    buf = (C.c_ubyte * 8)()
    n = oracle.lib.orc_var_put_u32(buf, C.c_uint(value))
    assert bytes(buf[:n]).hex() == hexstr
    out = C.c_uint(0)
    raw = (C.c_ubyte * 8)(*bytes.fromhex(hexstr))
    end = C.cast(C.addressof(raw) + len(hexstr) // 2, C.c_void_p)
    oracle.lib.orc_var_get_u32.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
    used = oracle.lib.orc_var_get_u32(C.addressof(raw), end, C.byref(out))
    assert used == len(hexstr) // 2 and out.value == value


def test_bound_values(oracle):
    # SURVEY.md §8(a1) [probe]
    assert oracle.bound(1 << 20, 0) == 1101802
    assert oracle.bound(1 << 20, 1) == 1299952
    assert oracle.bound(1 << 20, 193) == 1300730
    assert oracle.bound(1 << 16, 0) == 69610
    assert oracle.bound(1 << 16, 1) == 267760
    assert oracle.bound(1 << 16, 193) == 268538


def test_headline_sizes(oracle):
    # BASELINE.md §2 expected compressed sizes for 1 MiB tiles
    want = {("q4", 0): 80768, ("q8", 0): 236614, ("q40+dir", 0): 526965,
            ("q4", 1): 74990, ("q8", 1): 224009, ("q40+dir", 1): 507704,
            ("q4", 193): 67051, ("q8", 193): 216606, ("q40+dir", 193): 507704}
    for (name, order), size in want.items():
        got = oracle.compress(datagen.tile(name, 1 << 20).tobytes(), order)
        assert len(got) == size, (name, order)


def test_decode_rejects_garbage(oracle):
    rs = np.random.RandomState(5)
    assert oracle.uncompress(b"", capacity=10) is None
    for n in (1, 2, 5, 16, 40, 200):
        for _ in range(50):
            junk = rs.randint(0, 256, size=n).astype(np.uint8).tobytes()
            # must not crash; result may be None or bytes
            oracle.uncompress(junk, capacity=4096, out_size_hint=4096)
