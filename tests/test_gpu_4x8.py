"""rANS 4x8 on the GPU (include/rans4x8_hip.h) against the reference's fixtures and the oracle's
restatement (oracle/rans4x8_oracle.c, itself pinned to the fixtures and to the real rANS_static.c): bit-exact
both ways, through the drop-in symbols and the batch calls."""
import os

import numpy as np
import pytest

import datagen
from test_oracle4x8 import Codec8, _inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(datagen.GOLDEN, "r4x8")
FIXTURES = sorted(os.listdir(GOLD))


@pytest.fixture(scope="module")
def H():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import htscodecs_amd
    htscodecs_amd.load()
    return htscodecs_amd


@pytest.fixture(scope="module")
def orc8():
    import cpu_libs
    return Codec8(cpu_libs.oracle().lib, "orc8_")


@pytest.mark.parametrize("fn", FIXTURES)
def test_fixture_both_ways(H, fn):
    """tests/rans4x8.test: decode of the committed streams; and (stronger) the encoder reproduces them."""
    name, order = fn.rsplit(".", 1)
    with open(os.path.join(GOLD, fn), "rb") as f:
        comp = f.read()
    plain = datagen.base_text(name).tobytes()
    assert H.rans_uncompress(comp) == plain
    assert H.rans_compress(plain, int(order)) == comp


def test_random_differential_vs_oracle(H, orc8):
    rs = np.random.RandomState(4808)
    datas = _inputs(rs, 200) + [b"a", b"ab", b"abc", b"abcd", b"abcde", bytes(range(256)) * 5, bytes([0]) * 1000,
                                bytes(rs.randint(0, 256, size=70001).astype(np.uint8)),
                                datagen.tile("q40+dir", 1 << 20, 2).tobytes(), datagen.tile("q4", (1 << 20) + 3, 1).tobytes()]
    orders = [int(rs.randint(0, 2)) for _ in datas]
    enc, st = H.compress_batch_4x8(datas, orders)
    bad = [(len(d), o, s) for d, o, e, s in zip(datas, orders, enc, st) if e != orc8.compress(d, o)]
    assert not bad, bad[:10]
    dec, st = H.uncompress_batch_4x8(enc, [len(d) for d in datas])
    assert all(s == 0 for s in st), st
    assert dec == datas
    # empty input: refused (the reference divides by zero)
    enc, st = H.compress_batch_4x8([b""], [0])
    assert enc == [None] and st[0] != 0


def test_damaged_streams(H, orc8):
    """What the oracle rejects the device rejects; what both accept decodes identically; the device may refuse more
    (tables not listed in ascending order), never less."""
    rs = np.random.RandomState(8404)
    bads, caps, refs = [], [], []
    for d in _inputs(rs, 100):
        comp = bytearray(orc8.compress(d, int(rs.randint(0, 2))))
        for _ in range(5):
            bad = bytearray(comp)
            mode = rs.randint(0, 3)
            if mode == 0:
                bad[int(rs.randint(9, len(bad)))] ^= int(rs.randint(1, 256))
            elif mode == 1:
                bad = bad[:int(rs.randint(9, len(bad) + 1))]
                bad[1:5] = int(len(bad) - 9).to_bytes(4, "little")
            else:
                bad[int(rs.randint(0, len(bad)))] ^= 1 << int(rs.randint(0, 8))
            bads.append(bytes(bad))
            caps.append(len(d) + 64)
            refs.append(orc8.uncompress(bytes(bad)))
    for n in (0, 1, 8, 9, 26, 27, 100):
        for _ in range(10):
            bads.append(bytes(rs.randint(0, 256, size=n).astype(np.uint8)))
            caps.append(4096)
            refs.append(orc8.uncompress(bads[-1]))
    # streams whose size field exceeds the slot: CAPACITY, like the 4x16 calls
    dec, st = H.uncompress_batch_4x8(bads, caps)
    agree = 0
    for b, cap, x, s, r in zip(bads, caps, dec, st, refs):
        if r is None:
            assert x is None and s != 0
        elif len(r) > cap:
            assert x is None and s == 1
        elif x is not None:
            assert x == r
            agree += 1
        else:
            assert s in (3, 7)                 # TABLE (listing order, slot 4095) / CONTEXT
    assert agree > 150


def test_device_resident_4x8(H, orc8):
    """rans4x8_hip_{compress,uncompress}_dev on torch tensors: 64 blocks of 256 KiB, both orders, checked against the
    oracle byte for byte and round-tripped on the device."""
    import ctypes as C
    import torch
    from htscodecs_amd import codec
    L = H.load()
    ctx = codec._thread_ctx()
    dev = torch.device("cuda", 0)
    nblk, bs = 64, 1 << 18
    names = ["q4", "q8", "q40+dir", "qvar"]
    blocks = [datagen.tile(names[b % 4], bs, b) for b in range(nblk)]
    d_in = torch.from_numpy(np.concatenate(blocks)).to(dev)
    in_off = torch.arange(nblk, dtype=torch.int64, device=dev) * bs
    in_size = torch.full((nblk,), bs, dtype=torch.int32, device=dev)
    cap = L.rans4x8_hip_compress_bound(bs)
    slot = (cap + 255) // 256 * 256
    d_comp = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    comp_off = torch.arange(nblk, dtype=torch.int64, device=dev) * slot
    comp_cap = torch.full((nblk,), cap, dtype=torch.int32, device=dev)
    comp_size = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st = torch.full((nblk,), -1, dtype=torch.int32, device=dev)
    d_order = torch.tensor([b & 1 for b in range(nblk)], dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = L.rans4x8_hip_compress_dev(ctx.h, nblk, d_in.data_ptr(), in_off.data_ptr(), in_size.data_ptr(), d_comp.data_ptr(),
                                    comp_off.data_ptr(), comp_cap.data_ptr(), comp_size.data_ptr(), st.data_ptr(), 0,
                                    d_order.data_ptr(), bs, stream)
    assert rc == 0, ctx.error()
    d_back = torch.zeros_like(d_in)
    back_size = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st2 = torch.full((nblk,), -1, dtype=torch.int32, device=dev)
    rc = L.rans4x8_hip_uncompress_dev(ctx.h, nblk, d_comp.data_ptr(), comp_off.data_ptr(), comp_size.data_ptr(), d_back.data_ptr(),
                                      in_off.data_ptr(), in_size.data_ptr(), back_size.data_ptr(), st2.data_ptr(), stream)
    assert rc == 0, ctx.error()
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0 and int((st2 != 0).sum()) == 0
    assert torch.equal(d_back, d_in)
    csz = comp_size.cpu().numpy()
    comp = d_comp.cpu().numpy()
    for b in range(nblk):
        assert comp[b * slot:b * slot + csz[b]].tobytes() == orc8.compress(blocks[b].tobytes(), b & 1), b
