"""GPU tests added in round 3.  Everything goes through the C ABI (ctypes); the oracle is only the checker."""
import os
import sys

import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import htscodecs_amd
    htscodecs_amd.load()
    return htscodecs_amd


def _dev_roundtrip(H, name, nblk, bs, order):
    """nblk blocks through rans4x16_hip_{compress,uncompress}_dev; returns what the checks below need."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dc = H.DeviceCodec(0)
    dev = dc.dev
    d_in, in_off, in_size = bench.build_batch(torch, dev, name, nblk, bs, 0)
    cap = H.rans_compress_bound_4x16(bs, order)
    slot = (cap + 255) // 256 * 256
    d_comp = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    comp_off = torch.arange(nblk, dtype=torch.int64, device=dev) * slot
    comp_cap = torch.full((nblk,), cap, dtype=torch.int32, device=dev)
    comp_size = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st_enc = torch.full((nblk,), -1, dtype=torch.int32, device=dev)
    dc.compress(d_in, in_off, in_size, d_comp, comp_off, comp_cap, comp_size, st_enc, order, bs)
    d_back = torch.zeros_like(d_in)
    back_size = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st_dec = torch.full((nblk,), -1, dtype=torch.int32, device=dev)
    dc.uncompress(d_comp, comp_off, comp_size, d_back, in_off, in_size, back_size, st_dec, cap, bs if order & 0xc0 else 0)
    torch.cuda.synchronize()
    assert int((st_enc != 0).sum()) == 0 and int((st_dec != 0).sum()) == 0
    assert torch.equal(back_size, in_size)
    assert torch.equal(d_back, d_in)
    return bench, d_in, d_comp, comp_size.cpu().numpy(), slot


def test_two_rounds_of_full_size_blocks_byte_compared(H, oracle):
    """The headline shape walked for more than two rounds of the persistent chain kernels: 2 x (resident streams) + 7
    blocks of 1 MiB q40, order 1, through *_dev.  Every block round-trips on the device, and 200+ blocks - the first
    and the last of every round among them - are compared byte for byte with the oracle (a stream that is wrong but
    self-consistent in a later round would pass a round trip)."""
    dc = H.DeviceCodec(0)
    nsym = len(set(datagen.base_text("q40+dir").tolist()) | {0})
    spc, _, cus = dc.residency(True, nsym, 1, 10)
    res = spc * cus
    nblk, bs, order = 2 * res + 7, 1 << 20, 1
    bench, d_in, d_comp, csz, slot = _dev_roundtrip(H, "q40+dir", nblk, bs, order)
    rs = np.random.RandomState(33)
    edges = [0, 1, res - 1, res, res + 1, 2 * res - 1, 2 * res, 2 * res + 1, nblk - 2, nblk - 1]
    sample = sorted(set(edges + [int(x) for x in rs.randint(0, nblk, size=200)]))
    assert len(sample) >= 200
    for b in sample:
        raw = bench.block_bytes("q40+dir", bs, b, 0)
        got = d_comp[b * slot:b * slot + int(csz[b])].cpu().numpy().tobytes()
        assert got == oracle.compress(raw.tobytes(), order), b
    assert int(csz[0]) == 507704                             # the reference's size for this tile
