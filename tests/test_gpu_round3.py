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


def test_direct_rows_walk_several_rounds(H, oracle, opts):
    """The short-step ("direct", r4x16_common.h level 6) rows are meant for batches of at most one round of their
    resident streams; option dec_direct = 8 lets 2,100 q40 streams (1,024 are resident at four per CU) take them anyway,
    so that the persistent walk over several rounds is covered for this row kind too: every block round-trips, the
    streams are the oracle's."""
    opts.set("dec_direct", 8)
    opts.set("enc_direct", 8)
    nblk, bs, order = 2100, 65536, 1
    bench, d_in, d_comp, csz, slot = _dev_roundtrip(H, "q40+dir", nblk, bs, order)
    rs = np.random.RandomState(8)
    for b in sorted(set([0, 1023, 1024, 2047, 2048, nblk - 1] + [int(x) for x in rs.randint(0, nblk, size=40)])):
        raw = bench.block_bytes("q40+dir", bs, b, 0)
        got = d_comp[b * slot:b * slot + int(csz[b])].cpu().numpy().tobytes()
        assert got == oracle.compress(raw.tobytes(), order), b


@pytest.mark.parametrize("knob", [1, 0])
def test_small_batches_of_every_shape_both_row_kinds(H, oracle, opts, knob):
    """Batches far below one round of resident streams, with the short-step rows (knob 1, the default) and without:
    1, 3 and 64 blocks of 1 MiB and ragged sizes, q4 / q8 / q40, orders 0, 1, 65, 193, device-resident; the compressed
    bytes are the oracle's for every block and every block round-trips."""
    import torch
    opts.set("dec_direct", knob)
    opts.set("enc_direct", knob)
    dc = H.DeviceCodec(0)
    dev = dc.dev
    for nblk, sizes in ((1, [1 << 20]), (3, [1 << 20, 777777, 5]), (64, [65536, 40001, 3, 131072])):
        for name, order in (("q40+dir", 1), ("q40+dir", 0), ("q8", 1), ("q4", 193), ("q8", 65), ("q4", 1)):
            blocks = [datagen.tile(name, sizes[b % len(sizes)], b) for b in range(nblk)]
            in_off = np.cumsum([0] + [(len(b) + 255) // 256 * 256 for b in blocks])[:-1].astype(np.int64)
            in_size = np.array([len(b) for b in blocks], dtype=np.int32)
            arena = np.zeros(int(in_off[-1]) + ((len(blocks[-1]) + 255) // 256 * 256), dtype=np.uint8)
            for b, off in zip(blocks, in_off):
                arena[off:off + len(b)] = b
            caps = np.array([H.rans_compress_bound_4x16(len(b), order) for b in blocks], dtype=np.int32)
            out_off = np.cumsum([0] + [(int(c) + 255) // 256 * 256 for c in caps])[:-1].astype(np.int64)
            t = lambda a: torch.from_numpy(a).to(dev)
            d_in = t(arena)
            d_out = torch.zeros(int(out_off[-1]) + int(caps[-1]) + 256, dtype=torch.uint8, device=dev)
            d_in_off, d_in_size, d_out_off, d_caps = t(in_off), t(in_size), t(out_off), t(caps)
            d_osz = torch.zeros(nblk, dtype=torch.int32, device=dev)
            d_st = torch.full((nblk,), -1, dtype=torch.int32, device=dev)
            dc.compress(d_in, d_in_off, d_in_size, d_out, d_out_off, d_caps, d_osz, d_st, order, int(in_size.max()))
            torch.cuda.synchronize()
            assert (d_st == 0).all(), (name, order, d_st.tolist())
            osz = d_osz.cpu().numpy()
            comp = d_out.cpu().numpy()
            for i, b in enumerate(blocks):
                assert comp[out_off[i]:out_off[i] + osz[i]].tobytes() == oracle.compress(b.tobytes(), order), (name, order, nblk, i)
            d_dec = torch.zeros_like(d_in)
            d_dsz = torch.zeros(nblk, dtype=torch.int32, device=dev)
            d_dst = torch.full((nblk,), -1, dtype=torch.int32, device=dev)
            dc.uncompress(d_out, d_out_off, d_osz, d_dec, d_in_off, t(in_size.copy()), d_dsz, d_dst, int(osz.max()), int(in_size.max()))
            torch.cuda.synchronize()
            assert (d_dst == 0).all(), (name, order, d_dst.tolist())
            assert (d_dsz.cpu().numpy() == in_size).all()
            dec = d_dec.cpu().numpy()
            for b, off in zip(blocks, in_off):
                assert (dec[off:off + len(b)] == b).all(), (name, order, nblk)


def test_direct_rows_alphabet_shapes(H, oracle, opts):
    """The short-step rows' special cases, each against the oracle both ways through the batch calls (few blocks: the
    default budget gives every stream its direct rows):
      * many symbols ONE slot wide (a dominant symbol and a hundred rare ones): slot pairs shared by two symbols, ranks
        that step by one from pair to pair;
      * affine alphabets (byte = index + c) with and without byte 0 in the data, and alphabets that are not affine
        (gaps; byte 0 used beside an offset run), which keep the per-symbol alpha[] read;
      * exactly 128 symbols (the largest alphabet the 7-bit index field serves) and 129 (falls back to the other rows);
      * 12-bit order-1 tables (no room for the empty-row flag: never affine)."""
    rs = np.random.RandomState(77)
    datas, orders = [], []
    def add(a, both=True):
        for o in ((0, 1) if both else (1,)):
            datas.append(a.tobytes()); orders.append(o)
    n = 70000
    add(datagen.weighted(n, [5000] + [1] * 100, 5))                        # one-slot symbols
    add(datagen.weighted(n, [3000] + [2] * 60 + [1] * 60, 6))
    add((datagen.weighted(n, [400] + [1] * 40, 7).astype(np.int32) + 33).astype(np.uint8))   # offset run, no byte 0
    a = (datagen.weighted(n, [50] * 20, 8).astype(np.int32) + 40).astype(np.uint8)
    a[::97] = 0                                                            # byte 0 used beside an offset run: not affine
    add(a)
    add(datagen.weighted(n, [9] * 30, 9))                                   # contiguous from 0: affine with c = 0
    add((datagen.weighted(n, [7] * 25, 10).astype(np.int32) * 3 + 10).astype(np.uint8))     # gaps: not affine
    add(datagen.rand(n, 11, 128, 0))
    add(datagen.rand(n, 12, 128, 100))
    add(datagen.rand(n, 13, 129, 0))
    add(datagen.weighted(1 << 18, [30000] + [1] * 127, 3), both=False)      # shift 12 (SURVEY 8c's edge case, smaller alphabet)
    add(datagen.weighted(1 << 18, [100000, 50000] + [1] * 100, 3), both=False)
    add(datagen.tile("q40+dir", 100001, 3))
    add(datagen.tile("q8", 99999, 4))
    for knob in (1, 0):
        opts.set("dec_direct", knob)
        opts.set("enc_direct", knob)
        enc, st = H.compress_batch(datas, orders)
        assert all(s == 0 for s in st), st
        want = [oracle.compress(d, o) for d, o in zip(datas, orders)]
        bad = [(i, orders[i]) for i in range(len(datas)) if enc[i] != want[i]]
        assert not bad, (knob, bad)
        dec, st = H.uncompress_batch(want, [len(d) for d in datas])
        assert all(s == 0 for s in st), (knob, st)
        bad = [(i, orders[i]) for i in range(len(datas)) if dec[i] != datas[i]]
        assert not bad, (knob, bad)
    def shift_of(c):                                                        # flags, varint size, then shift << 4 | compressed
        i = 1
        while c[i] & 0x80:
            i += 1
        return c[i + 1] >> 4
    assert sum(1 for w in want if (w[0] & 1) and shift_of(w) == 12) >= 2, "no 12-bit order-1 table among the cases"


def test_combiner_keeps_callers_apart(oracle):
    """ADVICE (round 2): the combiner behind the five drop-in symbols serves unrelated callers with one batch - a bad or
    greedy request must not fail its neighbours.  A pool of 24 threads (fresh process: the combiner reads its knobs once)
    mixes valid blocks with garbage streams, truncated streams and streams whose size field claims 1.5 GB into a small
    caller buffer, with the batch byte cap at 1 MB so that batches split: every valid block decodes to its input, every
    bad one returns None, nobody hangs."""
    import subprocess, textwrap
    code = textwrap.dedent('''
        import sys, os, threading
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np, datagen, cpu_libs
        import htscodecs_amd as H
        orc = cpu_libs.oracle()
        good = [datagen.tile(["q4", "q8", "q40+dir"][i %% 3], 30000 + 997 * i, i).tobytes() for i in range(48)]
        comp = [orc.compress(d, (0, 1, 65, 193)[i %% 4]) for i, d in enumerate(good)]
        rs = np.random.RandomState(5)
        jobs = []
        for i, (d, c) in enumerate(zip(good, comp)):
            jobs.append(("ok", c, len(d), d))
            if i %% 3 == 0: jobs.append(("bad", rs.bytes(200 + i), 4096, None))
            if i %% 5 == 0: jobs.append(("bad", c[:len(c) // 2], len(d), None))
            if i %% 7 == 0: jobs.append(("bad", bytes([0]) + bytes([0x85, 0xcb, 0xa5, 0xe0, 0x00]) + c[4:], 1000, None))   # claims ~1.5 GB
        res = [None] * len(jobs)
        def work(k):
            for j in range(k, len(jobs), 24):
                kind, c, cap, want = jobs[j]
                res[j] = H.rans_uncompress_4x16(c, cap)
        th = [threading.Thread(target=work, args=(k,)) for k in range(24)]
        [t.start() for t in th]; [t.join(120) for t in th]
        assert not any(t.is_alive() for t in th), "a caller hangs"
        for (kind, c, cap, want), r in zip(jobs, res):
            if kind == "ok": assert r == want, "a valid block failed beside a bad one"
            else: assert r is None or r == orc.uncompress(c, cap), "a bad block did not fail like the oracle"
        # and the encode side: valid inputs beside an empty one
        enc = [None] * 48
        def work2(k):
            for j in range(k, 48, 24): enc[j] = H.rans_compress_4x16(good[j], (0, 1, 65, 193)[j %% 4])
        th = [threading.Thread(target=work2, args=(k,)) for k in range(24)]
        [t.start() for t in th]; [t.join(120) for t in th]
        assert enc == comp
        print("isolated ok")
    ''') % (ROOT, ROOT)
    env = dict(os.environ, R4X16_COMBINE_MAX_MB="1", R4X16_COMBINE_MAX="8")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "isolated ok" in r.stdout, r.stdout + r.stderr


def _runs(rs, n_runs, lens, syms):
    """n_runs runs: lengths drawn from `lens`, symbols from `syms` (neighbours differ)."""
    out, prev = [], -1
    for _ in range(n_runs):
        s = int(syms[rs.randint(len(syms))])
        if s == prev:
            s = int(syms[(list(syms).index(s) + 1) % len(syms)])
        out.append(np.full(int(lens[rs.randint(len(lens))]), s, dtype=np.uint8))
        prev = s
    return np.concatenate(out)


@pytest.mark.parametrize("route", [0, 99999], ids=["one-wave", "workgroup"])
def test_run_length_shapes(H, oracle, opts, route):
    """X_RLE both ways against the oracle, through both expansion kernels (option back_wg_per_cu picks per call):
      * run lengths around the varint sizes (127 / 128, 16,383 / 16,384, 2^21) and long stretches of two- and
        three-byte varints (the one-wave route decodes the run stream 60 bytes at a time with four bytes of history);
      * literal counts around the trip sizes (63 .. 65, 255 .. 257, 1,023 .. 1,025) and a run at the very end;
      * runs around 2^22 bytes (where a trip leaves its 32-bit sums for the plain route) and one of more than 2^24 in
        a 20 MiB block;
      * X_PACK in front (order 192 / 193) and quality-like data."""
    opts.set("back_wg_per_cu", route)
    rs = np.random.RandomState(4242)
    datas, orders = [], []
    def add(a, os_=(64, 65)):
        for o in os_:
            datas.append(np.ascontiguousarray(a).tobytes()); orders.append(o)
    add(_runs(rs, 3000, [1, 2, 3, 126, 127, 128, 129, 130], [10, 11, 12, 13, 200]))
    add(_runs(rs, 400, [16382, 16383, 16384, 16385, 300, 5000], [1, 2, 3]))
    add(_runs(rs, 2000, list(range(129, 400)), [7, 8, 9, 10, 11, 12]))              # two-byte varints back to back
    add(_runs(rs, 40, [1 << 21, (1 << 21) + 1, 70000], [5, 6]), os_=(64,))          # four-byte varints
    add(_runs(rs, 6, [(1 << 22) - 1, 1 << 22, (1 << 22) + 1], [5, 6]), os_=(64,))   # either side of the plain route's threshold
    for nlit in (63, 64, 65, 255, 256, 257, 1023, 1024, 1025):
        a = _runs(rs, nlit, [1, 1, 1, 2, 5, 40], [20, 21, 22, 23, 24, 25, 26, 27])
        add(a, os_=(65,))
        add(np.concatenate([a, np.full(777, 99, np.uint8)]), os_=(64,))             # ... and a run at the very end
    giant = np.concatenate([_runs(rs, 500, [1, 2, 3, 9], [1, 2, 3, 4]), np.full((1 << 24) + 1234567, 7, np.uint8),
                            _runs(rs, 500, [1, 2, 3, 9], [1, 2, 3, 4])])
    add(giant, os_=(64, 193))
    add(_runs(rs, 50000, [1, 1, 2, 3, 4, 6, 9, 30], [0, 1, 2, 3]), os_=(192, 193))   # four symbols: X_PACK packs four per byte
    add(datagen.tile("q4", 300001, 5), os_=(193, 65))
    add(datagen.tile("q8", 299999, 6), os_=(65, 64))
    enc, st = H.compress_batch(datas, orders)
    assert all(s == 0 for s in st), st
    want = [oracle.compress(d, o) for d, o in zip(datas, orders)]
    bad = [(i, orders[i], len(datas[i])) for i in range(len(datas)) if enc[i] != want[i]]
    assert not bad, bad
    assert sum(1 for w in want if w[0] & 64) >= len(want) - 4, "X_RLE was dropped by the encoder for most cases"
    dec, st = H.uncompress_batch(want, [len(d) for d in datas])
    assert all(s == 0 for s in st), st
    bad = [(i, orders[i], len(datas[i])) for i in range(len(datas)) if dec[i] != datas[i]]
    assert not bad, bad


def test_pack_widths(H, oracle):
    """X_PACK both ways against the oracle for every code width (1, 2, 4 bits and the copy case), alphabets at the edges
    of each (2, 3, 4, 5, 8, 16, 17 symbols), lengths around the sixteen-byte trips of the unpacking loops, with and
    without X_RLE and order 1 behind it."""
    rs = np.random.RandomState(99)
    datas, orders = [], []
    for nsym in (1, 2, 3, 4, 5, 8, 15, 16, 17):
        syms = rs.choice(256, nsym, replace=False).astype(np.uint8)
        for n in (15, 16, 17, 31, 33, 4095, 4096, 4097, 70001):
            a = syms[rs.randint(0, nsym, n)]
            for o in (128, 129, 192, 193):
                datas.append(a.tobytes()); orders.append(o)
    for name in ("q8", "q4"):
        for o in (128, 129, 193):
            datas.append(np.ascontiguousarray(datagen.tile(name, 300003, 7)).tobytes()); orders.append(o)
    enc, st = H.compress_batch(datas, orders)
    assert all(s == 0 for s in st), st
    want = [oracle.compress(d, o) for d, o in zip(datas, orders)]
    bad = [(i, orders[i], len(datas[i])) for i in range(len(datas)) if enc[i] != want[i]]
    assert not bad, bad[:8]
    dec, st = H.uncompress_batch(want, [len(d) for d in datas])
    assert all(s == 0 for s in st), st
    bad = [(i, orders[i], len(datas[i])) for i in range(len(datas)) if dec[i] != datas[i]]
    assert not bad, bad[:8]
