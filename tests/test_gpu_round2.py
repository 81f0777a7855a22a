"""GPU tests added in round 2: the reference driver's calling pattern, arbitrary bytes through the malloc'ing
entry point, an oversized block inside a device batch, the multi-device calls, the per-GPU share of
BASELINE.json configs[4], and bench.py's own N-rank launch."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# Every test of this module runs twice: with the short-step ("direct") rows that small batches take by default, and with
# them switched off, so that the compressed rows (u16 search trees, packed 10/11-bit rows) stay covered by the same cases.
@pytest.fixture(scope="module", params=["short-step", "compressed-rows"])
def H(request):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import htscodecs_amd
    htscodecs_amd.load()
    from conftest import _Options
    o = _Options()
    for k in ("dec_direct", "enc_direct"):
        o.set(k, 1 if request.param == "short-step" else 0)
    yield htscodecs_amd
    o.restore()


ALL_ORDERS = [0, 1, 64, 65, 128, 129, 192, 193, 8, 9, 0x48, 0xc9, (2 << 8) | 9, 16 | 1, 32]


def test_in_size_larger_than_the_stream(H, oracle):
    """tests/rANS_static4x16pr_test.c:206 hands rans_uncompress_to_4x16 the CAPACITY of the compressed buffer
    as in_size: the stream is followed by slack.  Every flag set (stripe included) must decode to the same
    bytes as with the exact length - checked against the oracle given the same oversized in_size - through
    the single-block symbol and through the batch call."""
    rs = np.random.RandomState(206)
    names = ["q4", "q8", "q40+dir", "qvar"]
    comps, datas = [], []
    for k, order in enumerate(ALL_ORDERS * 2):
        n = int(rs.choice([37, 1000, 30011, 70000]))
        d = datagen.tile(names[k % 4], n, k).tobytes()
        c = oracle.compress(d, order)
        slack = int(rs.choice([1, 2, 7, 64, 1000, len(c)]))
        fill = bytes(rs.randint(0, 256, size=slack).astype(np.uint8)) if k % 2 else b"\0" * slack
        comps.append(c + fill)
        datas.append(d)
    for c, d in zip(comps, datas):
        want = oracle.uncompress(c, capacity=len(d), out_size_hint=len(d))
        assert want == d                                   # the oracle (pinned to the reference) accepts slack
        if not (c[0] & 0x10):
            assert H.rans_uncompress_4x16(c) == d          # out == NULL
        assert H.rans_uncompress_4x16(c, len(d)) == d      # caller buffer
    dec, st = H.uncompress_batch(comps, [len(d) for d in datas])
    assert all(s == 0 for s in st), st
    assert dec == datas


def test_arbitrary_bytes_through_the_malloc_entry(H, oracle):
    """tests/rANS_static4x16pr_fuzz.c:70-76 feeds arbitrary bytes to rans_uncompress_4x16 (out == NULL: the callee
    sizes its buffer from the stream's own varint).  Outcome per input must match the oracle: rejected there ->
    rejected here; accepted by both -> same bytes.  Includes hostile size fields (up to 2^31 - 1) on tiny inputs."""
    rs = np.random.RandomState(7076)
    junk = []
    for n in (1, 2, 3, 5, 8, 16, 17, 40, 100, 300, 2000):
        for _ in range(40):
            junk.append(bytes(rs.randint(0, 256, size=n).astype(np.uint8)))
    # plausible headers in front of random payloads: every flag combination, small stored sizes
    for _ in range(600):
        flags = int(rs.randint(0, 256)) & ~0x08
        n = int(rs.randint(1, 400))
        body = bytes(rs.randint(0, 256, size=int(rs.randint(0, 120))).astype(np.uint8))
        junk.append(bytes([flags]) + (bytes([n]) if n < 128 else bytes([0x80 | (n >> 7), n & 0x7f])) + body)
    # valid streams with a few damaged bytes (these reach the chain kernels)
    for k in range(200):
        d = datagen.tile(["q4", "q8", "q40+dir"][k % 3], int(rs.randint(50, 3000)), k).tobytes()
        c = bytearray(oracle.compress(d, int(rs.choice([0, 1, 65, 129, 193]))))
        for _ in range(int(rs.randint(0, 3))):
            c[int(rs.randint(1, len(c)))] ^= 1 << int(rs.randint(0, 8))
        if not c[0] & 0x08:
            junk.append(bytes(c))
    hostile = [bytes([0x00, 0x87, 0xff, 0xff, 0xff, 0x7e]) + bytes(24),          # order 0, claims 2^31 - 2 bytes
               bytes([0x01, 0x87, 0xff, 0xff, 0xff, 0x7e]) + bytes(40),
               bytes([0xc0, 0x84, 0x80, 0x80, 0x80, 0x00]) + bytes(30),          # PACK|RLE, claims 1 GiB
               bytes([0x20, 0x87, 0xff, 0xff, 0xff, 0x7e]) + b"abc"]             # CAT longer than the input
    agree = 0
    for j in junk + hostile:
        want = oracle.uncompress_malloc(j)
        got = H.rans_uncompress_4x16(j)
        if want is None:
            assert got is None, j[:16].hex()
        elif got is not None:
            assert got == want, j[:16].hex()
            agree += 1
    assert agree > 100
    # the same inputs as one batch, capacities from the streams' own size fields (bounded for the test)
    import cpu_libs
    small = [j for j in junk if cpu_libs.peek_ulen(j) <= 1 << 16 and not (j[0] & 0x10)]
    caps = [cpu_libs.peek_ulen(j) for j in small]
    dec, st = H.uncompress_batch(small, caps)
    for j, cap, x, s in zip(small, caps, dec, st):
        want = oracle.uncompress(j, capacity=cap, out_size_hint=cap)
        if want is None:
            assert x is None and s != 0
        elif x is not None:
            assert x == want
        else:
            assert s in (6, 7, 8)
    # a hostile size field must not leave gigabytes pinned to the calling thread
    from htscodecs_amd import codec
    L = H.load()
    assert H.rans_uncompress_4x16(hostile[0]) is None
    assert H.rans_compress_4x16(b"abcd" * 100, 1) == oracle.compress(b"abcd" * 100, 1)


def test_oversized_block_in_a_device_batch_fails_alone(H, oracle):
    """rans4x16_hip_compress_dev sizes its workspace from max_in_size (or total_in_size) - a caller may lie.  Since round 4
    the payload is written into the caller's own bound-sized slot and nothing per block depends on max_in_size any more
    for the plain orders: a block larger than announced simply encodes.  With X_PACK / X_RLE the staging regions are laid
    out on the device from the real sizes inside an area sized from the announced ones: a block whose region would end
    beyond it must be refused (UNSUPPORTED) and every other block must carry the reference's bytes (an
    oversized block used to write backwards past the start of its scratch, into its neighbour's)."""
    import torch
    dc = H.DeviceCodec(0)
    dev = dc.dev
    sizes = [20000, 20000, 3000000, 20000, 20000, 20000]
    blocks = [datagen.tile("q8", s, i) for i, s in enumerate(sizes)]
    for order in (1, 0, 65, 193):
        in_off = np.cumsum([0] + [(s + 255) // 256 * 256 for s in sizes])[:-1].astype(np.int64)
        arena = np.zeros(int(in_off[-1]) + sizes[-1] + 256, dtype=np.uint8)
        for b, off in zip(blocks, in_off):
            arena[off:off + len(b)] = b
        caps = np.array([H.rans_compress_bound_4x16(s, order) for s in sizes], dtype=np.int32)
        out_off = np.cumsum([0] + [(int(c) + 255) // 256 * 256 for c in caps])[:-1].astype(np.int64)
        t = lambda a: torch.from_numpy(a).to(dev)
        d_in = t(arena)
        d_out = torch.zeros(int(out_off[-1]) + int(caps[-1]) + 256, dtype=torch.uint8, device=dev)
        d_osz = torch.zeros(len(sizes), dtype=torch.int32, device=dev)
        d_st = torch.full((len(sizes),), -1, dtype=torch.int32, device=dev)
        # the caller lies: max_in_size = 20000 although block 2 has three million bytes
        dc.compress(d_in, t(in_off), t(np.array(sizes, dtype=np.int32)), d_out, t(out_off), t(caps), d_osz, d_st,
                    order, 20000)
        torch.cuda.synchronize()
        st, osz, comp = d_st.cpu().numpy(), d_osz.cpu().numpy(), d_out.cpu().numpy()
        if order in (0, 1):
            assert st.tolist() == [0] * 6, (order, st.tolist())
        else:
            assert st[2] == 6 and osz[2] == 0, (order, st.tolist())          # 6 x 20,000 bytes announced: three million do not fit
            assert st[0] == 0 and st[1] == 0, (order, st.tolist())
        for i in range(6):
            assert st[i] in (0, 6), (order, st.tolist())
            if st[i] == 0:
                assert comp[out_off[i]:out_off[i] + osz[i]].tobytes() == oracle.compress(blocks[i].tobytes(), order), (i, order)
            else:
                assert osz[i] == 0


def test_multi_device_batch_calls(H, oracle):
    """rans4x16_hip_{compress,uncompress}_batch_multi: the library partitions the batch, one host thread and
    context per listed device (the one card of this box listed twice and three times), sizes and statuses land
    in block order - also for blocks that fail."""
    rs = np.random.RandomState(808)
    names = ["q4", "q8", "q40+dir"]
    datas = [datagen.tile(names[b % 3], int(rs.choice([300, 5000, 65536, 200000])), b).tobytes() for b in range(150)]
    datas += [b"", b"z"]
    orders = [int(rs.choice([0, 1, 65, 193, 9])) for _ in datas]
    want = [oracle.compress(d, o) for d, o in zip(datas, orders)]
    for devs in ([0, 0], [0, 0, 0], [0]):
        mc = H.MultiCodec(devs)
        assert mc.devices() == len(devs)
        enc, st = mc.compress_batch(datas, orders)
        assert all(s == 0 for s in st)
        assert enc == want
        comps = list(want)
        broken = [5, 77, 149]
        for i in broken:
            comps[i] = comps[i][:max(1, len(comps[i]) // 3)]
        comps[77] = b"\x01" + comps[77][1:9]          # order-1 flag, 8 bytes of payload: TRUNCATED for certain
        dec, st = mc.uncompress_batch(comps, [len(d) for d in datas])
        for i, (d, x, s) in enumerate(zip(datas, dec, st)):
            if i in broken:
                # a truncated stream may still decode (rANS_word.h:402-410 stops refilling at the end of the
                # input): whatever the oracle says for this input is the expected outcome
                ref = oracle.uncompress(comps[i], capacity=len(d), out_size_hint=len(d))
                assert x == ref, i
                assert (s != 0) == (ref is None)
            else:
                assert s == 0 and x == d, i


def test_configs4_share_device_resident(H, oracle):
    """BASELINE.json configs[4], one GPU's share: 32,768 blocks of 64 KiB cycling q4 / q8 / q40 by b mod 3, order 1,
    through rans4x16_hip_{compress,uncompress}_dev.  Every block must round-trip (checked on the device) and a
    sample of 120 compressed blocks is compared byte-for-byte with the oracle."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dc = H.DeviceCodec(0)
    dev = dc.dev
    nblk, bs, order = 32768, 65536, 1
    d_in, in_off, in_size = bench.build_batch(torch, dev, "mixed", nblk, bs, 0)
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
    dc.uncompress(d_comp, comp_off, comp_size, d_back, in_off, in_size, back_size, st_dec, cap, bs)
    torch.cuda.synchronize()
    assert int((st_enc != 0).sum()) == 0 and int((st_dec != 0).sum()) == 0
    assert torch.equal(back_size, in_size)
    assert torch.equal(d_back, d_in)
    csz = comp_size.cpu().numpy()
    rs = np.random.RandomState(4)
    for b in sorted(set([0, 1, 2, nblk - 1] + [int(x) for x in rs.randint(0, nblk, size=120)])):
        raw = bench.block_bytes("mixed", bs, b, 0)
        assert (d_in[b * bs:(b + 1) * bs].cpu().numpy() == raw).all(), b
        got = d_comp[b * slot:b * slot + int(csz[b])].cpu().numpy().tobytes()
        assert got == oracle.compress(raw.tobytes(), order), b


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the script spawns two ranks itself (before touching the GPU);
    on this one-GPU box both share the card over gloo (R4X16_OVERSUBSCRIBE=1).  One JSON line, n_gpus == 2."""
    env = dict(os.environ, R4X16_OVERSUBSCRIBE="1")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--blocks", "600",
                        "--steps", "2", "--warmup", "1", "--no-cpu", "--no-host", "--hetero-gib", "0.25"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["blocks_per_gpu"] == 600
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0
    # every rank's step time, and the heterogeneous leg cut by the library's weighted partition
    assert len(out["per_rank_ms_per_step"]["each"]) == 2
    het = out["hetero"]
    assert het["hetero"]["roundtrip_ok"] and het["uniform_64KiB"]["roundtrip_ok"]
    assert len(het["per_rank"]["bytes"]) == 2 and het["per_rank"]["max_share_over_mean"] <= 1.05


def test_single_block_calls_from_a_thread_pool_are_combined(H, oracle):
    """The literal drop-in case: 24 host threads each calling rans_compress_to_4x16 /
    rans_uncompress_to_4x16 block by block.  Calls that arrive together are served by one device batch
    (r4x16_api.hip: Combiner); every caller must get its own, reference-identical result - mixed sizes, orders,
    stripe blocks, and one caller whose stream is damaged."""
    from concurrent.futures import ThreadPoolExecutor
    rs = np.random.RandomState(2424)
    names = ["q4", "q8", "q40+dir", "qvar"]
    jobs = []
    for t in range(24):
        blocks = [datagen.tile(names[(t + k) % 4], int(rs.choice([100, 5000, 65536, 300000])), t * 7 + k).tobytes() for k in range(5)]
        orders = [int(rs.choice([0, 1, 65, 193, 9])) for _ in blocks]
        jobs.append((blocks, orders))

    def work(job):
        blocks, orders = job
        for d, o in zip(blocks, orders):
            c = H.rans_compress_4x16(d, o)
            if c != oracle.compress(d, o):
                return False
            if H.rans_uncompress_4x16(c) != d:
                return False
            if H.rans_uncompress_4x16(c[:max(1, len(c) // 2)], len(d)) != oracle.uncompress(c[:max(1, len(c) // 2)], capacity=len(d), out_size_hint=len(d)):
                return False
        return True

    with ThreadPoolExecutor(24) as ex:
        assert all(ex.map(work, jobs))


def test_alphabet_that_grows_after_the_first_64_kib(H, oracle):
    """k_enc_front sizes the pair counters of large order-1 blocks from the alphabet of their first 64 KiB plus one
    overflow symbol and falls back to the exact two-pass route when that symbol is ever hit: blocks whose later
    bytes bring new symbols (at the very end, in the middle, as a quarter start, as the last byte) must still be
    reference-identical; so must blocks where nothing new appears."""
    rs = np.random.RandomState(6464)
    n = 600000
    base = datagen.tile("q40+dir", n, 3)
    cases = []
    a = base.copy(); cases.append(a)                                      # nothing new
    a = base.copy(); a[-1] = 200; cases.append(a)                         # new symbol as the last byte
    a = base.copy(); a[300000:300050] = 7; cases.append(a)                # new symbol in the middle
    a = base.copy(); a[2 * (n >> 2)] = 250; cases.append(a)               # new symbol exactly at a quarter start
    a = base.copy(); a[70000:] = rs.randint(0, 256, size=n - 70000).astype(np.uint8); cases.append(a)   # everything new
    a = base.copy(); a[65536] = 1; cases.append(a)                        # first byte after the sample
    a = np.concatenate([datagen.tile("q4", 100000, 1), datagen.tile("q40+dir", n - 100000, 2)]); cases.append(a)
    datas = [c.tobytes() for c in cases]
    for order in (1, 193, 65):
        enc, st = H.compress_batch(datas, [order] * len(datas))
        assert all(s == 0 for s in st)
        for k, (d, e) in enumerate(zip(datas, enc)):
            assert e == oracle.compress(d, order), (k, order)


def test_pack_symbol_set_that_grows_after_the_head(H, oracle):
    """X_PACK takes its symbol set from the first 64 KiB of blocks of 256 KiB and more and repeats the step on the
    whole block when the packing pass meets a byte from outside it; order-1 blocks from 64 KiB up take their alphabet
    from their first quarter.  New symbols late in the block - keeping the packing width, changing it, ending packing
    altogether, after a constant head - must give reference-identical streams."""
    n = 600000
    q4 = datagen.tile("q4", n, 5)
    vals = sorted(set(q4[:70000].tolist()))
    fresh = [v for v in range(256) if v not in vals]
    cases = []
    a = q4.copy(); cases.append(a)                                        # nothing new
    a = q4.copy(); a[-1] = fresh[0]; cases.append(a)                      # a fifth symbol as the last byte: 4 -> 2 per byte
    a = q4.copy(); a[65536] = fresh[1]; cases.append(a)                   # first byte after the head
    a = q4.copy(); a[400000:400020] = np.array(fresh[:20], dtype=np.uint8); cases.append(a)   # more than sixteen: no packing
    a = q4.copy(); a[:70000] = vals[0]; cases.append(a)                   # constant head
    a = np.full(n, vals[1], dtype=np.uint8); a[-3] = vals[0]; cases.append(a)     # constant but for one late byte
    a = np.full(n, vals[1], dtype=np.uint8); cases.append(a)              # constant
    for m in (65536, 100000, 262143):                                     # quarter-sampled order-1 alphabets
        a = datagen.tile("q40+dir", m, 2); cases.append(a.copy())
        a = a.copy(); a[m // 4 + 16] = 201; cases.append(a)
        a = a.copy(); a[-1] = 202; cases.append(a)
    datas = [c.tobytes() for c in cases]
    for order in (193, 129, 128, 1):
        enc, st = H.compress_batch(datas, [order] * len(datas))
        assert all(s == 0 for s in st)
        for k, (d, e) in enumerate(zip(datas, enc)):
            assert e == oracle.compress(d, order), (k, order)
        dec, st2 = H.uncompress_batch(enc, [len(d) for d in datas])
        assert all(s == 0 for s in st2) and dec == datas


def test_stripe_blocks_device_resident(H, oracle):
    """X_STRIPE through rans4x16_hip_{compress,uncompress}_dev (r4x16_stripe.hip): planes, the N x K candidate
    encodings, the per-plane arg-min and the header on the device (rANS_static4x16pr.c:1154-1216); decode with the
    planes-per-block reservation.  The four stripe fixtures and the stripe edge shapes, bit-exact with the oracle;
    ordinary blocks ride along in the same decode batch; blocks of <= 20 bytes drop the flag (:1151)."""
    import torch
    dc = H.DeviceCodec(0)
    dev = dc.dev
    L = H.load()
    names = ["q4", "q8", "q40+dir", "qvar"]
    rs = np.random.RandomState(89)
    blocks = [datagen.base_text("q4").tobytes(), datagen.base_text("q40+dir").tobytes(), b"", b"a", b"abcdefghij" * 2, b"x" * 21,
              datagen.tile("q8", 1000, 1).tobytes(), datagen.tile("qvar", 65537, 2).tobytes(), datagen.tile("q40+dir", 300003, 5).tobytes()]
    blocks += [datagen.tile(names[k % 4], int(rs.randint(22, 5000)), k).tobytes() for k in range(30)]
    n = len(blocks)
    sizes = [len(b) for b in blocks]
    in_off = np.cumsum([0] + [(s + 255) // 256 * 256 + 256 for s in sizes])[:-1].astype(np.int64)
    arena = np.zeros(int(in_off[-1]) + sizes[-1] + 512, dtype=np.uint8)
    for b, off in zip(blocks, in_off):
        arena[off:off + len(b)] = np.frombuffer(b, dtype=np.uint8)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_in = t(arena)
    for order in (8, 9, 0x48, 0xc9, (2 << 8) | 9, (3 << 8) | 0xc9, (7 << 8) | 8):
        want = [oracle.compress(b, order) for b in blocks]
        caps = np.array([H.rans_compress_bound_4x16(s, order) for s in sizes], dtype=np.int32)
        out_off = np.cumsum([0] + [(int(c) + 255) // 256 * 256 for c in caps])[:-1].astype(np.int64)
        d_out = torch.zeros(int(out_off[-1]) + int(caps[-1]) + 256, dtype=torch.uint8, device=dev)
        d_osz = torch.zeros(n, dtype=torch.int32, device=dev)
        d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
        dc.compress(d_in, t(in_off), t(np.array(sizes, dtype=np.int32)), d_out, t(out_off), t(caps), d_osz, d_st, order, max(sizes))
        torch.cuda.synchronize()
        st, osz, comp = d_st.cpu().numpy(), d_osz.cpu().numpy(), d_out.cpu().numpy()
        assert (st == 0).all(), (order, st.tolist())
        for i in range(n):
            assert comp[out_off[i]:out_off[i] + osz[i]].tobytes() == want[i], (order, i, sizes[i])
        # decode on the device: the stripe streams just made, with two ordinary streams mixed in
        N = (order >> 8) or 4
        assert L.rans4x16_hip_set_dev_stripe_planes(dc.ctx.h, N, max(sizes)) == 0
        plain = [oracle.compress(blocks[6], 1), oracle.compress(blocks[8], 193)]
        streams = want + plain
        plains = blocks + [blocks[6], blocks[8]]
        m = len(streams)
        c_off = np.cumsum([0] + [(len(s) + 255) // 256 * 256 + 256 for s in streams])[:-1].astype(np.int64)
        carena = np.zeros(int(c_off[-1]) + len(streams[-1]) + 512, dtype=np.uint8)
        for sdata, off in zip(streams, c_off):
            carena[off:off + len(sdata)] = np.frombuffer(sdata, dtype=np.uint8)
        u_sizes = np.array([len(p) for p in plains], dtype=np.int32)
        u_off = np.cumsum([0] + [(len(p) + 255) // 256 * 256 + 256 for p in plains])[:-1].astype(np.int64)
        d_dec = torch.zeros(int(u_off[-1]) + len(plains[-1]) + 512, dtype=torch.uint8, device=dev)
        d_dsz = torch.zeros(m, dtype=torch.int32, device=dev)
        d_dst = torch.full((m,), -1, dtype=torch.int32, device=dev)
        dc.uncompress(t(carena), t(c_off), t(np.array([len(s) for s in streams], dtype=np.int32)), d_dec, t(u_off), t(u_sizes),
                      d_dsz, d_dst, max(len(s) for s in streams), max(sizes))
        torch.cuda.synchronize()
        dst, dsz, dec = d_dst.cpu().numpy(), d_dsz.cpu().numpy(), d_dec.cpu().numpy()
        for i, p in enumerate(plains):
            ref = oracle.uncompress(streams[i], capacity=len(p), out_size_hint=len(p))
            assert ref == p
            assert dst[i] == 0 and dsz[i] == len(p), (order, i, int(dst[i]))
            assert dec[u_off[i]:u_off[i] + len(p)].tobytes() == p, (order, i)
        # a stream with more planes than reserved is refused, its neighbours are not
        assert L.rans4x16_hip_set_dev_stripe_planes(dc.ctx.h, 1, max(sizes)) == 0
        d_dst.fill_(-1)
        dc.uncompress(t(carena), t(c_off), t(np.array([len(s) for s in streams], dtype=np.int32)), d_dec, t(u_off), t(u_sizes),
                      d_dsz, d_dst, max(len(s) for s in streams), max(sizes))
        torch.cuda.synchronize()
        dst = d_dst.cpu().numpy()
        assert dst[-1] == 0 and dst[-2] == 0
        if N > 1:
            assert dst[0] == 6
        assert L.rans4x16_hip_set_dev_stripe_planes(dc.ctx.h, 0, 0) == 0
