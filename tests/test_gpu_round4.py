"""Round 4 on the GPU: batches whose blocks differ in length, alphabet and order (the shape of the reference's real
callers: htscodecs/tokenise_name3.c:1246-1300 compresses token columns of any size, tests/rANS_static4x16pr_test.c:139-176
ends every file with a short block), through the device-resident calls with the chain kernels' scheduling switched every
way (r4x16_sched.h: length-sorted class lists, claimed shares, classes side by side); the sized calls; the options."""
import numpy as np
import pytest

import datagen

pytestmark = pytest.mark.gpu

FLAGSETS = [0, 1, 64, 65, 128, 129, 192, 193, 16, 17, 32]


@pytest.fixture(scope="module")
def H():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import htscodecs_amd
    htscodecs_amd.load()
    return htscodecs_amd


def _gen(rs, n):
    kind = int(rs.randint(0, 7))
    seed = int(rs.randint(0, 1 << 30))
    if kind == 0: a = datagen.rand(n, seed, int(rs.randint(1, 257)), 0)
    elif kind == 1: a = datagen.runs(n, int(rs.randint(1, 40)), int(rs.randint(1, 200)), seed, int(rs.randint(0, 200)))
    elif kind == 2: a = datagen.weighted(n, [int(rs.randint(1, 5000))] + [1] * int(rs.randint(1, 255)), seed)
    elif kind == 3: a = datagen.tile(str(rs.choice(["q4", "q8", "q40+dir", "qvar"])), n, int(rs.randint(0, 50)))
    elif kind == 4: a = np.resize(datagen.markov(min(n, 3000), int(rs.randint(2, 220)), seed, 0, float(rs.random_sample())), n)
    elif kind == 5: a = datagen.const(n, int(rs.randint(0, 256)))
    else: a = np.resize(datagen.markov(min(n, 3000), int(rs.randint(40, 160)), seed, int(rs.randint(0, 90)), 0.3), n)
    return np.ascontiguousarray(a, dtype=np.uint8)


def _hetero(rs, nblk, big=6):
    """nblk blocks: sizes log-uniform in 1 .. 300,000 (+ a few of 1 MiB and some empty / tiny ones), alphabets of
    1 .. 256 symbols of every generator kind, flag sets drawn from FLAGSETS."""
    sizes = np.exp(rs.uniform(0.0, np.log(300000.0), size=nblk)).astype(np.int64)
    sizes[rs.randint(0, nblk, size=big)] = 1 << 20
    sizes[rs.randint(0, nblk, size=12)] = rs.randint(0, 9, size=12)
    blocks = [_gen(rs, int(n)) for n in sizes]
    orders = [int(rs.choice(FLAGSETS)) for _ in blocks]
    return blocks, orders


@pytest.fixture(scope="module")
def batch5000(oracle):
    rs = np.random.RandomState(20261005)
    blocks, orders = _hetero(rs, 5000)
    want = [oracle.compress(b.tobytes(), o) for b, o in zip(blocks, orders)]
    return blocks, orders, want


class _Arena:
    """Blocks in device memory the way a caller of the *_dev entry points holds them."""

    def __init__(self, H, dc, blocks, caps):
        import torch
        self.torch, self.dc, self.dev = torch, dc, dc.dev
        sizes = np.array([len(b) for b in blocks], dtype=np.int64)
        self.in_off_h = np.concatenate([[0], np.cumsum((sizes + 15) // 16 * 16)[:-1]]).astype(np.int64)
        arena = np.zeros(int(self.in_off_h[-1] + sizes[-1] + 64), dtype=np.uint8)
        for b, off in zip(blocks, self.in_off_h):
            arena[off:off + len(b)] = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b
        caps = np.asarray(caps, dtype=np.int64)
        self.out_off_h = np.concatenate([[0], np.cumsum((caps + 255) // 256 * 256)[:-1]]).astype(np.int64)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.d_in, self.in_off, self.in_size = t(arena), t(self.in_off_h), t(sizes.astype(np.int32))
        self.d_out = torch.zeros(int(self.out_off_h[-1] + caps[-1] + 256), dtype=torch.uint8, device=self.dev)
        self.out_off, self.out_cap = t(self.out_off_h), t(caps.astype(np.int32))
        self.out_size = torch.zeros(len(blocks), dtype=torch.int32, device=self.dev)
        self.status = torch.full((len(blocks),), -1, dtype=torch.int32, device=self.dev)
        self.sizes, self.caps = sizes, caps

    def results(self):
        self.torch.cuda.synchronize()
        st, osz, out = self.status.cpu().numpy(), self.out_size.cpu().numpy(), self.d_out.cpu().numpy()
        return st, [out[o:o + n].tobytes() for o, n in zip(self.out_off_h, osz)]


@pytest.mark.parametrize("sched", [(1, 1, 1), (0, 0, 0), (1, 1, 0), (0, 1, 1)], ids=["default", "round3", "sorted-claimed", "unsorted-side-by-side"])
def test_heterogeneous_batch_device_resident(H, batch5000, sched):
    """5,000 blocks of any size, alphabet and flag set through rans4x16_hip_{compress,uncompress}_dev_sized: every
    compressed block is the oracle's, every block decodes to its input - with the class lists sorted by length or not,
    the shares claimed or strided, the classes side by side or one after the other.  Two passes each way: the second
    one is dealt out over the streams from what the first one reported (r4x16_sched.h, SchedHint)."""
    import torch
    blocks, orders, want = batch5000
    dc = H.DeviceCodec(0)
    for name, v in zip(("sched_sort", "sched_claim", "sched_concurrent"), sched):
        dc.set_option(name, v)
    L = H.load()
    caps = [L.rans_compress_bound_4x16(len(b), o) for b, o in zip(blocks, orders)]
    A = _Arena(H, dc, blocks, caps)
    d_order = torch.from_numpy(np.array(orders, dtype=np.int32)).to(dc.dev)
    total = int(A.sizes.sum())
    for rep in range(2):
        A.d_out.zero_(); A.out_size.zero_(); A.status.fill_(-1)
        dc.compress(A.d_in, A.in_off, A.in_size, A.d_out, A.out_off, A.out_cap, A.out_size, A.status, 0, int(A.sizes.max()),
                    d_order=d_order, total_in_size=total)
        st, enc = A.results()
        bad = [(i, len(blocks[i]), orders[i], int(st[i])) for i in range(len(blocks)) if st[i] != 0 or enc[i] != want[i]]
        assert not bad, (rep, bad[:10])
    # decode the oracle's streams; capacities = the original sizes
    B = _Arena(H, dc, want, [max(len(b), 1) for b in blocks])
    B.out_cap = torch.from_numpy(A.sizes.astype(np.int32)).to(dc.dev)
    for rep in range(2):
        B.d_out.zero_(); B.out_size.zero_(); B.status.fill_(-1)
        dc.uncompress(B.d_in, B.in_off, B.in_size, B.d_out, B.out_off, B.out_cap, B.out_size, B.status, int(B.sizes.max()),
                      int(A.sizes.max()), total_out_cap=total)
        st, dec = B.results()
        bad = [(i, len(blocks[i]), orders[i], int(st[i])) for i in range(len(blocks))
               if (st[i] != 0 and len(blocks[i]) > 0) or (st[i] == 0 and dec[i] != blocks[i].tobytes())]
        assert not bad, (rep, bad[:10])


def test_sized_calls_with_totals_that_lie(H, oracle):
    """total_in_size / total_out_cap only size the staging of X_PACK / X_RLE blocks.  A caller that announces less than
    the batch holds gets UNSUPPORTED for the blocks whose staging does not fit - and the reference's bytes for every
    other block; announcing nothing (0) or too much changes nothing."""
    import torch
    rs = np.random.RandomState(77)
    blocks = [datagen.tile("q8", int(n), i) for i, n in enumerate(rs.randint(30000, 90000, size=64))]
    orders = [65 if i % 2 else 193 for i in range(64)]
    want = [oracle.compress(b.tobytes(), o) for b, o in zip(blocks, orders)]
    dc = H.DeviceCodec(0)
    L = H.load()
    caps = [L.rans_compress_bound_4x16(len(b), o) for b, o in zip(blocks, orders)]
    d_order = torch.from_numpy(np.array(orders, dtype=np.int32)).to(dc.dev)
    true_total = sum(len(b) for b in blocks)
    for total, all_ok in ((true_total, True), (0, True), (10 * true_total, True), (true_total // 8, False)):
        A = _Arena(H, dc, blocks, caps)
        dc.compress(A.d_in, A.in_off, A.in_size, A.d_out, A.out_off, A.out_cap, A.out_size, A.status, 0,
                    max(len(b) for b in blocks), d_order=d_order, total_in_size=total)
        st, enc = A.results()
        assert set(st.tolist()) <= {0, 6}, st.tolist()
        for i in range(64):
            if st[i] == 0:
                assert enc[i] == want[i], (total, i)
            else:
                assert enc[i] == b""
        if all_ok:
            assert not st.any(), (total, st.tolist())
        else:
            assert st[0] == 0 and (st == 6).sum() >= 32, (total, st.tolist())     # an eighth of the room: most do not fit
    # decode: the same for total_out_cap
    B = _Arena(H, dc, want, [len(b) for b in blocks])
    mx = max(len(b) for b in blocks)
    for total, all_ok in ((true_total, True), (0, True), (true_total // 8, False)):
        B.d_out.zero_(); B.out_size.zero_(); B.status.fill_(-1)
        dc.uncompress(B.d_in, B.in_off, B.in_size, B.d_out, B.out_off, B.out_cap, B.out_size, B.status, int(B.sizes.max()), mx,
                      total_out_cap=total)
        st, dec = B.results()
        assert set(st.tolist()) <= {0, 6}, st.tolist()
        for i in range(64):
            if st[i] == 0:
                assert dec[i] == blocks[i].tobytes(), (total, i)
        assert (not st.any()) if all_ok else (st[0] == 0 and (st == 6).sum() >= 32), (total, st.tolist())


def test_payload_in_place_at_every_slot_alignment(H, oracle):
    """Since round 4 the coder writes a block's payload backwards from the end of the caller's own slot and k_enc_finish
    moves it down behind the table (what the reference's coders do inside their output buffer,
    rANS_static4x16pr.c:396-402, :706-710).  Slots at every byte alignment, capacities of exactly the bound and beyond,
    incompressible data (the move then overlaps its source), every plain order."""
    import torch
    rs = np.random.RandomState(5)
    dc = H.DeviceCodec(0)
    L = H.load()
    blocks, orders = [], []
    for i in range(48):
        n = int(rs.choice([33, 1000, 4097, 70000, 200001]))
        kind = i % 3
        b = datagen.rand(n, i + 1, 256, 0) if kind == 0 else datagen.tile("q40+dir", n, i) if kind == 1 else datagen.weighted(n, [50] + [1] * 200, i + 1)
        blocks.append(np.ascontiguousarray(b)); orders.append(int(rs.choice([0, 1, 64, 65, 193])))
    want = [oracle.compress(b.tobytes(), o) for b, o in zip(blocks, orders)]
    caps = np.array([L.rans_compress_bound_4x16(len(b), o) + (i % 3) * 37 for i, (b, o) in enumerate(zip(blocks, orders))], dtype=np.int64)
    A = _Arena(H, dc, blocks, caps)
    # slots at odd alignments: shift every slot by its index modulo 16
    shift = np.arange(len(blocks)) % 16
    A.out_off_h = A.out_off_h + shift
    A.out_off = torch.from_numpy(A.out_off_h).to(dc.dev)
    A.d_out = torch.zeros(int(A.out_off_h[-1] + caps[-1] + 256), dtype=torch.uint8, device=dc.dev)
    d_order = torch.from_numpy(np.array(orders, dtype=np.int32)).to(dc.dev)
    dc.compress(A.d_in, A.in_off, A.in_size, A.d_out, A.out_off, A.out_cap, A.out_size, A.status, 0, int(A.sizes.max()), d_order=d_order)
    st, enc = A.results()
    bad = [(i, len(blocks[i]), orders[i], int(st[i])) for i in range(len(blocks)) if st[i] != 0 or enc[i] != want[i]]
    assert not bad, bad


def test_options_on_a_context(H):
    dc = H.DeviceCodec(0)
    for name, v in (("dec_direct", 3), ("sched_sort", 0), ("max_workspace_mb", 1234), ("host_lanes", 3)):
        before = dc.get_option(name)
        dc.set_option(name, v)
        assert dc.get_option(name) == v
        dc.set_option(name, before)
    with pytest.raises(KeyError):
        dc.set_option("no_such_option", 1)


def test_mid_rows_alphabet_shapes_and_overflow_buckets(H, oracle):
    """The mid rows (r4x16_common.h level 10: a bucket index per sixteen slots + one 16-byte window of cumulative values)
    serve order-1 streams with 10-bit tables of 13 .. 64 symbols in batches of one partly filled round.  With the direct
    rows switched off and dec_mid = 8 a batch of 2,300 blocks takes them over several rounds of the class: alphabets at
    the boundaries (12 / 13 / 48 / 49 / 64 / 65 symbols: the neighbours keep their own row kinds), many symbols of
    frequency 1 next to each other (buckets that overflow the window: the scan), contexts without a table row (damaged
    streams are covered by the fuzz tests), quality data, ragged sizes.  Device-resident, every block the oracle's."""
    import torch
    rs = np.random.RandomState(1010)
    blocks = []
    for ns in (12, 13, 14, 20, 31, 32, 33, 47, 48, 49, 63, 64, 65, 80):
        blocks.append(datagen.rand(30000 + ns, ns, ns, 33))                                   # flat: every symbol ~1024 / ns slots
        blocks.append(datagen.weighted(40001, [3000] + [1] * (ns - 1), ns + 1))               # one heavy symbol, the rest at frequency 1-2
        blocks.append(datagen.weighted(50003, [500, 400, 300] + [2] * (ns - 3), ns + 2))
        blocks.append((datagen.weighted(20000, [9] * ns, ns + 3).astype(np.int32) * 3 + 5).astype(np.uint8))     # gaps in the alphabet
    while len(blocks) < 2300:
        n = int(rs.choice([3000, 9000, 65536, 70001, 150000]))
        blocks.append(datagen.tile("q40+dir", n, len(blocks)))
    blocks = [np.ascontiguousarray(b) for b in blocks]
    orders = [1] * len(blocks)
    want = [oracle.compress(b.tobytes(), 1) for b in blocks]
    dc = H.DeviceCodec(0)
    dc.set_option("dec_direct", 0)
    dc.set_option("dec_mid", 8)
    B = _Arena(H, dc, want, [len(b) for b in blocks])
    for rep in range(2):
        B.d_out.zero_(); B.out_size.zero_(); B.status.fill_(-1)
        dc.uncompress(B.d_in, B.in_off, B.in_size, B.d_out, B.out_off, B.out_cap, B.out_size, B.status, int(B.sizes.max()), 0)
        st, dec = B.results()
        bad = [(i, len(blocks[i]), int(st[i])) for i in range(len(blocks)) if st[i] != 0 or dec[i] != blocks[i].tobytes()]
        assert not bad, (rep, bad[:10])
    # and as the budget rule picks them: 3,000 blocks (more than the direct rows hold, one round of the mid class)
    dc2 = H.DeviceCodec(0)
    dc2.set_option("dec_mid", 1)
    more = [np.ascontiguousarray(datagen.tile("q40+dir", 65536, i)) for i in range(3000)]
    wm = [oracle.compress(b.tobytes(), 1) for b in more[:40]]
    src = wm * 75
    B = _Arena(H, dc2, src, [65536] * 3000)
    dc2.uncompress(B.d_in, B.in_off, B.in_size, B.d_out, B.out_off, B.out_cap, B.out_size, B.status, int(B.sizes.max()), 0)
    st, dec = B.results()
    assert not st.any()
    assert all(dec[i] == more[i % 40].tobytes() for i in range(3000))
