#!/usr/bin/env python3
"""bench.py — headline benchmark: rANS 4x16 order-1 encode + decode of 1 MiB q40 blocks.

One *step* = one pass of the hot path over one batch of blocks that already lives in HBM:
rans4x16_hip_compress_dev over the whole batch, then rans4x16_hip_uncompress_dev over the result
(the shape of the reference's own `rans4x16pr -t` loop, tests/rANS_static4x16pr_test.c:180-222,
with the serial per-block loop replaced by batch calls).

  value      MB/s (1e6 B/s) of UNCOMPRESSED bytes through encode+decode: batch bytes / (t_enc+t_dec)
  roofline   for the slowest kernel of the step: algorithmic bytes (uncompressed + compressed,
             SURVEY.md §8d) / its HIP-event time, against the 8 TB/s HBM peak
  cpu_baseline  the oracle (oracle/rans4x16_oracle.c: the scalar C restatement of the reference's algorithm, pinned by
             the reference's fixtures) timed on this host's cores on a bounded sample - kind "port"

Launch: python bench.py [--gpus N --steps K --warmup W].  N > 1: one rank per GPU, either started by
torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK in the environment) or - when those are absent - by
this script itself, which then spawns N child ranks BEFORE it touches the GPU and relays rank 0's JSON line.
Blocks are independent: the global batch of N x blocks is cut by the library's own partition
(rans4x16_hip_partition) and the ranks share nothing but the final barrier / max.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def build_batch(torch, dev, name, nblk, blk_size, first_block):
    """nblk blocks of the cyclic repetition of base text `name`, built in HBM: block b
    is bytes [b*S, (b+1)*S) of the infinitely repeated text (what `rans4x16pr -t` sees on a tiled file).
    `mixed` cycles q4, q8, q40+dir by b mod 3 (BASELINE.json configs[4])."""
    import datagen
    names = ["q4", "q8", "q40+dir"] if name == "mixed" else [name]
    per = (nblk + len(names) - 1) // len(names)
    parts = []
    for nm in names:
        base = datagen.base_text(nm)
        total = per * blk_size
        start = (first_block * blk_size) % len(base)
        reps = (start + total + len(base) - 1) // len(base) + 1
        d_base = torch.from_numpy(np.ascontiguousarray(base)).to(dev)
        parts.append(d_base.repeat(reps)[start:start + total].view(per, blk_size))
    if len(parts) == 1:
        d_in = parts[0].reshape(-1)[:nblk * blk_size].contiguous()
    else:
        d_in = torch.stack(parts, dim=1).reshape(-1)[:nblk * blk_size].contiguous()   # block b <- text b % 3
    in_off = torch.arange(nblk, dtype=torch.int64, device=dev) * blk_size
    in_size = torch.full((nblk,), blk_size, dtype=torch.int32, device=dev)
    return d_in, in_off, in_size


def block_bytes(name, blk_size, b, first_block=0):
    """Host copy of block b of build_batch(name, ..., first_block)."""
    import datagen
    if name == "mixed":
        return np.ascontiguousarray(datagen.tile(["q4", "q8", "q40+dir"][b % 3], blk_size, first_block + b // 3))
    return np.ascontiguousarray(datagen.tile(name, blk_size, first_block + b))


def cpu_baseline(order, blk_size, name, seconds_target=12.0):
    """Time the CPU oracle (the scalar C port) on a bounded sample of the same workload (rank 0, N=1 only)."""
    import cpu_libs
    import datagen
    lib, kind = cpu_libs.oracle(), "port"
    # one GPU's share of the host is 16 cores on the bench pool; never oversubscribe beyond that
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("R4X16_CPU_THREADS", 16)))
    blocks = [block_bytes(name, blk_size, b) for b in range(cores)]
    cap = lib.bound(blk_size, order)

    def one(idx, reps):
        src = blocks[idx % len(blocks)]
        comp = np.zeros(cap + 16, dtype=np.uint8)
        back = np.zeros(blk_size + 16, dtype=np.uint8)
        t_enc = t_dec = 0.0
        for _ in range(reps):
            n = C.c_uint(cap)
            t0 = time.perf_counter()
            r = lib.compress_to(src.ctypes.data, blk_size, comp.ctypes.data, C.byref(n), order)
            t1 = time.perf_counter()
            m = C.c_uint(blk_size)
            r2 = lib.uncompress_to(comp.ctypes.data, n.value, back.ctypes.data, C.byref(m))
            t2 = time.perf_counter()
            assert r and r2 and m.value == blk_size
            t_enc += t1 - t0
            t_dec += t2 - t1
        assert (back[:blk_size] == src).all()
        return t_enc, t_dec

    te, td = one(0, 3)                                   # calibrate: seconds per block
    per_blk = (te + td) / 3
    reps = max(2, int(seconds_target / per_blk / 3))     # ~1/3 of the budget single-threaded
    te1, td1 = one(0, reps)
    one_thread = blk_size * reps / (te1 + td1) / 1e6
    reps_mt = max(2, int(seconds_target / per_blk / 2))  # ~1/2 of the budget (wall) on all threads
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:                # ctypes drops the GIL: real parallelism
        list(ex.map(lambda i: one(i, reps_mt), range(cores)))
    wall = time.perf_counter() - t0
    all_cores = blk_size * reps_mt * cores / wall / 1e6
    return {
        "value": round(all_cores, 1), "unit": "MB/s", "cores": cores, "kind": kind,
        "sample": f"{reps_mt} x {cores} blocks of {blk_size} B ({name}, order {order}), encode+decode, "
                  f"one thread per core",
        "value_1thread": round(one_thread, 1), "enc_1thread": round(blk_size * reps / te1 / 1e6, 1),
        "dec_1thread": round(blk_size * reps / td1 / 1e6, 1),
    }


def library_hash():
    import hashlib
    from htscodecs_amd import lib as _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def chain_figure(torch, dc, kname, kernel_ms, launches_per_step, nblk, bs, spc, lanes, cus, probe, sq):
    """SURVEY 8(d)'s second figure for the dominant chain kernel: symbols/s against (resident lanes / measured
    per-step latency).  The latency is measured live on ONE workgroup holding its full set of streams with the
    rest of the chip idle (`probe`); the whole-grid figure shows what contention for a CU's LDS and issue slots
    costs on top of it."""
    clk_khz = dc.L.rans4x16_hip_device_clock_khz(dc.ctx.h)
    resident = spc * cus
    per_launch = nblk / launches_per_step
    rounds = per_launch / resident
    steps = bs // 4                                        # steps of one chain over one block
    which = 1 if kname == "k_dec_chain" else 0
    k, ms1 = probe[which]
    iso_ns = ms1 * 1e6 / steps if ms1 > 0 else None
    grid_ns = kernel_ms * 1e6 / (max(1.0, float(-(-per_launch // resident))) * steps)
    sym_s = per_launch * bs / (kernel_ms / 1e3)
    ideal = resident * 4 / (iso_ns * 1e-9) if iso_ns else None
    out = {"kernel": kname, "streams_per_cu": spc, "lanes_live_per_wave": lanes, "compute_units": cus,
           "resident_streams": resident, "rounds": round(rounds, 3), "steps_per_chain": steps,
           "ns_per_step_whole_grid": round(grid_ns, 1), "cycles_per_step_whole_grid": round(grid_ns * clk_khz / 1e6, 1),
           "ns_per_step_one_workgroup_alone": round(iso_ns, 1) if iso_ns else None,
           "probe_streams": k, "symbols_per_s": round(sym_s, 0),
           "ideal_symbols_per_s": round(ideal, 0) if ideal else None,
           "contention_frac": round(sym_s / ideal, 4) if ideal else None,
           "meaning": "NOT an efficiency: the kernel against ITSELF run alone - ideal = resident streams x 4 chains / the "
                      "step latency of one workgroup alone on the chip; < 1 is contention inside a CU plus a partly filled "
                      "last round.  How good the step is: roofline.issue"}
    if sq:
        out["sq_counters"] = sq                            # VALU issue fraction etc. from the committed PMC pass (same build)
    return out


def issue_figure(kname, kernel_ms, launches_per_step, nblk, bs, spc, lanes, cus, clk_khz, sq, isa):
    """What actually bounds the chain kernels: instruction issue.  A wave that has its SIMD to itself issues one
    instruction every four cycles, dependent or not, so a step costs 4 x its instruction count plus whatever latency no
    instruction hides.  issue_frac = VALU instructions per wave and step x 4 cycles / measured cycles per step x
    (waves per CU / 4 SIMDs): the fraction of the CU's vector issue slots that carry the codec's arithmetic (1.0 = all
    four SIMDs issuing a vector instruction every slot).  VALU counts come from the SQ counters of the committed PMC pass
    (same build, same workload); the static count from the ISA is given beside them."""
    resident = spc * cus
    per_launch = nblk / launches_per_step
    rounds = max(1.0, float(-(-per_launch // resident)))
    steps = bs // 4
    waves_per_cu = -(-spc // (lanes // 4)) if lanes >= 4 else 0
    cyc = kernel_ms * 1e6 / (rounds * steps) * clk_khz / 1e6
    out = {"kernel": kname, "cycles_per_step": round(cyc, 1), "waves_per_cu": waves_per_cu, "simds_per_cu": 4,
           "streams_per_wave": lanes // 4}
    if isa:
        out["isa_instructions_per_step"] = isa.get("instructions_per_step")
        out["isa_valu_per_step"] = isa.get("valu_per_step")
    if sq and sq.get("SQ_WAVES") and sq.get("SQ_INSTS_VALU"):
        wave_steps = sq["SQ_WAVES"] * rounds * steps
        valu = sq["SQ_INSTS_VALU"] / wave_steps
        out["valu_per_wave_step"] = round(valu, 1)
        if sq.get("SQ_INSTS_LDS"):
            out["lds_per_wave_step"] = round(sq["SQ_INSTS_LDS"] / wave_steps, 1)
        if sq.get("SQ_INSTS_SALU"):
            out["salu_per_wave_step"] = round(sq["SQ_INSTS_SALU"] / wave_steps, 1)
        out["wave_issue_frac"] = round(valu * 4 / cyc, 4)
        out["issue_frac"] = round(valu * 4 / cyc * waves_per_cu / 4, 4)
        out["source"] = "SQ_INSTS_VALU / (SQ_WAVES x rounds x steps), profiles PMC pass of this build"
    return out


def host_path(H, name, bs, order, nblk=3072):
    """PCIe-inclusive rate through rans4x16_hip_{compress,uncompress}_batch, called as a C program would call them
    (ctypes straight onto the C ABI: pageable host arenas, pointer / size arrays, no per-block Python work):
    a bounded sample of the same workload - the shape of the reference's `-t` loop
    (tests/rANS_static4x16pr_test.c:191-206) with the serial loop replaced by the batch calls.  One warm pass
    (contexts, pinned bounce buffers, arenas), then the best of two timed passes each way.
    Reported beside `value`, never as `value`."""
    import datagen
    from htscodecs_amd import codec
    L = H.load()
    ctx = codec._thread_ctx()
    names = ["q4", "q8", "q40+dir"] if name == "mixed" else [name]
    src = np.empty(nblk * bs, dtype=np.uint8)
    for b in range(nblk):
        src[b * bs:(b + 1) * bs] = datagen.tile(names[b % len(names)], bs, b)
    cap = L.rans_compress_bound_4x16(bs, order)
    comp = np.ones(nblk * cap, dtype=np.uint8)             # (ones: every page is touched before the clock starts)
    back = np.ones(nblk * bs, dtype=np.uint8)
    vp = lambda a, stride: (C.c_void_p * nblk)(*[a.ctypes.data + i * stride for i in range(nblk)])
    in_p, comp_p, back_p = vp(src, bs), vp(comp, cap), vp(back, bs)
    in_sz = (C.c_uint * nblk)(*([bs] * nblk))
    ords = (C.c_int * nblk)(*([order] * nblk))
    status = (C.c_int * nblk)()
    t_enc, t_dec = [], []
    for rep in range(3):
        comp_sz = (C.c_uint * nblk)(*([cap] * nblk))
        t0 = time.perf_counter()
        rc = L.rans4x16_hip_compress_batch(ctx.h, nblk, in_p, in_sz, comp_p, comp_sz, ords, status)
        t1 = time.perf_counter()
        assert rc == 0, ctx.error()
        back_sz = (C.c_uint * nblk)(*([bs] * nblk))
        t2 = time.perf_counter()
        rc = L.rans4x16_hip_uncompress_batch(ctx.h, nblk, comp_p, comp_sz, back_p, back_sz, status)
        t3 = time.perf_counter()
        assert rc == 0, ctx.error()
        if rep:
            t_enc.append(t1 - t0); t_dec.append(t3 - t2)
    assert (back == src).all(), "host path round trip mismatch"
    tot = nblk * bs
    return {"enc_MBps": round(tot / min(t_enc) / 1e6, 1), "dec_MBps": round(tot / min(t_dec) / 1e6, 1),
            "value": round(tot / (min(t_enc) + min(t_dec)) / 1e6, 1), "unit": "MB/s",
            "sample": f"{nblk} x {bs} B {name} blocks, order {order}: pageable host buffers in, pageable host buffers out, "
                      f"through rans4x16_hip_compress_batch / rans4x16_hip_uncompress_batch; best of two passes after a warm one"}


def configs4_leg(torch, H, dc, dev, dist, red_dev, shard, rank, world, steps=3, warmup=1, nblk=32768, bs=65536, order=1):
    """BASELINE.json configs[4]'s real shape beside the headline: every GPU's share of a batch of 64 KiB blocks cycling
    q4 / q8 / q40 (b mod 3), order 1, encode + decode, device-resident; same barrier / max-over-ranks timing as the
    headline.  Every block must round-trip; rank 0 bit-compares a sample with the CPU reference."""
    lo, hi = shard.uniform_share(world * nblk, world, rank)
    d_in, in_off, in_size = build_batch(torch, dev, "mixed", nblk, bs, lo)
    cap = H.rans_compress_bound_4x16(bs, order)
    slot = (cap + 255) // 256 * 256
    d_comp = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    comp_off = torch.arange(nblk, dtype=torch.int64, device=dev) * slot
    comp_cap = torch.full((nblk,), cap, dtype=torch.int32, device=dev)
    comp_size = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st_e = torch.full((nblk,), -1, dtype=torch.int32, device=dev)
    d_back = torch.zeros_like(d_in)
    back_size = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st_d = torch.full((nblk,), -1, dtype=torch.int32, device=dev)

    def step():
        dc.compress(d_in, in_off, in_size, d_comp, comp_off, comp_cap, comp_size, st_e, order, bs)
        dc.uncompress(d_comp, comp_off, comp_size, d_back, in_off, in_size, back_size, st_d, cap, 0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = shard.max_over_ranks(dist, time.perf_counter() - t0, red_dev)
    # gate: one more, untimed, step into CLEARED outputs (a step that silently wrote nothing must not pass on what the
    # warm-up left behind)
    d_back.zero_(); d_comp.zero_(); comp_size.zero_(); back_size.zero_(); st_e.fill_(-1); st_d.fill_(-1)
    step()
    torch.cuda.synchronize()
    assert int((st_e != 0).sum()) == 0 and int((st_d != 0).sum()) == 0 and torch.equal(back_size, in_size) \
        and torch.equal(d_back, d_in), "configs[4] leg: round trip"
    if rank == 0:
        import cpu_libs
        chk = cpu_libs.oracle()
        csz = comp_size.cpu().numpy()
        rs = np.random.RandomState(11)
        for b in sorted(set([0, 1, 2, nblk - 1] + [int(x) for x in rs.randint(0, nblk, size=28)])):
            # block b of this rank = global block lo + b: text (lo + b) % 3 ... build_batch cycles by LOCAL index
            want = chk.compress(block_bytes("mixed", bs, b, lo).tobytes(), order)
            got = d_comp[b * slot:b * slot + int(csz[b])].cpu().numpy().tobytes()
            assert got == want, f"configs[4] leg: block {b} differs from the CPU reference"
    return {"workload": f"{nblk} x {bs} B blocks per GPU cycling q4/q8/q40+dir, order {order}, encode+decode, device-resident",
            "value": round(world * nblk * bs / (elapsed / steps) / 1e6, 1), "unit": "MB/s", "n_gpus": world,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "warmup": warmup}


HETERO_ORDERS = (0, 1, 65, 193)


def hetero_plan(total_bytes, seed=2024, lo=4096, hi=1 << 20, uniform=None, first_block=0):
    """Block sizes, texts and orders of the heterogeneous batch (host arithmetic only): sizes log-uniform in
    [lo, hi] - or all `uniform` bytes for the equal-bytes, equal-mix comparison batch -, text q4 / q8 / q40+dir by
    b mod 3, order drawn from HETERO_ORDERS.  The shape of the reference's real callers: token columns of any size
    (htscodecs/tokenise_name3.c:1246-1300), a short last block (tests/rANS_static4x16pr_test.c:139-176)."""
    rs = np.random.RandomState(seed)
    if uniform:
        n = max(1, int(total_bytes // uniform))
        sizes = np.full(n, uniform, dtype=np.int64)
    else:
        mean = (hi - lo) / np.log(hi / lo)
        draw = np.exp(rs.uniform(np.log(lo), np.log(hi), size=int(total_bytes / mean * 1.3) + 64)).astype(np.int64)
        n = int(np.searchsorted(np.cumsum(draw), total_bytes)) + 1
        sizes = draw[:n]
    orders = np.asarray(HETERO_ORDERS, dtype=np.int32)[rs.randint(0, len(HETERO_ORDERS), size=n)]
    text = (np.arange(n) + first_block) % 3
    return sizes, orders, text


HETERO_TEXTS = ("q4", "q8", "q40+dir")


def hetero_batch(torch, dev, sizes, text):
    """The batch in HBM: the blocks of one text are consecutive slices of its cyclic repetition, the three texts'
    regions follow each other; block b lies at in_off[b].  Returns (d_in, in_off, in_size, text_off) - text_off[b] is
    the block's position in its text's repetition (for the host copy)."""
    import datagen
    n = len(sizes)
    in_off = np.zeros(n, dtype=np.int64)
    text_off = np.zeros(n, dtype=np.int64)
    parts, base_at = [], 0
    for t, nm in enumerate(HETERO_TEXTS):
        idx = np.nonzero(text == t)[0]
        sz = sizes[idx]
        starts = np.concatenate([[0], np.cumsum(sz)[:-1]]) if len(sz) else np.zeros(0, dtype=np.int64)
        tot = int(sz.sum())
        base = datagen.base_text(nm)
        d_base = torch.from_numpy(np.ascontiguousarray(base)).to(dev)
        parts.append(d_base.repeat(tot // len(base) + 2)[:tot])
        in_off[idx] = base_at + starts
        text_off[idx] = starts
        base_at += tot
    d_in = torch.cat(parts)
    return d_in, torch.from_numpy(in_off).to(dev), torch.from_numpy(sizes.astype(np.int32)).to(dev), text_off


def hetero_block(sizes, text, text_off, b):
    import datagen
    return np.ascontiguousarray(datagen.tile(HETERO_TEXTS[int(text[b])], int(sizes[b]), 0, offset=int(text_off[b])))


def hetero_run(torch, H, dc, dev, total_bytes, uniform=None, reps=8, check=64, seed=2024, first_block=0, part=None):
    """One heterogeneous (or comparison) batch through rans4x16_hip_{compress,uncompress}_dev with per-block orders:
    best-of-`reps` HIP-event times of each direction (every pass's time is in the result too: the context's shares settle
    over its first batches - option sched_learn - and the encoder's side-by-side classes vary by +-10 % from pass to pass),
    every block round-tripped, `check` blocks byte-compared with
    the CPU checker.  part = (rank, world): the plan is the whole job's (world x total_bytes), cut into contiguous ranges
    of near-equal bytes by the library's own weighted partition (rans4x16_hip_partition); this rank runs its range."""
    share = None
    if part is not None and part[1] > 1:
        from htscodecs_amd import shard
        rank, world = part
        sizes, orders, text = hetero_plan(total_bytes * world, seed=seed, uniform=uniform)
        lo, hi = shard.contiguous_partition(sizes, world)[rank]
        share = {"blocks": [int(lo), int(hi)], "bytes": int(sizes[lo:hi].sum()), "job_bytes": int(sizes.sum()),
                 "share_over_mean": round(float(sizes[lo:hi].sum()) * world / float(sizes.sum()), 4)}
        sizes, orders, text = sizes[lo:hi].copy(), orders[lo:hi].copy(), text[lo:hi].copy()
    else:
        sizes, orders, text = hetero_plan(total_bytes, seed=seed, uniform=uniform, first_block=first_block)
    n = len(sizes)
    d_in, in_off, in_size, text_off = hetero_batch(torch, dev, sizes, text)
    L = H.load()
    caps = np.fromiter((L.rans_compress_bound_4x16(int(s), int(o)) for s, o in zip(sizes, orders)), dtype=np.int64, count=n)
    slots = (caps + 255) // 256 * 256
    comp_off_h = np.concatenate([[0], np.cumsum(slots)[:-1]])
    d_comp = torch.zeros(int(slots.sum()), dtype=torch.uint8, device=dev)
    comp_off = torch.from_numpy(comp_off_h).to(dev)
    comp_cap = torch.from_numpy(caps.astype(np.int32)).to(dev)
    d_order = torch.from_numpy(orders).to(dev)
    comp_size = torch.zeros(n, dtype=torch.int32, device=dev)
    st_e = torch.full((n,), -1, dtype=torch.int32, device=dev)
    d_back = torch.zeros_like(d_in)
    back_size = torch.zeros(n, dtype=torch.int32, device=dev)
    st_d = torch.full((n,), -1, dtype=torch.int32, device=dev)
    max_in, max_cap = int(sizes.max()), int(caps.max())

    def enc():
        dc.compress(d_in, in_off, in_size, d_comp, comp_off, comp_cap, comp_size, st_e, 0, max_in, d_order=d_order,
                    total_in_size=int(sizes.sum()))

    def dec():
        dc.uncompress(d_comp, comp_off, comp_size, d_back, in_off, in_size, back_size, st_d, max_cap, max_in,
                      total_out_cap=int(sizes.sum()))

    enc(); dec(); torch.cuda.synchronize()                  # warm: workspace, side streams
    te, td = [], []
    for _ in range(reps):
        d_back.zero_(); back_size.zero_(); comp_size.zero_(); st_e.fill_(-1); st_d.fill_(-1)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); enc(); e1.record(); dec(); e2.record()
        torch.cuda.synchronize()
        te.append(e0.elapsed_time(e1)); td.append(e1.elapsed_time(e2))
    ok = int((st_e != 0).sum()) == 0 and int((st_d != 0).sum()) == 0 and bool(torch.equal(back_size, in_size)) \
        and bool(torch.equal(d_back, d_in))
    import cpu_libs
    chk = cpu_libs.oracle()
    csz = comp_size.cpu().numpy()
    rs = np.random.RandomState(5)
    big, small = int(np.argmax(sizes)), int(np.argmin(sizes))
    sample = sorted(set([0, n - 1, big, small] + [int(x) for x in rs.randint(0, n, size=max(0, check - 4))]))
    same = 0
    for b in sample:
        want = chk.compress(hetero_block(sizes, text, text_off, b).tobytes(), int(orders[b]))
        got = d_comp[int(comp_off_h[b]):int(comp_off_h[b]) + int(csz[b])].cpu().numpy().tobytes()
        assert got == want, f"hetero leg: block {b} (size {sizes[b]}, order {orders[b]}) differs from the CPU checker"
        same += 1
    tot = int(sizes.sum())
    be, bd = min(te), min(td)
    if os.environ.get("HETERO_VERBOSE"):
        print("hetero passes: enc ms", [round(x, 1) for x in te], "dec ms", [round(x, 1) for x in td], file=sys.stderr)
    return {"blocks": n, "bytes": tot, "sizes": "uniform %d" % uniform if uniform else "log-uniform 4096..1048576",
            "texts": "q4/q8/q40+dir by b mod 3", "orders": "drawn from %s" % (HETERO_ORDERS,),
            "enc_ms": round(be, 3), "dec_ms": round(bd, 3),
            "enc_ms_passes": [round(x, 1) for x in te], "dec_ms_passes": [round(x, 1) for x in td],
            "enc_ms_median": round(float(np.median(te)), 3), "dec_ms_median": round(float(np.median(td)), 3),
            "enc_GBps": round(tot / be / 1e6, 2), "dec_GBps": round(tot / bd / 1e6, 2),
            "both_GBps": round(tot / (be + bd) / 1e6, 2), "ratio": round(float(csz.sum()) / tot, 4),
            "roundtrip_ok": ok, "bytes_equal_cpu_blocks": same, "workspace_GB": round(dc.workspace_bytes() / 2**30, 2),
            **({"partition": share} if share else {})}


def hetero_leg(torch, H, dc, dev, total_bytes=16 << 30, first_block=0, part=None):
    """a batch whose blocks differ in length, alphabet and order, beside an equal-bytes, equal-mix
    batch of uniform 64 KiB blocks."""
    het = hetero_run(torch, H, dc, dev, total_bytes, first_block=first_block, part=part)
    assert het["roundtrip_ok"], "hetero leg: round trip"
    torch.cuda.empty_cache()
    uni = hetero_run(torch, H, dc, dev, total_bytes, uniform=65536, first_block=first_block, part=part)
    assert uni["roundtrip_ok"], "hetero leg (uniform 64 KiB): round trip"
    torch.cuda.empty_cache()
    return {"hetero": het, "uniform_64KiB": uni,
            "enc_ratio": round(het["enc_GBps"] / uni["enc_GBps"], 3), "dec_ratio": round(het["dec_GBps"] / uni["dec_GBps"], 3),
            "enc_ratio_median": round(uni["enc_ms_median"] * het["bytes"] / (het["enc_ms_median"] * uni["bytes"]), 3),
            "dec_ratio_median": round(uni["dec_ms_median"] * het["bytes"] / (het["dec_ms_median"] * uni["bytes"]), 3)}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (fresh processes; this parent
    never initialises the GPU), wait for them, pass rank 0's output through."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()            # (may open the runtime in THIS process on some wheels: harmless, the
                                                #  parent only starts fresh children and never execs)
    env = dict(os.environ)
    if have < n:
        if os.environ.get("R4X16_OVERSUBSCRIBE") != "1":
            sys.exit(f"bench.py: --gpus {n} but {have} GPU(s) visible (R4X16_OVERSUBSCRIBE=1 rehearses "
                     f"{n} ranks on the cards present, over gloo)")
        env["R4X16_DIST_BACKEND"] = "gloo"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if have and have < n:
            e["R4X16_FORCE_DEVICE"] = str(r % have)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    # poll: when one rank dies the others would sit in the barrier for ever - end them and fail
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = max(rc, abs(code))
                for q in live:
                    q.terminate()
                for q in live:
                    try:
                        q.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        q.kill()
                live = []
                break
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    # default: two full rounds of the decode chain kernel's resident streams (rans4x16_hip_residency x CUs; 45 per CU
    # x 256 = 11,520 for the 46-symbol order-1 tables of q40): the chain kernels are persistent, a batch is walked
    # in rounds of the resident stream count, and a partly filled last round costs a whole one
    ap.add_argument("--blocks", type=int, default=int(os.environ.get("R4X16_BLOCKS", 0)))
    ap.add_argument("--block-size", type=int, default=1 << 20)
    ap.add_argument("--data", default="q40+dir")
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host", action="store_true", help="skip the PCIe-inclusive host-buffer figure")
    ap.add_argument("--no-configs4", action="store_true", help="skip the mixed 64 KiB leg (BASELINE configs[4]'s shape)")
    ap.add_argument("--no-hetero", action="store_true", help="skip the heterogeneous leg (blocks of any size, alphabet and order)")
    ap.add_argument("--hetero-gib", type=float, default=16.0, help="bytes per GPU of the heterogeneous leg")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                       # does not return

    import torch
    import htscodecs_amd as H

    from htscodecs_amd import shard
    rank, world, local = shard.env_rank()
    # rehearsal knobs (not used by the driver): several ranks on one card over gloo
    if "R4X16_FORCE_DEVICE" in os.environ:
        local = int(os.environ["R4X16_FORCE_DEVICE"])
    elif os.environ.get("R4X16_OVERSUBSCRIBE") == "1" and 0 < torch.cuda.device_count() < world:
        local %= torch.cuda.device_count()      # a launcher's ranks on fewer cards than ranks (rehearsal): every rank
        os.environ.setdefault("R4X16_DIST_BACKEND", "gloo")     # takes gloo, RCCL cannot put two ranks on one card
    backend = os.environ.get("R4X16_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = shard.init(backend, device_id=dev)       # nccl = RCCL; used for the barrier and the max only
    red_dev = dev if backend == "nccl" else None

    bs, order = args.block_size, args.order
    dc = H.DeviceCodec(local)
    import datagen
    names = ["q4", "q8", "q40+dir"] if args.data == "mixed" else [args.data]
    nsym = max(len(set(datagen.base_text(nm).tolist()) | ({0} if order & 1 else set())) for nm in names)
    dec_spc, dec_lanes, cus = dc.residency(True, nsym, order & 1, 10)      # (every BASELINE text chooses shift 10)
    enc_spc, enc_lanes, _ = dc.residency(False, nsym, order & 1, 10)
    nblk = args.blocks if args.blocks > 0 else 2 * dec_spc * cus
    # the job is world x nblk blocks; this rank's contiguous share comes from the library's partition
    lo, hi = shard.uniform_share(world * nblk, world, rank)
    assert hi - lo == nblk, (lo, hi, nblk)
    first = lo
    d_in, in_off, in_size = build_batch(torch, dev, args.data, nblk, bs, first)
    cap = H.rans_compress_bound_4x16(bs, order)
    slot = (cap + 255) // 256 * 256
    d_comp = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    comp_off = torch.arange(nblk, dtype=torch.int64, device=dev) * slot
    comp_cap = torch.full((nblk,), cap, dtype=torch.int32, device=dev)
    comp_size = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st_enc = torch.zeros(nblk, dtype=torch.int32, device=dev)
    d_back = torch.zeros_like(d_in)
    back_size = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st_dec = torch.zeros(nblk, dtype=torch.int32, device=dev)

    # stage buffers for the inverse transforms only when the streams can carry them (include/rans4x16_hip.h)
    xf_cap = bs if order & 0xc0 else 0

    def step(ev=None):
        dc.compress(d_in, in_off, in_size, d_comp, comp_off, comp_cap, comp_size, st_enc, order, bs)
        if ev is not None:
            ev.record()
        dc.uncompress(d_comp, comp_off, comp_size, d_back, in_off, in_size, back_size, st_dec, cap, xf_cap)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    dc.timing(True)
    dc.timing_read(0); dc.timing_read(1)
    mids = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    begs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        begs[k].record()
        step(mids[k])
        ends[k].record()
    barrier()
    elapsed = time.perf_counter() - t0
    dc.timing(False)
    enc_chain_ms, enc_launches = dc.timing_read(0)
    dec_chain_ms, dec_launches = dc.timing_read(1)
    t_enc = sum(b.elapsed_time(m) for b, m in zip(begs, mids)) / 1e3 / args.steps
    t_dec = sum(m.elapsed_time(e) for m, e in zip(mids, ends)) / 1e3 / args.steps

    # ---- correctness gate: one more, untimed, step into CLEARED outputs (a step that silently wrote nothing must not
    # pass on what the warm-up left behind): every block decodes to its input, every status is 0, and 64+ blocks -
    # the first and last of every round of resident streams among them - are bit-compared with the CPU reference
    d_back.zero_(); d_comp.zero_(); comp_size.zero_(); back_size.zero_()
    st_enc.fill_(-1); st_dec.fill_(-1)
    step()
    torch.cuda.synchronize()
    assert int((st_enc != 0).sum()) == 0 and int((st_dec != 0).sum()) == 0, "device reported failures"
    assert torch.equal(back_size, in_size), "decoded sizes differ"
    assert torch.equal(d_back, d_in), "round trip mismatch"
    csz = comp_size.cpu().numpy()
    gate_blocks = 0
    if rank == 0:
        import cpu_libs
        import datagen
        chk = cpu_libs.oracle()
        res = max(1, dec_spc * cus)
        edges = [b for r in range(0, nblk, res) for b in (r, min(nblk, r + res) - 1)]
        rs = np.random.RandomState(7)
        sample = sorted(set(edges + [int(x) for x in rs.randint(0, nblk, size=64)]))
        for b in sample:
            want = chk.compress(block_bytes(args.data, bs, b, first).tobytes(), order)
            got = d_comp[b * slot:b * slot + int(csz[b])].cpu().numpy().tobytes()
            assert got == want, f"block {b}: device stream differs from the CPU reference"
        gate_blocks = len(sample)

    # ---- one workgroup alone: the per-step latency of a full set of streams without neighbours on the chip.  The SAME
    # kernel as the full grid's, so the short-step routes that such a small batch would take by itself are switched
    # off for these two calls (options of the context)
    probe = None
    if rank == 0:
        probe = {}
        saved = {k: dc.get_option(k) for k in ("dec_direct", "enc_direct")}
        for k_ in saved:
            dc.set_option(k_, 0)
        try:
            for which, spc, lanes in ((1, dec_spc, dec_lanes), (0, enc_spc, enc_lanes)):
                k = lanes // 4 if which == 1 else spc          # decode: one wave per workgroup; encode: one workgroup per CU
                k = min(k, nblk)
                for timed in (False, True):                    # (a first call of this size sets up side streams etc.)
                    dc.timing(timed); dc.timing_read(which)
                    if which == 1:
                        dc.uncompress(d_comp, comp_off[:k], comp_size[:k], d_back, in_off[:k], in_size[:k], back_size[:k], st_dec[:k], cap, xf_cap)
                    else:
                        dc.compress(d_in, in_off[:k], in_size[:k], d_comp, comp_off[:k], comp_cap[:k], comp_size[:k], st_enc[:k], order, bs)
                    torch.cuda.synchronize()
                ms, _ = dc.timing_read(which)
                dc.timing(False)
                probe[which] = (k, ms)
        finally:
            for k_, v_ in saved.items():
                dc.set_option(k_, v_)

    ws_headline = dc.workspace_bytes()
    per_rank_ms = [round(x / args.steps * 1e3, 3) for x in shard.gather_over_ranks(dist, elapsed, red_dev)]
    elapsed = shard.max_over_ranks(dist, elapsed, red_dev)
    c4 = None
    if not args.no_configs4:
        del d_back, d_comp
        d_back = d_comp = None
        torch.cuda.empty_cache()
        c4 = configs4_leg(torch, H, dc, dev, dist, red_dev, shard, rank, world)
    het = None
    if not args.no_hetero:
        # (the headline's arenas are gone by now: the leg needs ~70 GB of its own beside the workspace)
        d_back = d_comp = None
        d_in = in_off = in_size = comp_off = comp_cap = None
        torch.cuda.empty_cache()
        mine = hetero_leg(torch, H, dc, dev, total_bytes=int(args.hetero_gib * (1 << 30)), part=(rank, world))
        if dist is not None:
            dist.barrier()
        # every rank's figure, in rank order: how evenly the weighted partition cut the job
        per = {k: shard.gather_over_ranks(dist, mine["hetero"][k], red_dev) for k in ("enc_ms", "dec_ms", "bytes")}
        het = dict(mine)
        if world > 1:
            tot = sum(per["bytes"])
            het["per_rank"] = {"enc_ms": [round(x, 3) for x in per["enc_ms"]], "dec_ms": [round(x, 3) for x in per["dec_ms"]],
                               "bytes": [int(x) for x in per["bytes"]],
                               "max_share_over_mean": round(max(per["bytes"]) * world / tot, 4)}
            het["job"] = {"bytes": int(tot), "enc_GBps": round(tot / max(per["enc_ms"]) / 1e6, 2),
                          "dec_GBps": round(tot / max(per["dec_ms"]) / 1e6, 2)}

    if rank == 0:
        total_unc = nblk * bs * world
        comp_bytes = int(csz.sum())
        ms_per_step = elapsed / args.steps * 1e3
        value = total_unc / (elapsed / args.steps) / 1e6
        # dominant kernel = the slower chain kernel; algorithmic bytes = uncompressed + compressed
        enc_avg = enc_chain_ms / max(enc_launches, 1)
        dec_avg = dec_chain_ms / max(dec_launches, 1)
        launches_per_step = max(enc_launches, 1) / args.steps
        if enc_avg >= dec_avg:
            kname, kavg = "k_enc_chain", enc_avg
        else:
            kname, kavg = "k_dec_chain", dec_avg
        alg_bytes = (nblk * bs + comp_bytes) / launches_per_step
        achieved = alg_bytes / (kavg / 1e3) / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate runs, gfx950 FETCH correction applied) - only if they were taken on this very
        # workload AND with this very build of the library (its hash is stored with them): a kernel change without
        # a counter refresh reports null, never a stale figure
        traffic, traffic_source, sq, sq_all = None, None, None, {}
        build = library_hash()
        isa = {}
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_counts.json")), reverse=True):
            try:
                with open(path) as f:
                    j = json.load(f)
                if j.get("library_sha256_16") == build:
                    isa = j["kernels"]
                    break
            except (OSError, KeyError, ValueError):
                pass
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                with open(path) as f:
                    pmc = json.load(f)
                w = pmc["workload"]
                if (w["blocks"], w["block_size"], w["data"], w["order"]) != (nblk, bs, args.data, order) \
                        or pmc.get("library_sha256_16") != build:
                    continue
                for key, v in pmc["kernels"].items():       # the instantiation that did the work (k_dec_chain<true, 1>, not the
                    if key.startswith(kname) and v["traffic_bytes"] > (traffic or 0):   # nested tables' k_dec_chain<true, 3>)
                        traffic = v["traffic_bytes"]
                        traffic_source = os.path.relpath(path, ROOT)
                sq_all = pmc.get("sq", {})
                sq = sq_all.get(kname)
                break
            except (OSError, KeyError, ValueError):
                pass
        chain = chain_figure(torch, dc, kname, dec_avg if kname == "k_dec_chain" else enc_avg, launches_per_step,
                             nblk, bs, dec_spc if kname == "k_dec_chain" else enc_spc,
                             dec_lanes if kname == "k_dec_chain" else enc_lanes, cus, probe, sq)
        clk_khz = dc.L.rans4x16_hip_device_clock_khz(dc.ctx.h)
        issue = {
            "k_dec_chain": issue_figure("k_dec_chain", dec_avg, launches_per_step, nblk, bs, dec_spc, dec_lanes, cus, clk_khz,
                                        sq_all.get("k_dec_chain"), isa.get("k_dec_chain<true,1>")),
            "k_enc_chain": issue_figure("k_enc_chain", enc_avg, launches_per_step, nblk, bs, enc_spc, enc_lanes, cus, clk_khz,
                                        sq_all.get("k_enc_chain"), isa.get("k_enc_chain<true,true>")),
        }
        out = {
            "metric": "MB/s uncompressed throughput (encode+decode), rANS4x16 order-1, q40 blocks",
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32", "data": "synthetic",
            "config": {"workload": f"rANS4x16 order-{order & 1} encode+decode, {nblk} x {bs} B {args.data} "
                                   f"blocks per GPU (cyclic tiles of tests/dat/{args.data}), device-resident",
                       "order": order, "blocks_per_gpu": nblk, "block_size": bs,
                       "ratio": round(comp_bytes / (nblk * bs), 4)},
            "enc_MBps": round(nblk * bs / t_enc / 1e6, 1), "dec_MBps": round(nblk * bs / t_dec / 1e6, 1),
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_source, "library_sha256_16": build,
                         "limiter": "instruction issue of lone waves x resident streams, not bandwidth (see issue, chain)",
                         "issue": issue, "chain": chain, "avg_kernel_ms": round(kavg, 3),
                         "enc_chain_ms": round(enc_avg, 3), "dec_chain_ms": round(dec_avg, 3),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "note": "avg_kernel_ms = HIP-event time of the chain kernel per step; the kernel is launched "
                                 "once per LDS size class and all but one class exit in microseconds, so compare with "
                                 "rocprof's TotalDurationNs / steps (profiles/r04_final_working_launches.csv)"},
            "workspace_GB": round(ws_headline / 2**30, 2),
            "gate": {"roundtrip_blocks": nblk, "bytes_equal_cpu_blocks": gate_blocks,
                     "how": "untimed extra step into cleared outputs after the timed ones"},
        }
        out["per_rank_ms_per_step"] = {"each": per_rank_ms, "min": min(per_rank_ms), "max": max(per_rank_ms)}
        if c4:
            out["configs4"] = c4
        if het:
            out["hetero"] = het
        if world == 1 and not args.no_host:
            # The host-buffer calls are measured as a C program would see them: through their own context, with the
            # benchmark's device-resident context (its 35 GB workspace, its arenas) released first - an idle context
            # holding that much device memory was measured to cost these calls a quarter of their rate
            # (tools/host_probe2.py: 16.8 -> 23.6 GB/s decode).
            d_back = d_comp = d_in = None
            dc = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            if os.environ.get("BENCH_DEBUG"):
                free, total = torch.cuda.mem_get_info()
                print("before host_path: device memory free %.1f of %.1f GB; referrers of the codec: %s" % (
                    free / 1e9, total / 1e9, [type(o).__name__ for o in gc.get_objects() if type(o).__name__ in ("DeviceCodec", "_Ctx")]), file=sys.stderr)
            out["host_path"] = host_path(H, args.data, bs, order)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(order, bs, args.data)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
