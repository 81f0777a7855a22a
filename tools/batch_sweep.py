#!/usr/bin/env python3
"""Batch-size sweep of the device-resident path: what a caller gets at realistic batch sizes, not only at the
headline's two full rounds of resident streams.  One JSON line per shape:

    whole encode pass / decode pass (HIP events around the *_dev calls), the chain kernels alone, GB/s of
    uncompressed data, plus the single-call latency of the five drop-in symbols on one 1 MiB block.

Usage:  python tools/batch_sweep.py [out.jsonl]      (env SHAPES="64,256,..." overrides the 1 MiB q40 counts)
Every round trip is checked (decode == input, all statuses 0); a sample of blocks is byte-compared with the oracle.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import htscodecs_amd as H
import bench


def run(dc, nblk, bs, name, order, reps=3, check=4):
    dev = torch.device("cuda", 0)
    d_in, in_off, in_size = bench.build_batch(torch, dev, name, nblk, bs, 0)
    cap = H.rans_compress_bound_4x16(bs, order)
    slot = (cap + 255) // 256 * 256
    d_comp = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    comp_off = torch.arange(nblk, dtype=torch.int64, device=dev) * slot
    comp_cap = torch.full((nblk,), cap, dtype=torch.int32, device=dev)
    comp_size = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st = torch.zeros(nblk, dtype=torch.int32, device=dev)
    d_back = torch.zeros_like(d_in)
    bsz = torch.zeros(nblk, dtype=torch.int32, device=dev)
    st2 = torch.zeros(nblk, dtype=torch.int32, device=dev)
    xf = bs if order & 0xc0 else 0

    def enc():
        dc.compress(d_in, in_off, in_size, d_comp, comp_off, comp_cap, comp_size, st, order, bs)

    def dec():
        dc.uncompress(d_comp, comp_off, comp_size, d_back, in_off, in_size, bsz, st2, cap, xf)

    enc(); dec(); torch.cuda.synchronize()
    dc.timing(True); dc.timing_read(0); dc.timing_read(1)
    te, td = [], []
    for _ in range(reps):
        d_back.zero_(); bsz.zero_()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); enc(); e1.record(); dec(); e2.record()
        torch.cuda.synchronize()
        te.append(e0.elapsed_time(e1)); td.append(e1.elapsed_time(e2))
    encc, ne = dc.timing_read(0)
    decc, nd = dc.timing_read(1)
    dc.timing(False)
    ok = bool(torch.equal(d_back, d_in)) and int((st != 0).sum()) == 0 and int((st2 != 0).sum()) == 0
    # a few blocks against the CPU checker (first, last, and evenly spaced ones)
    import cpu_libs
    chk = cpu_libs.oracle()
    csz = comp_size.cpu().numpy()
    same = True
    for b in sorted(set([0, nblk - 1] + [int(x) for x in np.linspace(0, nblk - 1, check)])):
        want = chk.compress(bench.block_bytes(name, bs, b).tobytes(), order)
        got = d_comp[b * slot:b * slot + int(csz[b])].cpu().numpy().tobytes()
        same = same and got == want
    tot = nblk * bs
    best_e, best_d = min(te), min(td)
    return {"blocks": nblk, "block_size": bs, "data": name, "order": order,
            "enc_ms": round(best_e, 3), "dec_ms": round(best_d, 3),
            "enc_chain_ms": round(encc / reps, 3), "dec_chain_ms": round(decc / reps, 3),
            "enc_GBps": round(tot / best_e / 1e6, 2), "dec_GBps": round(tot / best_d / 1e6, 2),
            "both_GBps": round(tot / (best_e + best_d) / 1e6, 2),
            "roundtrip_ok": ok, "bytes_equal_cpu": same}


def single_call(size=1 << 20, name="q40+dir", order=1, reps=10):
    """The literal drop-in: rans_compress_to_4x16 / rans_uncompress_to_4x16 on one block from one thread."""
    import datagen
    d = np.ascontiguousarray(datagen.tile(name, size, 0)).tobytes()
    c = H.rans_compress_4x16(d, order)
    assert H.rans_uncompress_4x16(c) == d
    t0 = time.perf_counter()
    for _ in range(reps):
        c = H.rans_compress_4x16(d, order)
    t1 = time.perf_counter()
    for _ in range(reps):
        u = H.rans_uncompress_4x16(c)
    t2 = time.perf_counter()
    return {"single_call": True, "block_size": size, "data": name, "order": order,
            "compress_ms": round((t1 - t0) / reps * 1e3, 3), "uncompress_ms": round((t2 - t1) / reps * 1e3, 3)}


def hetero(out_path, gib):
    """Batches whose blocks differ in length, alphabet and order (bench.hetero_leg), with the scheduling of the chain
    kernels switched on step by step: round 3's behaviour (arrival order, fixed stride, classes one after the other),
    + length-sorted class lists, + claimed shares, + classes side by side - one JSON line each, the same library."""
    out = open(out_path, "w") if out_path else None
    dev = torch.device("cuda", 0)
    steps = (("round 3: arrival order, fixed stride, classes in stream order", 0, 0, 0),
             ("+ class lists sorted by length", 1, 0, 0),
             ("+ shares claimed from a counter", 1, 1, 0),
             ("+ classes side by side (device-written plan)", 1, 1, 1))
    only = os.environ.get("HETERO_STEPS")
    for what, so, cl, co in steps:
        if only and "%d%d%d" % (so, cl, co) not in only.split(","):
            continue
        dc = H.DeviceCodec(0)
        dc.set_option("sched_sort", so); dc.set_option("sched_claim", cl); dc.set_option("sched_concurrent", int(os.environ.get("CONC_MODE", 1)) if co else 0)
        if os.environ.get("MAX_WS_MB"):
            dc.set_option("max_workspace_mb", int(os.environ["MAX_WS_MB"]))
        r = bench.hetero_leg(torch, H, dc, dev, total_bytes=int(gib * (1 << 30)))
        r.update({"scheduling": what, "sched_sort": so, "sched_claim": cl, "sched_concurrent": co, "GiB": gib})
        line = json.dumps(r)
        print(line, flush=True)
        if out:
            out.write(line + "\n"); out.flush()
        del dc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--hetero":
        hetero(sys.argv[2] if len(sys.argv) > 2 else None, float(os.environ.get("HETERO_GIB", 16)))
        sys.exit(0)
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    dc = H.DeviceCodec(0)

    def emit(r):
        line = json.dumps(r)
        print(line, flush=True)
        if out:
            out.write(line + "\n"); out.flush()

    counts = [int(x) for x in os.environ.get("SHAPES", "1,64,256,1024,4096,11520,23040").split(",")]
    for n in counts:
        emit(run(dc, n, 1 << 20, "q40+dir", 1))
    if not os.environ.get("SHAPES"):
        emit(run(dc, 32768, 65536, "mixed", 1))                 # configs[4], one GPU's share
        emit(run(dc, 1024, 65536, "mixed", 1))
        emit(run(dc, 1024, 1 << 20, "q8", 1))
        emit(run(dc, 1024, 1 << 20, "q4", 193))
        emit(run(dc, 1024, 1 << 20, "q40+dir", 0))
        emit(run(dc, 4096, 1 << 20, "q4", 193))                 # configs[3]'s flags on a batch (X_PACK | X_RLE | order 1)
        emit(run(dc, 4096, 1 << 20, "q8", 65))                  # X_RLE without X_PACK: two long streams per block
        emit(run(dc, 64, 1 << 20, "q8", 65))
        emit(run(dc, 1, 1 << 20, "q8", 65))
    del dc
    torch.cuda.empty_cache()
    emit(single_call())
    emit(single_call(65536))
    emit(single_call(1 << 20, "q8", 65))
