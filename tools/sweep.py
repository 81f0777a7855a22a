#!/usr/bin/env python3
"""Occupancy / latency probe: times the encode and decode chain kernels for several batch sizes."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import htscodecs_amd as H
import bench

def run(nblk, bs, name, order, reps=2):
    dev = torch.device("cuda", 0)
    dc = H.DeviceCodec(0)
    for kv in filter(None, os.environ.get("OPTS", "").split(",")):      # OPTS=name=value,... : options of this context
        k, v = kv.split("="); dc.set_option(k, int(v))
    d_in, in_off, in_size = bench.build_batch(torch, dev, name, nblk, bs, 0)
    cap = H.rans_compress_bound_4x16(bs, order); slot = (cap + 255) // 256 * 256
    d_comp = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
    comp_off = torch.arange(nblk, dtype=torch.int64, device=dev) * slot
    comp_cap = torch.full((nblk,), cap, dtype=torch.int32, device=dev)
    comp_size = torch.zeros(nblk, dtype=torch.int32, device=dev); st = torch.zeros(nblk, dtype=torch.int32, device=dev)
    d_back = torch.zeros_like(d_in); bsz = torch.zeros(nblk, dtype=torch.int32, device=dev); st2 = torch.zeros(nblk, dtype=torch.int32, device=dev)
    def step():
        dc.compress(d_in, in_off, in_size, d_comp, comp_off, comp_cap, comp_size, st, order, bs)
        dc.uncompress(d_comp, comp_off, comp_size, d_back, in_off, in_size, bsz, st2, cap, bs)
    step(); torch.cuda.synchronize()
    dc.timing(True); dc.timing_read(0); dc.timing_read(1)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): step()
    e1.record(); torch.cuda.synchronize()
    enc, ne = dc.timing_read(0); dec, nd = dc.timing_read(1)
    ok = bool(torch.equal(d_back, d_in)) and int((st != 0).sum()) == 0 and int((st2 != 0).sum()) == 0
    tot = e0.elapsed_time(e1) / reps
    return dict(nblk=nblk, bs=bs, data=name, order=order, enc_chain_ms=round(enc / reps, 3), dec_chain_ms=round(dec / reps, 3),
                step_ms=round(tot, 3), enc_GBps=round(nblk * bs / (enc / reps) / 1e6, 2), dec_GBps=round(nblk * bs / (dec / reps) / 1e6, 2), ok=ok)

if __name__ == "__main__":
    bs = int(os.environ.get("BS", 1 << 18))
    name = os.environ.get("DATA", "q40+dir")
    order = int(os.environ.get("ORDER", 1))
    for n in [int(x) for x in sys.argv[1:]] or [256, 512, 1024, 1792, 2048, 3584, 4096]:
        print(json.dumps(run(n, bs, name, order)), flush=True)
