#!/usr/bin/env python3
"""Host-buffer decode of N x 1 MiB q40 blocks through the plain batch call and through the multi-device call with ONE
device listed (which binds its worker and copier threads to the device's NUMA node): is a slow box a wrong-socket box?
usage: host_probe_numa.py [blocks]"""
import ctypes as C, os, sys, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import htscodecs_amd as H
from htscodecs_amd import codec
import datagen
L = H.load()
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
bs, order = 1 << 20, 1
print("cpus allowed:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4], "...", "nodes:", [(os.path.basename(p), open(p + "/cpulist").read().strip()) for p in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))])
src = np.empty(nblk * bs, dtype=np.uint8)
for b in range(nblk):
    src[b * bs:(b + 1) * bs] = datagen.tile("q40+dir", bs, b)
cap = L.rans_compress_bound_4x16(bs, order)
comp = np.ones(nblk * cap, dtype=np.uint8); back = np.ones(nblk * bs, dtype=np.uint8)
vp = lambda a, stride: (C.c_void_p * nblk)(*[a.ctypes.data + i * stride for i in range(nblk)])
in_p, comp_p, back_p = vp(src, bs), vp(comp, cap), vp(back, bs)
in_sz = (C.c_uint * nblk)(*([bs] * nblk)); ords = (C.c_int * nblk)(*([order] * nblk)); status = (C.c_int * nblk)()
ctx = codec._thread_ctx()
m = L.rans4x16_hip_multi_create(1, (C.c_int * 1)(0))
print("device NUMA node:", L.rans4x16_hip_multi_numa_node(m, 0))
comp_sz = (C.c_uint * nblk)(*([cap] * nblk))
assert L.rans4x16_hip_compress_batch(ctx.h, nblk, in_p, in_sz, comp_p, comp_sz, ords, status) == 0
for name, call in (("plain", lambda bsz: L.rans4x16_hip_uncompress_batch(ctx.h, nblk, comp_p, comp_sz, back_p, bsz, status)),
                   ("multi(1 device, node-bound)", lambda bsz: L.rans4x16_hip_uncompress_batch_multi(m, nblk, comp_p, comp_sz, back_p, bsz, status)),
                   ("plain again", lambda bsz: L.rans4x16_hip_uncompress_batch(ctx.h, nblk, comp_p, comp_sz, back_p, bsz, status))):
    ts = []
    for rep in range(4):
        bsz = (C.c_uint * nblk)(*([bs] * nblk))
        t0 = time.perf_counter(); rc = call(bsz); t1 = time.perf_counter()
        assert rc == 0
        ts.append(t1 - t0)
    print(f"{name}: decode GB/s per pass", [round(nblk * bs / t / 1e9, 1) for t in ts])
assert (back == src).all()
# alternating directions, as bench.py's host_path runs them: does a decode right after an encode pay for something?
ts_e, ts_d = [], []
for rep in range(4):
    comp_sz = (C.c_uint * nblk)(*([cap] * nblk))
    t0 = time.perf_counter(); rc = L.rans4x16_hip_compress_batch(ctx.h, nblk, in_p, in_sz, comp_p, comp_sz, ords, status); t1 = time.perf_counter()
    assert rc == 0
    bsz = (C.c_uint * nblk)(*([bs] * nblk))
    t2 = time.perf_counter(); rc = L.rans4x16_hip_uncompress_batch(ctx.h, nblk, comp_p, comp_sz, back_p, bsz, status); t3 = time.perf_counter()
    assert rc == 0
    ts_e.append(t1 - t0); ts_d.append(t3 - t2)
print("alternating: encode GB/s", [round(nblk * bs / t / 1e9, 1) for t in ts_e], "decode GB/s", [round(nblk * bs / t / 1e9, 1) for t in ts_d])
print("workspace GB", round(L.rans4x16_hip_workspace_bytes(ctx.h) / 2**30, 2))
