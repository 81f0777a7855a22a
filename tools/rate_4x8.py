#!/usr/bin/env python3
"""Device-resident rate of the rANS 4x8 kernels (first, plain version): N blocks of 1 MiB q40, both orders."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
import htscodecs_amd as H
from htscodecs_amd import codec
import bench
L = H.load(); ctx = codec._thread_ctx(); dev = torch.device("cuda", 0)
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bs = 1 << 20
d_in, in_off, in_size = bench.build_batch(torch, dev, "q40+dir", nblk, bs, 0)
cap = L.rans4x8_hip_compress_bound(bs); slot = (cap + 255) // 256 * 256
d_comp = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
comp_off = torch.arange(nblk, dtype=torch.int64, device=dev) * slot
comp_cap = torch.full((nblk,), cap, dtype=torch.int32, device=dev)
csz = torch.zeros(nblk, dtype=torch.int32, device=dev); st = torch.zeros(nblk, dtype=torch.int32, device=dev)
d_back = torch.zeros_like(d_in); bsz = torch.zeros(nblk, dtype=torch.int32, device=dev); st2 = torch.zeros(nblk, dtype=torch.int32, device=dev)
s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for order in (0, 1):
    def enc(): assert L.rans4x8_hip_compress_dev(ctx.h, nblk, d_in.data_ptr(), in_off.data_ptr(), in_size.data_ptr(), d_comp.data_ptr(), comp_off.data_ptr(), comp_cap.data_ptr(), csz.data_ptr(), st.data_ptr(), order, None, bs, s) == 0
    def dec(): assert L.rans4x8_hip_uncompress_dev(ctx.h, nblk, d_comp.data_ptr(), comp_off.data_ptr(), csz.data_ptr(), d_back.data_ptr(), in_off.data_ptr(), in_size.data_ptr(), bsz.data_ptr(), st2.data_ptr(), s) == 0
    enc(); dec(); torch.cuda.synchronize()
    t0 = time.perf_counter(); enc(); torch.cuda.synchronize(); t1 = time.perf_counter(); dec(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ok = bool(torch.equal(d_back, d_in)) and int((st != 0).sum()) == 0 and int((st2 != 0).sum()) == 0
    print(f"rANS 4x8 order {order}: {nblk} x 1 MiB q40: encode {nblk*bs/(t1-t0)/1e9:.1f} GB/s, decode {nblk*bs/(t2-t1)/1e9:.1f} GB/s, ratio {float(csz.sum())/(nblk*bs):.4f}, ok {ok}")
