"""Why is bench.py's in-process host_path slower than the same calls alone?  Variants: plain; after a small device-resident
call (side streams exist); with 60 GB of torch tensors alive."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import htscodecs_amd as H
import bench
mode = sys.argv[1]
keep = []
if mode in ("dev", "both"):
    import batch_sweep
    dc = H.DeviceCodec(0)
    batch_sweep.run(dc, 15, 1 << 20, "q40+dir", 1, reps=1, check=2)
    keep.append(dc)
if mode in ("mem", "both"):
    keep.append(torch.zeros(60 << 30, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
if mode.startswith("big"):
    import batch_sweep
    dc = H.DeviceCodec(0)
    batch_sweep.run(dc, 23040, 1 << 20, "q40+dir", 1, reps=1, check=2)
    torch.cuda.empty_cache()
    if mode == "bigdel":
        del dc
    else:
        keep.append(dc)
    import gc; gc.collect()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    print("free GB", free >> 30)
r = bench.host_path(H, "q40+dir", 1 << 20, 1, 3072)
print(mode, r["enc_MBps"], r["dec_MBps"])
