"""One heterogeneous batch (bench.hetero_run) through *_dev, for a rocprofv3 kernel trace: the timeline of the class launches.
Usage: hetero_trace.py GiB [uniform_block_size]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench, htscodecs_amd as H
dc = H.DeviceCodec(0)
for k in ("sched_sort", "sched_claim", "sched_concurrent"):
    if os.environ.get(k.upper()):
        dc.set_option(k, int(os.environ[k.upper()]))
uni = int(sys.argv[2]) if len(sys.argv) > 2 else None
print(json.dumps(bench.hetero_run(torch, H, dc, torch.device("cuda", 0), int(float(sys.argv[1]) * (1 << 30)), uniform=uni, reps=int(os.environ.get("REPS", 1)), check=8)))
