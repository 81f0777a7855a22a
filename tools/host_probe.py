"""PCIe-inclusive batch calls on 3,072 x 1 MiB q40 blocks (bench.py's host_path), alone in the process, with the
pipeline's own trace (R4X16_HOST_TRACE=1) for the timeline."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import htscodecs_amd as H
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
print(json.dumps(bench.host_path(H, "q40+dir", 1 << 20, 1, n)))
