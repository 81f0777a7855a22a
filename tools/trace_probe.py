"""One small mixed batch through *_dev, for a rocprofv3 kernel trace (timeline of the class launches)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import batch_sweep as B, htscodecs_amd as H
dc = H.DeviceCodec(0)
n, bs, name, order = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
print(B.run(dc, n, bs, name, order, reps=1, check=2))
