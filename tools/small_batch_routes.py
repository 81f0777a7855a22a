import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, htscodecs_amd as H, datagen
t = np.tile(np.ascontiguousarray(datagen.base_text("q40+dir")), 2)
for nblk, sz in ((5000, 4096), (500, 4096), (64, 65536), (2000, 32768), (16, 1 << 20)):
    datas = [t[(i * 997) % 90000:(i * 997) % 90000 + sz].tobytes() if sz <= 100000 else datagen.tile("q40+dir", sz, i).tobytes() for i in range(nblk)]
    orders = [1] * nblk
    for mode in ("0", "1"):
        os.environ["R4X16_HOST_PIPE_MB"] = mode
        best = 1e9
        for rep in range(4):
            t0 = time.time(); enc, st = H.compress_batch(datas, orders); t1 = time.time()
            dec, st = H.uncompress_batch(enc, [sz] * nblk); t2 = time.time()
            best = min(best, t1 - t0 + t2 - t1)
            e1, d1 = t1 - t0, t2 - t1
        assert dec == datas
        print("blocks %5d x %7d  route %s  enc+dec best %.1f ms (last enc %.1f dec %.1f)" % (nblk, sz, "single" if mode == "0" else "pipeline", best * 1e3, e1 * 1e3, d1 * 1e3), flush=True)
