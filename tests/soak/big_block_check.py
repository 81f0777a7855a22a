import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, htscodecs_amd as H, datagen, cpu_libs
orc = cpu_libs.oracle()
for name, sz, order in (("q40+dir", 40 * (1 << 20) + 3, 1), ("q4", 130 * (1 << 20) + 1, 193), ("q8", 70 * (1 << 20), 0), ("q40+dir", 33 * (1 << 20) + 2, 9)):
    d = datagen.tile(name, sz, 1).tobytes()
    t0 = time.time(); e = H.rans_compress_4x16(d, order); t1 = time.time()
    w = orc.compress(d, order)
    x = H.rans_uncompress_4x16(w, len(d)); t2 = time.time()
    print(name, sz, "order", order, "enc ok", e == w, "dec ok", x == d, "clen", len(w), "enc %.2f s" % (t1 - t0), flush=True)
