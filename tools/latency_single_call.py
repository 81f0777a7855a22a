import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, datagen
import htscodecs_amd as H
for size in (1000, 65536, 1<<20):
    d = np.ascontiguousarray(datagen.tile('q40+dir', size, 0)).tobytes()
    c = H.rans_compress_4x16(d, 1); u = H.rans_uncompress_4x16(c)
    assert u == d
    t0=time.perf_counter(); 
    for _ in range(20): c = H.rans_compress_4x16(d, 1)
    t1=time.perf_counter()
    for _ in range(20): u = H.rans_uncompress_4x16(c)
    t2=time.perf_counter()
    print(size, "compress ms", round((t1-t0)/20*1e3,3), "uncompress ms", round((t2-t1)/20*1e3,3))
