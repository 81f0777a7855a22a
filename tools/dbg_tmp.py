import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, datagen
import htscodecs_amd as H
d = np.ascontiguousarray(datagen.tile('q4', 100000, 0)).tobytes()
for order in (0, 1):
    out, st = H.compress_batch([d, d[:1000], d[:37]], [order]*3)
    print(order, [None if o is None else len(o) for o in out], st)
