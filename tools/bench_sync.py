"""Cost of the host round trips: a fresh 8x8 tensor's extract_linear (scan kernel + 40-byte read-back) and a
coefficient() read-back, versus an asynchronous op of the same size (add), per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genfer_amd  # noqa: E402

genfer_amd.init(0)
L = genfer_amd.lib()
F = genfer_amd.TaylorPoly
rng = np.random.default_rng(0)
a = F.new(rng.random((8, 8)), [8, 8])
b = F.new(rng.random((8, 8)), [8, 8])
N = 2000


def per_call(fn):
    fn()
    L.gft_synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    L.gft_synchronize()
    return (time.perf_counter() - t0) / N * 1e6


print(f"async add            : {per_call(lambda: a + b):7.2f} us/op")
print(f"add + extract_linear : {per_call(lambda: (a + b).extract_linear()):7.2f} us/op")
print(f"add + coefficient    : {per_call(lambda: (a + b).coefficient([1, 1])):7.2f} us/op")
print(f"add + synchronize    : {per_call(lambda: ((a + b), L.gft_synchronize())):7.2f} us/op")
