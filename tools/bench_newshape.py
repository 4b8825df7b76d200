"""Cost of a product whose shape has not been seen before (plan build + upload) vs a repeated shape, on the tiled
path, and the same products on the staged reference-order kernel: Genfer's supports grow, so new shapes are common."""
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
for base in [(60, 60), (100, 100), (150, 150), (200, 200), (20, 20, 20), (40, 40, 40)]:
    for mode, name in ((3, "staged"), (0, "auto  ")):
        L.gft_set_conv_mode(mode)
        ops = []
        for i in range(24):  # 24 distinct shapes
            sh = tuple(b + i for b in base)
            ops.append((F.new(rng.random(sh), list(sh)), F.new(rng.random(sh), list(sh))))
        L.gft_synchronize()
        t0 = time.perf_counter()
        for a, b in ops:
            c = a * b
        L.gft_synchronize()
        t_new = (time.perf_counter() - t0) / len(ops)
        t0 = time.perf_counter()
        for a, b in ops:
            c = a * b
        L.gft_synchronize()
        t_rep = (time.perf_counter() - t0) / len(ops)
        print(f"{str(base):>14s} {name}: new shape {t_new * 1e6:8.1f} us/product   repeated {t_rep * 1e6:8.1f} us/product", flush=True)
