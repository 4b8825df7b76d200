"""div / exp / log (leading-axis slab recurrences, mt:1076-1231) on the GPU vs the CPU oracle."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genfer_amd  # noqa: E402
from genfer_amd.taylor import bind  # noqa: E402

genfer_amd.init(0)
L = genfer_amd.lib()
G = genfer_amd.TaylorPoly
O = bind(ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liborc.so")), "orc_")
rng = np.random.default_rng(0)
shapes = [(32, 32, 32), (64, 64, 64), (200, 200), (24, 24, 24, 24)]
if len(sys.argv) > 1:
    shapes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]]
for sh in shapes:
    x = rng.random(sh) * 0.1
    y = rng.random(sh) * 0.1
    y.flat[0] = 1.0
    x.flat[0] = 0.5
    for name, fg, fo in (("div", lambda a, b: a / b, None), ("exp", lambda a, b: a.exp(), None), ("log", lambda a, b: b.log(), None)):
        ga, gb = G.new(x, list(sh)), G.new(y, list(sh))
        r = fg(ga, gb)
        L.gft_synchronize()
        tg = 1e9
        for _ in range(3):  # best of 3
            t0 = time.perf_counter()
            r = fg(ga, gb)
            L.gft_synchronize()
            tg = min(tg, time.perf_counter() - t0)
        row = f"{'x'.join(map(str, sh)):>14s} {name}: gpu {tg * 1e3:9.2f} ms"
        if np.prod(sh) <= 64 ** 3:
            oa, ob = O.new(x, list(sh)), O.new(y, list(sh))
            t0 = time.perf_counter()
            ro = fg(oa, ob)
            tc = time.perf_counter() - t0
            err = np.max(np.abs(np.asarray(r.array()) - np.asarray(ro.array())) / (np.abs(np.asarray(ro.array())) + 1e-300))
            row += f"   cpu oracle {tc * 1e3:9.2f} ms   max rel err {err:.2e}"
        print(row, flush=True)
