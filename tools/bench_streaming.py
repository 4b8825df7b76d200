#!/usr/bin/env python3
"""HBM-roofline check of the streaming TaylorPoly ops (A5, A6, A12, A13, A14 are judged
against HBM bandwidth at >= 64 MB tensors).  Prints GB/s of algorithmic traffic per op."""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import genfer_amd  # noqa: E402

genfer_amd.init(0)
L = genfer_amd.lib()
TP = genfer_amd.TaylorPoly
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shape = (n, n, n)
N = n**3
rng = np.random.default_rng(0)
a = TP.new(rng.random(shape), shape)
b = TP.new(rng.random(shape), shape)
two = TP.from_scalar(2.0)
results = {}


def timeit(name, fn, bytes_moved, reps=5):
    fn()  # warm
    L.gft_synchronize()
    best = 1e9
    for _ in range(reps):
        L.gft_event_record(0)
        r = fn()
        L.gft_event_record(1)
        ms = L.gft_event_elapsed_ms(0, 1)
        best = min(best, ms)
        del r
    gbs = bytes_moved / (best * 1e-3) / 1e9
    results[name] = {"ms": best, "GB/s": gbs, "frac_of_8TBps": gbs / 8000.0}
    print(f"{name:34s} {best:8.3f} ms  {gbs:8.1f} GB/s  ({gbs / 80:5.1f}% of 8 TB/s)")


B = 8
timeit("add (x + y)", lambda: a + b, 3 * N * B)
timeit("sub (x - y)", lambda: a - b, 3 * N * B)
timeit("neg", lambda: -a, 2 * N * B)
timeit("scale (2 * x)", lambda: two * a, 2 * N * B)
timeit("div by const", lambda: a / two, 2 * N * B)
for v in range(3):
    timeit(f"derivative(v={v}, 1)", lambda v=v: a.derivative(v, 1), 2 * (N - N // n) * B)
for v in range(3):
    timeit(f"shift_down(v={v}, 1)", lambda v=v: a.shift_down(v, 1), (2 * N - N // n) * B)
for v in range(3):
    timeit(f"shift_down(v={v}, n-1) [axis sum]", lambda v=v: a.shift_down(v, n - 1), (N + N // n) * B)
timeit("truncate_to_degree_p1(n/2)", lambda: a.truncate_to_degree_p1(n // 2), 2 * (N // 8) * B)
timeit("mul_var path (x * 0.5 eps_1)", lambda: a * (TP.from_scalar(0.5) * TP.var_at_zero(1, n)), 2 * N * B)
timeit("clone (O(1))", lambda: a.clone(), 0)
timeit("extract_linear scan", lambda: a.extract_linear(), N * B)
print(json.dumps(results))
