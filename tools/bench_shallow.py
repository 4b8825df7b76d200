"""Shallow (stencil) products and fused general Horner steps (K<E>::conv_shallow): microseconds per launch and the
algorithmic HBM rate (read the large operand once, write the result once) against 8 TB/s.
Usage: bench_shallow.py [XSHAPE:YSHAPE ...]   e.g. 104x103x102:2x2x1"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genfer_amd  # noqa: E402

genfer_amd.init(0)
L = genfer_amd.lib()
L.gft_set_option(b"host_max_elems", 0.0)
rng = np.random.default_rng(0)
cases = ["104x103x102:2x2x1", "104x103x102:1x2x2", "52x52x51:2x2x1", "378x378:2x2", "28x27x26x25:2x1x2x1", "1x1x100:100x100x1",
         "100x1x1:100x100x100", "290x290:64x1"]
if len(sys.argv) > 1:
    cases = sys.argv[1:]
for interval in (False, True):
    F = genfer_amd.IntervalTaylorPoly if interval else genfer_amd.TaylorPoly
    for c in cases:
        xs, ys = (tuple(int(t) for t in s.split("x")) for s in c.split(":"))
        deg = [max(a + b - 1, 1) for a, b in zip(xs, ys)]
        x, y = rng.random(xs) + 0.1, rng.random(ys) + 0.1
        if interval:
            x, y = np.stack([x, x * 1.0000001]), np.stack([y, y * 1.0000001])
        a, b = F.new(x, deg), F.new(y, deg)
        r = a * b
        L.gft_synchronize()
        before = genfer_amd.op_stats()
        reps = 50
        L.gft_event_record(0)
        for _ in range(reps):
            r = a * b
        L.gft_event_record(1)
        us = L.gft_event_elapsed_ms(0, 1) / reps * 1e3
        after = genfer_amd.op_stats()
        nz = int(np.prod(r.coeffs_shape()))
        w = 2 if interval else 1
        gbs = 8.0 * w * (int(np.prod(xs)) + nz) / (us * 1e-6) / 1e9
        kind = "shallow" if after["shallow_products"] > before["shallow_products"] else ("tiled" if after["tiled"] > before["tiled"] else "other")
        print(f"{'interval' if interval else 'f64':>8s} {c:>28s}  z={nz:9d}  {us:8.1f} us  {gbs:7.0f} GB/s ({100 * gbs / 8000:4.1f}% of 8 TB/s)  [{kind}]", flush=True)

# the fused general Horner step: subst_var of a dense tensor by a 2 x 2 (binomial-like) substitution, per step
for interval in (False, True):
    F = genfer_amd.IntervalTaylorPoly if interval else genfer_amd.TaylorPoly
    for shape, v, w in (((104, 103, 102), 0, 1), ((378, 378), 0, 1), ((28, 27, 26, 25), 0, 2)):
        nd = len(shape)
        s = np.zeros([2 if ax in (v, w) else 1 for ax in range(nd)])
        idx = lambda i, j: tuple((i if ax == v else (j if ax == w else 0)) for ax in range(nd))
        s[idx(1, 0)], s[idx(0, 1)], s[idx(1, 1)] = 0.7, 0.15, 0.3
        a = rng.random(shape) + 0.1
        if interval:
            a, s = np.stack([a, a * 1.0000001]), np.stack([s, s * 1.0000001])
        A, S = F.new(a, list(shape)), F.new(s, list(shape))
        r = A.subst_var(v, S)
        L.gft_synchronize()
        before = genfer_amd.op_stats()
        L.gft_event_record(0)
        r = A.subst_var(v, S)
        L.gft_event_record(1)
        ms = L.gft_event_elapsed_ms(0, 1)
        after = genfer_amd.op_stats()
        steps = shape[v]
        print(f"{'interval' if interval else 'f64':>8s} subst_var {'x'.join(map(str, shape))} by 2x2 stencil: {ms * 1e3:9.1f} us total, {ms * 1e3 / steps:7.1f} us per step, "
              f"{after['launches'] - before['launches']} launches, {after['fused_horner_steps'] - before['fused_horner_steps']} fused steps", flush=True)
