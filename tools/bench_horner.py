"""subst_var with a linear substitution c + m*eps_w whose constant is a non-zero (widened) interval — the Horner loop
`--bounds` programs spend their time in (one launch runs every step; one workgroup per line along w).
Usage: bench_horner.py [n ...]   (2-d n x n tensors, substitution for variable 0 along variable 1)"""
import sys

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np

import genfer_amd

genfer_amd.init(0)
L = genfer_amd.lib()
TPI, TP = genfer_amd.IntervalTaylorPoly, genfer_amd.TaylorPoly
sizes = [int(a) for a in sys.argv[1:]] or [60, 120, 180, 270]
for n in sizes:
    rng = np.random.default_rng(0)
    lo = rng.random((n, n)) + 0.01
    a = TPI.new(np.stack([lo, lo * (1 + 1e-15)]), (n, n))
    s = np.zeros((2, 1, 2))
    s[0, 0, 0], s[1, 0, 0] = 0.25, 0.25 * (1 + 1e-15)
    s[0, 0, 1], s[1, 0, 1] = 0.5, 0.5 * (1 + 1e-15)
    sub = TPI.new(s, (n, n))
    s0 = np.ascontiguousarray(s.reshape(2, 2, 1))  # the same substitution for variable 0 along variable 0 itself (`--bounds`: v -> c + m*v)
    sub0 = TPI.new(s0, (n, n))
    for cls, aa, ss, tag in ((TPI, a, sub, "interval"), (TP, TP.new(lo, (n, n)), TP.new(s[0], (n, n)), "f64"),
                             (TPI, a, sub0, "iv  w==v"), (TP, TP.new(lo, (n, n)), TP.new(s0[0], (n, n)), "f64 w==v")):
        r = aa.subst_var(0, ss)
        L.gft_synchronize()
        reps = 5
        L.gft_event_record(0)
        for _ in range(reps):
            r = aa.subst_var(0, ss)
        L.gft_event_record(1)
        ms = L.gft_event_elapsed_ms(0, 1) / reps
        print(f"{tag:9s} subst_var {n}x{n}: {ms * 1000:8.1f} us  ({ms * 1000 / n:6.2f} us per Horner step)  result shape {r.shape()}", flush=True)
