#!/usr/bin/env python3
"""HBM-roofline check of the streaming TaylorPoly ops (A5, A6, A12, A13, A14 are judged
against HBM bandwidth at >= 64 MB tensors).  Prints GB/s of algorithmic traffic per op.

Every row times KERNELS: operations the library would only record on the handle (deferred chains, gft_api.hip §3.4
of DESIGN.md) are timed with `defer = 0` — one launch per operation, the kernel a consumer-less materialisation runs — and
the chain kernel that evaluates recorded stages inside an Add has rows of its own.  Rows whose result is memoised on the
buffer (extract_linear) get a fresh tensor per repetition, built outside the timed region."""
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


def timeit(name, fn, bytes_moved, reps=5, setup=None, defer=True):
    L.gft_set_option(b"defer", 1.0 if defer else 0.0)
    try:
        arg = setup() if setup else None
        fn(arg) if setup else fn()  # warm
        L.gft_synchronize()
        best = 1e9
        launches = None
        for _ in range(reps):
            arg = setup() if setup else None
            L.gft_synchronize()
            before = genfer_amd.op_stats()["launches"]
            L.gft_event_record(0)
            r = fn(arg) if setup else fn()
            L.gft_event_record(1)
            ms = L.gft_event_elapsed_ms(0, 1)
            launches = genfer_amd.op_stats()["launches"] - before
            best = min(best, ms)
            del r
    finally:
        L.gft_set_option(b"defer", 1.0)
    gbs = bytes_moved / (best * 1e-3) / 1e9
    results[name] = {"ms": best, "GB/s": gbs, "frac_of_8TBps": gbs / 8000.0, "launches": launches}
    flag = "" if (launches or bytes_moved == 0) else "   <-- NO KERNEL WAS LAUNCHED: not a measurement"
    print(f"{name:44s} {best:8.3f} ms  {gbs:8.1f} GB/s  ({gbs / 80:5.1f}% of 8 TB/s)  {launches} launch(es){flag}")


B = 8
timeit("add (x + y)", lambda: a + b, 3 * N * B)
timeit("sub (x - y)", lambda: a - b, 3 * N * B)
timeit("neg [defer=0: gather]", lambda: -a, 2 * N * B, defer=False)
timeit("scale (2 * x) [defer=0: gather]", lambda: two * a, 2 * N * B, defer=False)
timeit("div by const [defer=0: gather]", lambda: a / two, 2 * N * B, defer=False)
# the chain kernel: recorded stages evaluated inside the consuming Add (k_chain<E,true>), one pass over both operands
timeit("chain add: (-x) + (2 * y)  [k_chain]", lambda: (-a) + (two * b), 3 * N * B)
timeit("chain add: (x / 2 + 1) - y [k_chain]", lambda: (a / two + TP.from_scalar(1.0)) - b, 3 * N * B)
for v in range(3):
    timeit(f"derivative(v={v}, 1)", lambda v=v: a.derivative(v, 1), 2 * (N - N // n) * B)
for v in range(3):
    timeit(f"shift_down(v={v}, 1)", lambda v=v: a.shift_down(v, 1), (2 * N - N // n) * B)
for v in range(3):
    timeit(f"shift_down(v={v}, n-1) [axis sum]", lambda v=v: a.shift_down(v, n - 1), (N + N // n) * B)
timeit("truncate_to_degree_p1(n/2) [defer=0: gather]", lambda: a.truncate_to_degree_p1(n // 2), 2 * (N // 8) * B, defer=False)
timeit("mul_var path (x * 0.5 eps_1) [defer=0]", lambda: a * (TP.from_scalar(0.5) * TP.var_at_zero(1, n)), 2 * N * B, defer=False)
timeit("clone (O(1))", lambda: a.clone(), 0)
# the verdict is memoised per buffer: a fresh (dense => early-exit) tensor per repetition; worst case = a tensor that IS linear
timeit("extract_linear scan (dense: early exit; latency)", lambda t: t.extract_linear(), 0, setup=lambda: a + b)
lin = np.zeros(shape)
lin[0, 0, 0], lin[0, 1, 0] = 1.0, 2.0
timeit("extract_linear scan (linear tensor: full)", lambda t: t.extract_linear(), N * B, setup=lambda: TP.new(lin, shape))
bad = {k: v for k, v in results.items() if v["frac_of_8TBps"] > 1.0}
print(json.dumps(results))
if bad:
    print("rows above 100 % of the HBM roof (not kernels):", sorted(bad), file=sys.stderr)
    sys.exit(1)
