"""Reference-order product: one-thread-per-output kernel (conv mode 1) vs conv mode 3 — the row-pair form (k_pair_sums +
k_pair_collect, round 4) where it applies (rank 2-4, rows of 8 .. 256 / 128 for intervals, row sums within the workspace cap),
else the LDS-staged kernel; GFT_RB_PAIRS=0 times the LDS-staged kernel alone — on the GPU: time per product and a
bit-exactness check of the two results (`staged_ms` is the mode-3 time).
    python tools/bench_staged.py            (needs a gfx950 device)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genfer_amd  # noqa: E402

genfer_amd.init(0)
L = genfer_amd.lib()
F, I = genfer_amd.TaylorPoly, genfer_amd.IntervalTaylorPoly


def timed(fn, reps):
    fn()
    L.gft_synchronize()
    L.gft_event_record(0)
    for _ in range(reps):
        r = fn()
    L.gft_event_record(1)
    return L.gft_event_elapsed_ms(0, 1) / reps, r


def case(shape, interval=False, reps=3):
    rng = np.random.default_rng(1)
    x, y = rng.random(shape) - 0.3, rng.random(shape) - 0.3
    if interval:
        a, b = I.new(np.stack([x, x + 1e-9]), list(shape)), I.new(np.stack([y, y + 1e-9]), list(shape))
    else:
        a, b = F.new(x, list(shape)), F.new(y, list(shape))
    macs = genfer_amd.conv_macs(shape, shape, shape)
    out = {"shape": list(shape), "interval": interval, "macs": macs}
    res = {}
    for mode, name in ((1, "naive"), (3, "staged")):
        L.gft_set_conv_mode(mode)
        ms, r = timed(lambda: a * b, reps)
        out[name + "_ms"] = round(ms, 4)
        out[name + "_tmacs"] = round(macs / ms / 1e9, 4)
        res[name] = np.asarray(r.array())
    L.gft_set_conv_mode(0)
    out["bit_identical"] = bool(np.array_equal(res["naive"], res["staged"], equal_nan=True))
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    rows = []
    shapes = [(4096,), (378, 378), (1000, 1000), (30, 30, 30), (64, 64, 64), (100, 20, 20), (16, 16, 16, 16),
              (8, 8, 8, 8, 8), (200, 200, 8), (24, 24), (12, 12, 12), (6, 6, 6), (40, 5, 5), (100, 100, 100)]
    ishapes = [(378, 378), (32, 32, 32), (64, 64, 64), (16, 16, 16, 16), (48, 48, 48)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(t) for t in arg.split("x")) for arg in sys.argv[1:]]
        ishapes = shapes
    for sh in shapes:
        rows.append(case(sh))
    for sh in ishapes:
        rows.append(case(sh, interval=True, reps=2))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/staged_vs_naive.json", "w"), indent=1)
