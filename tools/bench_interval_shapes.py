"""Interval<F64> products of assorted shapes: the row-pair form (gft_conv_staged.hip k_pair_sums + k_pair_collect) forced on
("conv_rb_pairs" = 2) against forced off (0: k_conv_staged / k_conv_rows_rb) — where the size threshold belongs.
Usage: bench_interval_shapes.py [AxBxC ...]"""
import sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np
import genfer_amd

genfer_amd.init(0)
L = genfer_amd.lib()
TPI = genfer_amd.IntervalTaylorPoly
L.gft_set_option(b"host_max_elems", 0.0)
only = [tuple(int(t) for t in a.split('x')) for a in sys.argv[1:]]
shapes = only or [(40, 40), (48, 48), (64, 64), (100, 100), (128, 128), (300, 100), (12, 12, 12), (16, 16, 16), (20, 20, 20), (24, 24, 24), (8, 8, 32), (16, 16, 32), (24, 24, 32), (32, 32, 32), (20, 20, 64), (40, 40, 40), (12, 12, 12, 32), (16, 16, 16, 48), (64, 64, 128), (32, 32, 32, 32)]
for sh in shapes:
    rng = np.random.default_rng(0)
    lo = rng.random(sh); x = np.stack([lo, lo * (1 + 1e-15)])
    lo = rng.random(sh); y = np.stack([lo, lo * (1 + 1e-15)])
    macs = 1
    for n in sh:
        macs *= n * (n + 1) // 2
    row = []
    for mode in (0.0, 2.0):
        L.gft_set_option(b"conv_rb_pairs", mode)
        a, b = TPI.new(x, sh), TPI.new(y, sh)
        r = a * b
        L.gft_synchronize()
        best = 1e9
        for _ in range(3):
            L.gft_event_record(0)
            r = a * b
            L.gft_event_record(1)
            best = min(best, L.gft_event_elapsed_ms(0, 1))
        row.append(best)
    L.gft_set_option(b"conv_rb_pairs", -1.0)
    print(f"{'x'.join(map(str, sh)):>14s}  {macs:10.3e} MACs   off {row[0]:9.3f} ms   pairs {row[1]:9.3f} ms   x{row[0] / row[1]:.2f}", flush=True)
