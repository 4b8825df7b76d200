import sys, time
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np, genfer_amd
genfer_amd.init(0)
L = genfer_amd.lib()
TPI, TP = genfer_amd.IntervalTaylorPoly, genfer_amd.TaylorPoly
sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128]
for n in sizes:
    rng = np.random.default_rng(0)
    lo = rng.random((n, n, n)); x = np.stack([lo, lo * (1 + 1e-15)])
    lo = rng.random((n, n, n)); y = np.stack([lo, lo * (1 + 1e-15)])
    a, b = TPI.new(x, (n,)*3), TPI.new(y, (n,)*3)
    r = a * b; L.gft_synchronize()
    L.gft_event_record(0); r = a * b; L.gft_event_record(1)
    ms = L.gft_event_elapsed_ms(0, 1)
    macs = (n * (n + 1) // 2) ** 3
    print(f"interval mul {n}^3 (positive data): {ms:.2f} ms  {macs / ms / 1e9:.3f} TMAC/s (interval MACs)", flush=True)
    # mixed-sign data: the general path of the element functor (no positive-regime shortcut)
    xm, ym = np.stack([x[0] - 0.5, x[0] - 0.5 + 1e-15]), np.stack([y[0] - 0.5, y[0] - 0.5 + 1e-15])
    if n <= 64:
        am, bm = TPI.new(xm, (n,)*3), TPI.new(ym, (n,)*3)
        r = am * bm; L.gft_synchronize()
        L.gft_event_record(0); r = am * bm; L.gft_event_record(1)
        print(f"   mixed-sign data: {L.gft_event_elapsed_ms(0, 1):.2f} ms", flush=True)
