import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, genfer_amd
genfer_amd.init(0)
L = genfer_amd.lib()
TPI, TP = genfer_amd.IntervalTaylorPoly, genfer_amd.TaylorPoly
for n in (32, 64):
    rng = np.random.default_rng(0)
    lo = rng.random((n, n, n)); x = np.stack([lo, lo * (1 + 1e-15)])
    lo = rng.random((n, n, n)); y = np.stack([lo, lo * (1 + 1e-15)])
    a, b = TPI.new(x, (n,)*3), TPI.new(y, (n,)*3)
    r = a * b; L.gft_synchronize()
    L.gft_event_record(0); r = a * b; L.gft_event_record(1)
    ms = L.gft_event_elapsed_ms(0, 1)
    macs = (n * (n + 1) // 2) ** 3
    print(f"interval mul {n}^3: {ms:.2f} ms  {macs / ms / 1e9:.3f} TMAC/s (interval MACs)")
    L.gft_set_conv_mode(1)
    af, bf = TP.new(x[0], (n,)*3), TP.new(y[0], (n,)*3)
    r = af * bf; L.gft_synchronize()
    L.gft_event_record(0); r = af * bf; L.gft_event_record(1)
    print(f"   f64 reference-order kernel: {L.gft_event_elapsed_ms(0, 1):.2f} ms")
    L.gft_set_conv_mode(0)
