"""One 2-d slab division (the unit of the blocked div / log recurrences, gft_div2d.hip) timed with HIP events.
GFT_DIV2D_DIAG=1|2|4 switches parts of the kernel off (timing only, wrong results)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, genfer_amd
genfer_amd.init(0)
L = genfer_amd.lib()
L.gft_set_option(b"host_max_elems", 0.0)
T = genfer_amd.TaylorPoly
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or [32, 64]:
    x = rng.random((n, n)) * 0.1; y = rng.random((n, n)) * 0.1; y[0, 0] = 1.0
    a, b = T.new(x, [n, n]), T.new(y, [n, n])
    r = a / b; L.gft_synchronize()
    L.gft_event_record(0)
    for _ in range(10): r = a / b
    L.gft_event_record(1)
    print(f"{n}x{n} slab division: {L.gft_event_elapsed_ms(0, 1) / 10 * 1e3:.1f} us  (GFT_DIV2D_DIAG={os.environ.get('GFT_DIV2D_DIAG', '0')})")
