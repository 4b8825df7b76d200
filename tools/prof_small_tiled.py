"""Per-kernel cost of ONE mid-size product on the tiled path (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genfer_amd  # noqa: E402

genfer_amd.init(0)
L = genfer_amd.lib()
F = genfer_amd.TaylorPoly
sh = tuple(int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "24x24x24").split("x"))
rng = np.random.default_rng(0)
a, b = F.new(rng.random(sh), list(sh)), F.new(rng.random(sh), list(sh))
L.gft_set_conv_mode(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
for _ in range(50):
    c = a * b
L.gft_synchronize()
