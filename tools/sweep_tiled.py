"""Throughput of the tiled f64 product across sizes (TMAC/s), forced tiled mode."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genfer_amd  # noqa: E402

genfer_amd.init(0)
L = genfer_amd.lib()
F = genfer_amd.TaylorPoly
L.gft_set_conv_mode(int(os.environ.get("SWEEP_CONV_MODE", "2")))  # 2: forced tiled (the default of this tool); 0: what gft_mul does by itself
rng = np.random.default_rng(0)
shapes = [(32,) * 3, (40,) * 3, (48,) * 3, (56,) * 3, (64,) * 3, (80,) * 3, (96,) * 3, (112,) * 3, (128,) * 3, (16,) * 4, (24,) * 4,
          (32,) * 4, (48,) * 4, (64, 64, 128), (128, 128, 32), (256, 256, 16), (200, 200, 64),
          (378, 378), (1000, 1000), (2000, 2000), (100, 100, 300), (64, 64, 512)]
if len(sys.argv) > 1:
    shapes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]]
for sh in shapes:
    a, b = F.new(rng.random(sh), list(sh)), F.new(rng.random(sh), list(sh))
    macs = genfer_amd.conv_macs(sh, sh, sh)
    reps = max(3, min(50, int(2e11 / macs)))
    c = a * b
    L.gft_synchronize()
    L.gft_event_record(0)
    for _ in range(reps):
        c = a * b
    L.gft_event_record(1)
    ms = L.gft_event_elapsed_ms(0, 1) / reps
    print(f"{'x'.join(map(str, sh)):>16s}  {macs:10.3e} MACs  {ms * 1000:10.1f} us  {macs / ms / 1e9:7.2f} TMAC/s", flush=True)
