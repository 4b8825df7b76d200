#!/bin/bash
# Samples the shader clock and the socket power while the C2 product (128^3 f64, tiled kernel) runs back to back:
# is the FP64 peak quoted at 2.4 GHz reachable on this box?
export HIP_FORCE_DEV_KERNARG=1
python - <<'PY' &
import time, numpy as np, sys
sys.path.insert(0, '.')
import genfer_amd
genfer_amd.init(0)
L = genfer_amd.lib()
G = genfer_amd.TaylorPoly
rng = np.random.default_rng(0)
a, b = G.new(rng.random((128,) * 3), [128] * 3), G.new(rng.random((128,) * 3), [128] * 3)
t0 = time.time()
n = 0
while time.time() - t0 < 25:
    for _ in range(20):
        r = a * b
    L.gft_synchronize()
    n += 20
print("products per second", n / (time.time() - t0), flush=True)
PY
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Package Power" | tr -s ' \t' ' '
  echo "--"
  sleep 1
done
wait $BP
