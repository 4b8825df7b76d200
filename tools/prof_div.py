import sys
sys.path.insert(0, "/root/repo")
import numpy as np, genfer_amd
genfer_amd.init(0)
G = genfer_amd.TaylorPoly
rng = np.random.default_rng(0)
sh = (64, 64, 64)
x = rng.random(sh) * 0.1; y = rng.random(sh) * 0.1; y.flat[0] = 1.0
a, b = G.new(x, list(sh)), G.new(y, list(sh))
r = a / b
genfer_amd.lib().gft_synchronize()
