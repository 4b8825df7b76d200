import sys; sys.path.insert(0, '.')
import numpy as np, genfer_amd
genfer_amd.init(0)
L = genfer_amd.lib()
L.gft_set_option(b"host_max_elems", 0.0)
T = genfer_amd.TaylorPoly
rng = np.random.default_rng(0)
A = T.new(rng.random((1, 40, 40)), [3, 40, 40])
lin = T.var(0, 0.8, 3) * T.from_scalar(0.2) + T.from_scalar(0.64)
b = genfer_amd.op_stats()
r = A * lin
a = genfer_amd.op_stats()
print({k: a[k] - b[k] for k in a if a[k] != b[k]}, r.coeffs_shape())
A2 = T.new(rng.random((2, 40, 40)), [3, 40, 40])
b = genfer_amd.op_stats(); r = A2 * lin; a = genfer_amd.op_stats()
print({k: a[k] - b[k] for k in a if a[k] != b[k]}, r.coeffs_shape())
