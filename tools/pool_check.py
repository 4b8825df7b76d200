import sys, ctypes
sys.path.insert(0, "/root/repo")
import genfer_amd
genfer_amd.init(0)
L = genfer_amd.lib()
src = open("/root/repo/tests/golden/sgcl/neurips2023/approx/mixture/mixture.sgcl").read()
for i in range(3):
    text, t = genfer_amd.run_sgcl(src, "--limit 60")
    st = (ctypes.c_size_t * 3)()
    L.gft_pool_stats(st)
    print(i, "in_use", st[0], "cached", st[1], "peak", st[2], "time", t["time_infer"])
