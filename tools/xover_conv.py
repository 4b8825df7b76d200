import sys, os, json
import numpy as np
sys.path.insert(0, "/root/repo")
import genfer_amd
genfer_amd.init(0)
L = genfer_amd.lib(); F = genfer_amd.TaylorPoly
def timed(fn, reps=20):
    fn(); L.gft_synchronize(); L.gft_event_record(0)
    for _ in range(reps): r = fn()
    L.gft_event_record(1); return L.gft_event_elapsed_ms(0, 1) / reps
rng = np.random.default_rng(0)
for sh in [(8,8,8),(12,12,12),(16,16,16),(20,20,20),(24,24,24),(30,30,30),(40,40,40),(8,8,8,8),(12,12,12,12),(6,6,6,6),(100,10,10),(10,10,100),(30,30,4)]:
    x, y = rng.random(sh), rng.random(sh)
    a, b = F.new(x, list(sh)), F.new(y, list(sh))
    out = {"shape": sh, "macs": genfer_amd.conv_macs(sh, sh, sh)}
    for mode, name in ((3,"staged"),(2,"tiled")):
        L.gft_set_conv_mode(mode)
        try:
            out[name] = round(timed(lambda: a*b)*1000, 1)
        except Exception as e:
            out[name] = "unsupported"
    print(out, flush=True)
