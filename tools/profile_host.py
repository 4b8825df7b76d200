#!/usr/bin/env python3
"""Flat PC profile of the CALLING thread (interpreter + library host code) over N runs of one program.
Usage: profile_host.py <program substring> [flags] [runs] [backend: gpu|oracle]  ->  gpurun_out/host_profile_<name>.samples
Symbolise with tools/symbolize_samples.py (works on another machine with the same image: addresses are file offsets)."""
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genfer_amd  # noqa: E402

only = sys.argv[1]
flags = sys.argv[2] if len(sys.argv) > 2 else "--limit 100"
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 5
backend = sys.argv[4] if len(sys.argv) > 4 else "gpu"
f = [p for p in sorted(glob.glob(os.path.join(ROOT, "tests/golden/sgcl/**/*.sgcl"), recursive=True)) if only in p][0]
src = open(f).read()
SO = os.path.join(ROOT, "tools", "sampler", "libsampler.so")
if not os.path.exists(SO):  # (built here, before anything initialises the GPU)
    import subprocess
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", SO, os.path.join(ROOT, "tools", "sampler", "sampler.c"), "-lrt"])
S = ctypes.CDLL(SO)
S.sampler_start.argtypes, S.sampler_stop.argtypes, S.sampler_stop.restype = [ctypes.c_int, ctypes.c_size_t], [ctypes.c_char_p], ctypes.c_long
bounds = "--bounds" in flags
if backend == "gpu":
    genfer_amd.init(0)
    lib, prefix = genfer_amd.LIB_PATH, "gfti_" if bounds else "gft_"
else:
    lib, prefix = os.path.join(ROOT, "oracle", "liborc.so"), "orci_" if bounds else "orc_"
rc, text, t = genfer_amd.run_sgcl_with_backend(src, flags, lib, prefix)  # warm
assert rc == 0, text
PERIOD = int(os.environ.get("SAMPLE_US", "100"))
assert S.sampler_start(PERIOD, 1 << 20) == 0
best = None
for _ in range(runs):
    rc, text, t = genfer_amd.run_sgcl_with_backend(src, flags, lib, prefix)
    best = t["time_infer"] if best is None else min(best, t["time_infer"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = os.path.join(ROOT, "gpurun_out", "host_profile_%s%s.samples" % (os.path.basename(f).replace(".sgcl", ""), "_bounds" if bounds else ""))
n = S.sampler_stop(out.encode())
print(f"{n} samples of {PERIOD} us over {runs} runs (best Total inference time {best:.6f} s) -> {out}")
