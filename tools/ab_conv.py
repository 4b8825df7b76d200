#!/usr/bin/env python3
"""Interleaved A/B timing of tiled-kernel variants in ONE process on ONE device (the only valid
way to rank builds: cdna_hip_programming.md §5.4 rule 24).  Usage: ab_conv.py v0 v1 ... [--rounds=N] [--shape=AxBxC]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch  # noqa: E402

import genfer_amd  # noqa: E402
from bench import splitmix64_uniform  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = 7
for a in sys.argv[1:]:
    if a.startswith("--rounds="):
        rounds = int(a.split("=")[1])
variants = [int(v) for v in args] or [0, 1]
shape = [128, 128, 128]
for a in sys.argv[1:]:
    if a.startswith("--shape="):
        shape = [int(t) for t in a.split("=")[1].split("x")]
n = int(np.prod(shape))
genfer_amd.init(0)
L = genfer_amd.lib()
L.gft_set_conv_variant.argtypes = [ctypes.c_int]
L.gft_set_conv_mode(2)
x = torch.from_numpy(splitmix64_uniform(1, n).reshape(shape)).cuda()
y = torch.from_numpy(splitmix64_uniform(2, n).reshape(shape)).cuda()
z = torch.zeros(shape, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
times = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        L.gft_set_conv_variant(v)
        genfer_amd.conv_raw(x.data_ptr(), shape, y.data_ptr(), shape, z.data_ptr(), shape)  # warm / plan
        L.gft_event_record(0)
        for _ in range(3):
            genfer_amd.conv_raw(x.data_ptr(), shape, y.data_ptr(), shape, z.data_ptr(), shape)
        L.gft_event_record(1)
        ms = L.gft_event_elapsed_ms(0, 1) / 3
        if r > 0:
            times[v].append(ms)
macs = genfer_amd.conv_macs(shape, shape, shape)
for v in variants:
    t = np.array(times[v])
    med = float(np.median(t))
    print(f"variant {v:4d}: median {med:7.3f} ms  min {t.min():7.3f}  max {t.max():7.3f}   "
          f"{macs / med / 1e9:7.2f} TMAC/s = {2 * macs / (med * 1e-3) / 78.6e12 * 100:5.1f}% of 78.6 TFLOP/s")
