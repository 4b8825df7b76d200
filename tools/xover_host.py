#!/usr/bin/env python3
"""Crossover of the size-threshold dispatch: the same dependent chain of TaylorPoly operations with every
tensor on the device (host_max_elems = 0) and with every tensor on the host tier (thresholds lifted), per size.  What is
timed is what a Genfer program does: K dependent operations, then ONE value read (the host round trip a device
chain ends in).  Prints microseconds per operation; the library defaults (gft_api.hip Runtime::HOST_MAX_*) are read off
this table.  Usage: python tools/xover_host.py > profiles/r02/xover_host.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genfer_amd  # noqa: E402

genfer_amd.init(0)
L = genfer_amd.lib()
T = genfer_amd.TaylorPoly
K = 200


def tier(host):
    L.gft_set_option(b"host_max_elems", 1e12 if host else 0.0)
    L.gft_set_option(b"host_max_macs", 1e18 if host else 0.0)


def chain(shape, op):
    deg = list(shape)
    rng = np.random.default_rng(1)
    x, y = T.new(rng.random(shape) * 0.5, deg), T.new(rng.random(shape) * 0.5, deg)
    best = None
    for _ in range(3):
        z = x
        t0 = time.perf_counter()
        for _ in range(K):
            z = op(z, y)
        z.coefficient([0] * len(shape))
        dt = (time.perf_counter() - t0) / K * 1e6
        best = dt if best is None else min(best, dt)
    return best


print(f"{'op':10s} {'shape':>14s} {'elems':>8s} {'MACs':>10s} {'device us/op':>13s} {'host us/op':>11s}")
for name, op, shapes in (
    ("add", lambda a, b: a + b, [(n,) for n in (16, 64, 256, 1024, 4096, 16384, 65536)]),
    ("mul 1-d", lambda a, b: a * b, [(n,) for n in (16, 32, 64, 128, 256, 512, 1024, 2048)]),
    ("mul 2-d", lambda a, b: a * b, [(n, n) for n in (4, 8, 12, 16, 24, 32, 48)]),
    ("mul 3-d", lambda a, b: a * b, [(n, n, n) for n in (3, 4, 6, 8, 10, 12)]),
):
    for shape in shapes:
        row = []
        for host in (False, True):
            tier(host)
            row.append(chain(shape, op))
        macs = genfer_amd.conv_macs(shape, shape, shape) if name.startswith("mul") else 0
        print(f"{name:10s} {str(shape):>14s} {int(np.prod(shape)):8d} {macs:10.0f} {row[0]:13.2f} {row[1]:11.2f}", flush=True)
tier(True)
L.gft_set_option(b"host_max_elems", -1.0)
L.gft_set_option(b"host_max_macs", -1.0)
