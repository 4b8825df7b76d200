#!/usr/bin/env python3
"""Predicts the 1 -> 8 GPU curve of the output-sharded product (BASELINE configs[3], 64^4) on ONE GPU: for world in {2, 4, 8}
and every rank r, runs exactly the local launches gft_conv_raw_sharded issues for gft_plan_slabs(n0, world, r) — the low
slab group and the mirrored high group, the same O::conv slab ranges as dist_conv (gft_api.hip) — and times them with the
library's own events.  Reports per-rank kernel ms and TMAC/s, max / mean over ranks, and the projected speed-up
    T_1 / (max_r T_r + result_bytes * (world - 1) / world / link_bw)
with link_bw = 300 GB/s (two of the seven xGMI links of a rank at ~153 GB/s: the all-gather's per-rank ingress on a ring).
Usage: rank_emulation.py [n] [rank] [reps]   (defaults 64 4 3 -> 64^4)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import genfer_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
LINK = 300e9
genfer_amd.init(0)
L = genfer_amd.lib()
shape = [n] * nd
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
x = torch.rand(shape, dtype=torch.float64, generator=g).to(dev)
y = torch.rand(shape, dtype=torch.float64, generator=g).to(dev)
z = torch.zeros(shape, dtype=torch.float64, device=dev)


def timed(ranges):
    """best-of-reps ms for the launches of the given slab ranges, back to back on the library's stream"""
    best = None
    for _ in range(reps + 1):
        L.gft_event_record(0)
        for lo, hi in ranges:
            if hi > lo:
                genfer_amd.conv_raw(x.data_ptr(), shape, y.data_ptr(), shape, z.data_ptr(), shape, lo, hi)
        L.gft_event_record(1)
        ms = float(L.gft_event_elapsed_ms(0, 1))
        best = ms if best is None else min(best, ms)
    return best


def macs(ranges):
    return sum(genfer_amd.conv_macs(shape, shape, shape, lo, hi) for lo, hi in ranges if hi > lo)


whole = [(0, n)]
t1 = timed(whole)
m1 = macs(whole)
rows = {"shape": shape, "whole_ms": t1, "whole_tmacs": m1 / t1 / 1e9, "worlds": {}}
print(f"{n}^{nd}: whole tensor {t1:.2f} ms = {m1 / t1 / 1e9:.2f} TMAC/s")
result_bytes = 8.0 * n**nd
for world in (2, 4, 8):
    per = []
    for r in range(world):
        (a, b), (c, d), even = genfer_amd.plan_slabs(n, world, r)
        ranges = [(a, d)] if b == c else [(a, b), (c, d)]
        ms = timed(ranges)
        mm = macs(ranges)
        per.append({"rank": r, "ranges": ranges, "ms": ms, "macs": mm, "tmacs": mm / ms / 1e9})
        print(f"  world {world} rank {r}: slabs {ranges}  {ms:8.2f} ms  {mm / ms / 1e9:6.2f} TMAC/s  ({mm / m1 * 100:5.2f} % of the work)")
    tmax = max(p["ms"] for p in per)
    tmean = sum(p["ms"] for p in per) / world
    comm_ms = result_bytes * (world - 1) / world / LINK * 1e3
    proj = t1 / (tmax + comm_ms)
    slowest = min(p["tmacs"] for p in per)
    rows["worlds"][world] = {"per_rank": per, "max_ms": tmax, "mean_ms": tmean, "max_over_mean": tmax / tmean, "exchange_ms_model": comm_ms,
                             "projected_speedup": proj, "projected_efficiency": proj / world, "slowest_rank_rate_vs_whole": slowest / (m1 / t1 / 1e9)}
    print(f"  world {world}: max {tmax:.2f} ms, mean {tmean:.2f} ms (max/mean {tmax / tmean:.3f}), exchange model {comm_ms:.2f} ms -> projected speed-up "
          f"{proj:.2f}x ({proj / world * 100:.1f} % efficiency); slowest rank's rate {slowest / (m1 / t1 / 1e9) * 100:.1f} % of the whole-tensor rate")
print(json.dumps(rows))
