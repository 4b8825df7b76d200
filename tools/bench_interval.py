"""Interval<F64> product (BASELINE configs[4]): interval multiply-adds per second against an explicit VALU-issue roof.

Roof: the positive-regime interval MAC (gft_elem.hpp mac_pos_unchecked) is 2 v_mul_f64 + 2 v_add_f64 + 4 integer steps
(`bits -/+ 1`, one v_lshl_add_u64 each) = 8 wave-instructions of 4 cycles ⇒ 64 lanes / 32 cycles = 2 interval MACs per
SIMD-cycle ⇒ 256 CU x 4 SIMD x 2 x 2.4 GHz = 4.9 interval-TMAC/s.  Mixed-sign data runs in the finite regime (round 3: no
operand can short-circuit, unguarded outward steps validated on the finished sum): 4 v_mul_f64 + 6 min / max + 2 v_add_f64 +
4 outward steps of 4 instructions = 28 wave-instructions ⇒ 64 / 112 MACs per SIMD-cycle ⇒ 1.40 interval-TMAC/s.
(4 cycles per wave-instruction is what tools/microbench_int64.hip measures for v_lshl_add_u64, v_mul_f64 and v_add_f64 at 8 waves
per SIMD: 4.4-4.6 cycles at the nominal 2.4 GHz; profiles/r03/microbench_int64.txt.)  Products of 3e5 multiply-adds and more (rank 2: 1e6; rows of at most 128)
run as row-pair sums (k_pair_sums + k_pair_collect, gft_conv_staged.hip) while their row sums fit 24 GiB (88^3); beyond, from 1.5e10
multiply-adds, on k_conv_rows_rb, else on k_conv_staged; GFT_RB_PAIRS=0 / GFT_CONV_RB=0 switch the first / second off.
Usage: bench_interval.py [n ...]"""
import sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import numpy as np
import genfer_amd

ROOF_POS, ROOF_GENERAL = 4.9, 1.40  # interval-TMAC/s
genfer_amd.init(0)
L = genfer_amd.lib()
TPI = genfer_amd.IntervalTaylorPoly
sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 80, 96, 112, 128]
for n in sizes:
    rng = np.random.default_rng(0)
    lo = rng.random((n, n, n)); x = np.stack([lo, lo * (1 + 1e-15)])
    lo = rng.random((n, n, n)); y = np.stack([lo, lo * (1 + 1e-15)])
    macs = (n * (n + 1) // 2) ** 3
    for tag, xx, yy, roof in (("positive data", x, y, ROOF_POS),
                              ("mixed-sign data", np.stack([x[0] - 0.5, x[0] - 0.5 + 1e-15]), np.stack([y[0] - 0.5, y[0] - 0.5 + 1e-15]), ROOF_GENERAL)):
        a, b = TPI.new(xx, (n,) * 3), TPI.new(yy, (n,) * 3)
        r = a * b
        L.gft_synchronize()
        L.gft_event_record(0)
        r = a * b
        L.gft_event_record(1)
        ms = L.gft_event_elapsed_ms(0, 1)
        rate = macs / ms / 1e9
        print(f"interval mul {n}^3 ({tag}): {ms:9.2f} ms  {rate:6.3f} interval-TMAC/s = {100 * rate / roof:5.1f} % of the {roof} TMAC/s issue roof", flush=True)
