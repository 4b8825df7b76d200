"""Independent pins for Interval<F64> arithmetic (the reference holds no test vector for `--bounds`).

1. Exact-rational enclosure KATs (tests/golden/exact_kats.json kind "iv_chain", generated with fractions.Fraction by
   tests/golden/make_exact_kats.py, independent of every Interval implementation): q = (x*y + w)/d evaluated on point
   intervals must enclose the exact coefficients and stay within ~1e-12 relative width.  Run on the oracle (CPU) and on
   the HIP library (GPU, both tiers of the size-threshold dispatch).
2. Three restatements of src/interval.rs live in this repository — the kernels' element functor (gft_elem.hpp), the
   interpreter's own number type (genfer_amd/csrc/host/gfh_number.hpp) and the test oracle.  They were written
   separately; they must agree BIT FOR BIT on a grid of f64 edge values: oracle vs interpreter here on the CPU,
   HIP vs oracle in test_parity_gpu.py::test_interval_edge_values_bit_exact and below for scalar handles."""
import ctypes
import itertools
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN

INF, NAN = math.inf, math.nan
EDGE = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, 3.0, -3.0, 0.1, -0.1, 1.0 + 2.0**-52, 1.0 - 2.0**-53, 5e-324, -5e-324, 2.2250738585072014e-308,
        1.7976931348623157e308, -1.7976931348623157e308, INF, -INF, NAN]


def edge_intervals():
    ivs = []
    for lo, hi in itertools.product(EDGE, EDGE):
        if math.isnan(lo) or math.isnan(hi):
            if math.isnan(lo) and math.isnan(hi):
                ivs.append((lo, hi))
            continue
        if lo <= hi:
            ivs.append((lo, hi))
    return ivs


def _bits_equal(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(GOLDEN, "exact_kats.json")) as f:
        return [c for c in json.load(f)["cases"] if c["kind"] == "iv_chain"]


def _scalar_op(fn, op, a, b=None):
    A = (ctypes.c_double * 2)(*a)
    B = (ctypes.c_double * 2)(*b) if b is not None else None
    out = (ctypes.c_double * 2)()
    assert fn(op, A, B, out) == 0
    return out[0], out[1]


def test_oracle_and_interpreter_interval_agree_bitwise(oracle_lib):
    """add sub mul div neg exp log of oracle::Interval (oracle/taylor_oracle.hpp) and gfh::Interval (gfh_number.hpp)."""
    import genfer_amd

    H = genfer_amd.host_lib()
    for lib, name in ((oracle_lib, "orci_scalar_op"), (H, "gfh_interval_op")):
        f = getattr(lib, name)
        f.restype, f.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    ivs = edge_intervals()
    n = 0
    for a in ivs:
        for op in (4, 5, 6):
            assert _bits_equal(_scalar_op(oracle_lib.orci_scalar_op, op, a), _scalar_op(H.gfh_interval_op, op, a)), (op, a)
        for b in ivs:
            for op in (0, 1, 2, 3):
                ro, rh = _scalar_op(oracle_lib.orci_scalar_op, op, a, b), _scalar_op(H.gfh_interval_op, op, a, b)
                assert _bits_equal(ro, rh), (op, a, b, ro, rh)
                n += 1
    assert n > 50000


def _chain(T, c):
    deg = c["deg"]

    def pt(name):
        a = np.asarray(c[name], dtype=np.float64)
        return T.new(np.stack([a, a]), deg)

    return ((pt("x") * pt("y") + pt("w")) / pt("d")).array()


def _assert_encloses(r, c):
    lo, hi = np.asarray(r[0]), np.asarray(r[1])
    down, up, mag = (np.asarray(c[k], dtype=np.float64) for k in ("r_down", "r_up", "mag"))
    assert lo.shape == down.shape
    assert np.all(lo <= down) and np.all(up <= hi), "the interval result does not enclose the exact rational coefficient"
    assert np.all(hi - lo <= 1e-12 * mag), np.max((hi - lo) / mag)
    assert np.any(hi - lo > 0)  # widened somewhere: this really is interval arithmetic, not a point evaluation


def test_oracle_interval_chain_encloses_exact_rationals(kats, OTPI):
    assert len(kats) >= 4
    for c in kats:
        _assert_encloses(_chain(OTPI, c), c)


@pytest.mark.gpu
def test_hip_interval_chain_encloses_exact_rationals(kats, GTPI, OTPI):
    for c in kats:
        r = _chain(GTPI, c)
        _assert_encloses(r, c)
        assert _bits_equal(r, _chain(OTPI, c))  # and is the oracle's interval, bit for bit


@pytest.mark.gpu
def test_hip_interval_scalar_handles_equal_oracle_on_edge_values(GTPI, OTPI):
    """1-element (x) 1-element operations are evaluated on the host by the kernels' own element functor
    (gft_elem.hpp, one source for both sides): same bits as the oracle's handle operations, edge values included."""
    ivs = edge_intervals()[::3]
    for a in ivs:
        ga, oa = GTPI.from_scalar(a), OTPI.from_scalar(a)
        assert _bits_equal((-ga).constant_term(), (-oa).constant_term())
        for b in ivs:
            gb, ob = GTPI.from_scalar(b), OTPI.from_scalar(b)
            for f in (lambda p, q: p + q, lambda p, q: p - q, lambda p, q: p * q, lambda p, q: p / q):
                assert _bits_equal(f(ga, gb).constant_term(), f(oa, ob).constant_term()), (a, b)
