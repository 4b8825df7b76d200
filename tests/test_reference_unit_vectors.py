"""The reference's own TaylorPoly unit tests (src/multivariate_taylor.rs:733-1513), restated
against the C-ABI mirror class.  Literal arrays live in tests/golden/unit_vectors.json.

`oracle` backend: every assertion is bit-exact, exactly as the reference's assert_eq! — this
is what pins the oracle.  `hip` backend (-m gpu): the same programs through the HIP library;
integer bookkeeping bit-exact, values within 1e-10 relative (bit-exact where every
intermediate is an exactly representable integer).
"""
import numpy as np
import pytest

from conftest import assert_poly_equal


def eq(got, want, backend, **kw):
    assert_poly_equal(got, want, backend, **kw)


def lit(arr):
    a = np.asarray(arr, dtype=np.float64)
    return (a, a.shape)


# mt:733-772
def test_2d_derivative(TP, backend, unit_vectors):
    d = unit_vectors["test_2d_derivative"]
    t = TP.taylor(d["taylor"])
    eq(t.derivative(0, 1), lit(d["d_var0_1"]), backend, exact_on_hip=True)
    eq(t.derivative(1, 1), lit(d["d_var1_1"]), backend, exact_on_hip=True)
    eq(t.derivative(0, 2), t.derivative(0, 1).derivative(0, 1), backend, exact_on_hip=True)
    eq(t.derivative(1, 2), t.derivative(1, 1).derivative(1, 1), backend, exact_on_hip=True)
    eq(t.derivative(0, 3), t.derivative(0, 1).derivative(0, 1).derivative(0, 1), backend, exact_on_hip=True)


# mt:775-803
def test_2d_taylor_expansion_of_coeff(TP, backend, unit_vectors):
    d = unit_vectors["test_2d_taylor_expansion_of_coeff"]
    t = TP.taylor(d["taylor"])
    eq(t.taylor_expansion_of_coeff(0, 2), lit(d["c_var0_2"]), backend, exact_on_hip=True)
    eq(t.taylor_expansion_of_coeff(1, 3), lit(d["c_var1_3"]), backend, exact_on_hip=True)
    eq(t.taylor_expansion_of_coeff(0, 2).taylor_expansion_of_coeff(1, 2), lit(d["c_both_2"]), backend, exact_on_hip=True)
    eq(t.taylor_expansion_of_coeff(1, 2).taylor_expansion_of_coeff(0, 2), lit(d["c_both_2"]), backend, exact_on_hip=True)


# mt:806-829
def test_2d_subst_var(TP, backend, unit_vectors):
    d = unit_vectors["test_2d_subst_var"]
    t, s = TP.taylor(d["taylor"]), TP.taylor(d["subst"])
    eq(t.subst_var(0, s), lit(d["subst_var0"]), backend, exact_on_hip=True)
    eq(t.subst_var(1, s), lit(d["subst_var1"]), backend, exact_on_hip=True)
    assert t.subst_var(0, s).subst_var(1, s) != t.subst_var(1, s).subst_var(0, s)


# mt:885-892
def test_add_mismatched_shapes(TP, backend):
    a, b = TP.var(0, 1.0, 5), TP.var(1, 1.0, 4)
    eq((a + b).extend([5, 4]), a.extend([5, 4]) + b.extend([5, 4]), backend, exact_on_hip=True)


# mt:940-947
def test_sub_mismatched_shapes(TP, backend):
    a, b = TP.var(0, 1.0, 5), TP.var(1, 1.0, 4)
    eq((a - b).extend([5, 4]), a.extend([5, 4]) - b.extend([5, 4]), backend, exact_on_hip=True)


# mt:1081-1094
def test_mul_mismatched_shapes(TP, backend):
    a, b = TP.var(0, 1.0, 5), TP.var(1, 1.0, 4)
    eq((a * b).extend([5, 4]), a.extend([5, 4]) * b.extend([5, 4]), backend, exact_on_hip=True)
    c = a * a * a
    d = b * b
    eq((c * d).extend([5, 4]), c.extend([5, 4]) * d.extend([5, 4]), backend, exact_on_hip=True)


# mt:1097-1102
def test_2d_mul(TP, backend, unit_vectors):
    d = unit_vectors["test_2d_mul"]
    eq(TP.taylor(d["f"]) * TP.taylor(d["g"]), lit(d["f_times_g"]), backend, exact_on_hip=True)


# mt:1105-1127
def test_2d_mul_const(TP, backend, unit_vectors):
    d, m = unit_vectors["test_2d_mul_const"], unit_vectors["test_2d_mul"]
    f, g = TP.taylor(d["f"]), TP.taylor(m["g"])
    eq(f * g, lit(m["f_times_g"]), backend, exact_on_hip=True)
    assert f * TP.zero() == TP.zero_with([2, 2])
    assert TP.zero() * f == TP.zero_with([2, 2])
    assert f * TP.one() == f
    assert TP.one() * f == f
    eq(TP.from_u32(2) * f, lit(d["two_f"]), backend, exact_on_hip=True)
    eq(f * TP.from_u32(2), lit(d["two_f"]), backend, exact_on_hip=True)


# mt:1130-1160
def test_2d_mul_factor_linear(TP, backend, unit_vectors):
    d = unit_vectors["test_2d_mul_factor_linear"]
    f = TP.taylor(d["f"])
    g0 = TP.from_u32(2) * TP.var_at_zero(0, 2)
    assert g0.extract_linear() == (0.0, 2.0, 0)
    g1 = TP.from_u32(3) * TP.var_at_zero(1, 2)
    assert g1.extract_linear() == (0.0, 3.0, 1)
    eq(f * g0, lit(d["f_g0"]), backend, exact_on_hip=True)
    eq(f * g1, lit(d["f_g1"]), backend, exact_on_hip=True)
    eq(g0 * f, lit(d["f_g0"]), backend, exact_on_hip=True)
    eq(g1 * f, lit(d["f_g1"]), backend, exact_on_hip=True)
    eq(g0 * g1, lit(d["g0_g1"]), backend, exact_on_hip=True)
    eq(g1 * g0, lit(d["g0_g1"]), backend, exact_on_hip=True)

    h0, h1 = TP.taylor(d["h0"]), TP.taylor(d["h1"])
    assert h0.extract_linear() == (3.0, 2.0, 0)
    assert h1.extract_linear() == (3.0, 2.0, 1)
    eq(f * h0, lit(d["f_h0"]), backend, exact_on_hip=True)
    eq(f * h1, lit(d["f_h1"]), backend, exact_on_hip=True)
    eq(h0 * f, lit(d["f_h0"]), backend, exact_on_hip=True)
    eq(h1 * f, lit(d["f_h1"]), backend, exact_on_hip=True)
    eq(h0 * h1, lit(d["h0_h1"]), backend, exact_on_hip=True)
    eq(h1 * h0, lit(d["h0_h1"]), backend, exact_on_hip=True)


# mt:1240-1253
def test_div_mismatched_shapes(TP, backend):
    a, b = TP.var(0, 1.0, 5), TP.var(1, 1.0, 4)
    eq((a / b).extend([5, 4]), a.extend([5, 4]) / b.extend([5, 4]), backend, exact_on_hip=True)
    c = a * a * a
    d = b * b
    eq((c * d).extend([5, 4]), c.extend([5, 4]) * d.extend([5, 4]), backend, exact_on_hip=True)


# mt:1256-1268
def test_2d_div(TP, backend, unit_vectors):
    d = unit_vectors["test_2d_div"]
    f, g = TP.taylor(d["f"]), TP.taylor(d["g"])
    r = f / g
    eq(r, lit(d["f_over_g"]), backend)
    eq(r * g, f, backend)


# mt:1389-1402
def test_exp_mismatched_shapes(TP, backend, unit_vectors):
    a = TP.var(0, 1.0, 5)
    eq(a.exp().extend([5, 4]), a.extend([5, 4]).exp(), backend)
    c = a * a * a
    eq(c.exp().extend([5, 4]), c.extend([5, 4]).exp(), backend)
    m = unit_vectors["mismatched_shapes"]
    a = TP.taylor(m["a_compact"], m["a_degrees"])
    eq(a.exp().extend([5, 4]), a.extend([5, 4]).exp(), backend)
    c = a * a * a
    eq(c.exp().extend([5, 4]), c.extend([5, 4]).exp(), backend)


# mt:1406-1437
def test_2d_exp(TP, backend, unit_vectors):
    d = unit_vectors["test_2d_exp"]
    assert TP.zero().exp() == TP.one()
    f, g = TP.taylor(d["f"]), TP.taylor(d["g"])
    eq(f.exp(), lit(d["exp_f"]), backend)
    # cancellation to exact zeros: compare against the magnitude of the terms being cancelled
    scale = np.abs((f.exp()).array()).max() ** 2
    eq(f.exp() * (-f).exp(), lit(d["exp_f_times_exp_neg_f"]), backend, scale=np.full((2, 2), scale))
    eq(f.exp() * g.exp(), lit(d["exp_f_times_exp_g"]), backend)
    eq((f + g).exp(), lit(d["exp_f_plus_g"]), backend)


# mt:1440-1453
def test_log_mismatched_shapes(TP, backend, unit_vectors):
    a = TP.var(0, 1.0, 5)
    eq(a.log().extend([5, 4]), a.extend([5, 4]).log(), backend)
    c = a * a * a
    eq(c.log().extend([5, 4]), c.extend([5, 4]).log(), backend)
    m = unit_vectors["mismatched_shapes"]
    a = TP.taylor(m["a_compact"], m["a_degrees"])
    eq(a.log().extend([5, 4]), a.extend([5, 4]).log(), backend)
    c = a * a * a
    eq(c.log().extend([5, 4]), c.extend([5, 4]).log(), backend)


# mt:1456-1513
def test_2d_log(TP, backend, unit_vectors):
    d = unit_vectors["test_2d_log"]
    assert TP.one().log() == TP.zero()
    xp1 = TP.var(0, 1.0, 5)
    eq(xp1.log(), lit(d["log_xp1"]), backend)
    e = TP.taylor(d["e"])
    eq(e.log(), lit(d["log_e"]), backend)
    eq(e.log().exp(), e, backend)
    f, g = TP.taylor(d["f"]), TP.taylor(d["g"])
    big = np.full((3, 3), 50.0)  # magnitude of the intermediate terms that cancel
    eq(f.log(), lit(d["log_f"]), backend, scale=big)
    eq(f.log().exp(), lit(d["f"]), backend, scale=big)
    eq(f.exp().log(), lit(d["exp_f_log"]), backend, scale=big)
    eq(f.log() + (TP.one() / f).log(), lit(d["zeros"]), backend, scale=big)
    eq(f.log() + g.log(), lit(d["log_f_plus_log_g"]), backend, scale=big)
    eq((f * g).log(), lit(d["log_fg"]), backend, scale=big)


# src/univariate_taylor.rs:479-578 — same recurrences at d = 1.  The univariate type is a
# different implementation (different association order), so these are 4-ulp cross-checks of
# the multivariate recurrences, not bit-exact pins.
def test_univariate_cross_checks(TP, backend, unit_vectors):
    d = unit_vectors["univariate"]

    def close(p, want):
        got = p.array()
        want = np.asarray(want)
        assert got.shape == want.shape
        assert np.all(np.abs(got - want) <= 4 * np.spacing(np.maximum(np.abs(want), 1e-3)))

    x = TP.var(0, 0.0, 10)
    close((x * x - TP.one()).exp(), d["exp_x2_minus_1"])
    close(x / (x - TP.one()), d["x_over_x_minus_1"])
    close(x / x.exp(), d["x_over_exp_x"])
    close(TP.one() / (x - TP.one()), d["one_over_x_minus_1"])
    close(TP.one() / x.exp(), d["one_over_exp_x"])
    close(TP.var(0, 1.0, 5).log(), d["log_1_plus_x"])
