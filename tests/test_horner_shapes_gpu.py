"""subst_var's Horner loop keeps the reference's STORED shapes bit-exactly, also when the accumulator cancels back to
an exactly linear polynomial in the middle of the loop (src/multivariate_taylor.rs:569-579: every step goes through
`Mul`'s dispatcher, mt:1052-1061, which asks whether the accumulator is linear and, if so, multiplies the other way
round — compacting the stored shape).  The device path speculates "not linear" once the accumulator has been seen
non-linear and verifies the speculation with per-step witnesses (gft_api.hip horner_speculative); these tests
CONSTRUCT the cancellation, so the verification must fail and the exact loop must take over.  Shapes and values are
compared with the oracle bit for bit, on the device tier and under the default size-threshold dispatch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(o, g):
    assert g.degrees_p1() == o.degrees_p1()
    assert g.coeffs_shape() == o.coeffs_shape(), (g.coeffs_shape(), o.coeffs_shape())
    assert np.array_equal(np.asarray(o.array()), np.asarray(g.array()), equal_nan=True)


def _case_linear_subst():
    # p(x, y) = sum_i a_i(y) x^i, x -> s = 1 + y.  Horner: res = a_3 = 1 + y + y^2 (non-linear), then
    # res*s + a_2 = (1 + 2y + 2y^2 + y^3) + (0, 0, -2, -1) = 1 + 2y: exactly linear again.
    a = np.zeros((4, 4))
    a[3, :3] = [1.0, 1.0, 1.0]
    a[2, :] = [0.0, 0.0, -2.0, -1.0]
    a[1, 0] = 1.0
    a[0, :2] = [0.5, 0.25]
    s = np.array([[1.0, 1.0]])
    return a, [6, 6], s, [6, 6]


def _case_general_subst():
    # x -> s = 1 + y + y^2 (not linear).  res = a_2 = 1 + y^2; res*s = 1 + y + 2y^2 + y^3 + y^4;
    # + a_1 = (0, 0, -2, -1, -1) gives 1 + y.
    a = np.zeros((3, 5))
    a[2, :3] = [1.0, 0.0, 1.0]
    a[1, :] = [0.0, 0.0, -2.0, -1.0, -1.0]
    a[0, :3] = [2.0, 0.0, 3.0]
    s = np.array([[1.0, 1.0, 1.0]])
    return a, [7, 7], s, [7, 7]


def _case_three_axes():
    # the same cancellation with a bystander axis z: coefficient slabs of shape (1, 4, 2)
    a = np.zeros((4, 4, 2))
    a[3, :3, 0] = [1.0, 1.0, 1.0]
    a[2, :, 0] = [0.0, 0.0, -2.0, -1.0]
    a[1, 0, 0] = 1.0
    a[1, 0, 1] = 0.5
    a[0, :2, 1] = [0.5, 0.25]
    s = np.array([[[1.0], [1.0]]])
    return a, [6, 6, 3], s, [6, 6, 3]


CASES = {"linear_subst": _case_linear_subst, "general_subst": _case_general_subst, "three_axes": _case_three_axes}


@pytest.mark.parametrize("case", sorted(CASES))
def test_accumulator_cancels_back_to_linear(case, OTP, GTP):
    a, adeg, s, sdeg = CASES[case]()
    o = OTP.new(a, adeg).subst_var(0, OTP.new(s, sdeg))
    g = GTP.new(a, adeg).subst_var(0, GTP.new(s, sdeg))
    _check(o, g)
    # the case really is the compacting one: a loop that never re-checks the accumulator would store explicit zeros
    # up to the naive extent (len_y of the top slab + one per remaining step, capped by degrees_p1)
    naive = min(adeg[1], int(np.max(np.nonzero(a.reshape(a.shape[0], a.shape[1], -1).any(axis=2)[-1])[0])) + 1 + (s.shape[1] - 1) * (a.shape[0] - 1))
    assert o.coeffs_shape()[1] < naive, (o.coeffs_shape(), naive)


@pytest.mark.parametrize("case", sorted(CASES))
def test_interval_accumulator_cancels_back_to_linear(case, OTPI, GTPI):
    a, adeg, s, sdeg = CASES[case]()
    ai, si = np.stack([a, a]), np.stack([s, s])  # point intervals: the same exact cancellation (0 + x and x*1 are exact)
    o = OTPI.new(ai, adeg).subst_var(0, OTPI.new(si, sdeg))
    g = GTPI.new(ai, adeg).subst_var(0, GTPI.new(si, sdeg))
    _check(o, g)


@pytest.mark.parametrize("seed", range(8))
def test_near_linear_accumulators_random(seed, OTP, GTP):
    """Sparse integer coefficients make accumulators that sit on unit positions of one or two axes (no witness) or
    cancel partly: whatever the verdict, shapes and values equal the oracle's."""
    rng = np.random.default_rng(seed)
    a = rng.integers(-1, 2, size=(5, 4, 3)).astype(float) * (rng.random((5, 4, 3)) < 0.25)
    a[4] = 0.0
    a[4, 0, 0] = 1.0
    a[4, 1, 0] = float(rng.integers(0, 2))
    a[4, 0, 1] = float(rng.integers(0, 2))
    s = np.zeros((1, 2, 2))
    s[0, 0, 0], s[0, 1, 0], s[0, 0, 1] = 1.0, 1.0, float(seed % 2)
    deg = [6, 5, 4]
    o = OTP.new(a, deg).subst_var(0, OTP.new(s, deg))
    g = GTP.new(a, deg).subst_var(0, GTP.new(s, deg))
    _check(o, g)
