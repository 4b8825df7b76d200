"""subst_var's Horner loop keeps the reference's STORED shapes bit-exactly, also when the accumulator cancels back to
an exactly linear polynomial in the middle of the loop (src/multivariate_taylor.rs:569-579: every step goes through
`Mul`'s dispatcher, mt:1052-1061, which asks whether the accumulator is linear and, if so, multiplies the other way
round — compacting the stored shape).  The device path speculates "not linear" once the accumulator has been seen
non-linear and verifies the speculation with per-step witnesses (gft_api.hip horner_speculative); these tests
CONSTRUCT the cancellation, so the verification must fail and the exact loop must take over.  Shapes and values are
compared with the oracle bit for bit, on the device tier and under the default size-threshold dispatch."""
import os

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


# ---- round 4: the general Horner step with a stencil substitution in ONE launch (K<E>::conv_shallow) --------------------
# `b +~ Binomial(a, p)` substitutes a -> a (1 - p + p b): a 2 x 2 coefficient tensor that is not linear, so every step of
# subst_var's loop is a GENERAL product of the accumulator with a handful of coefficients plus a slab (mt:569-579).  The
# fused kernel keeps the reference's loop nest per output and applies Add's operations to each finished sum, so the
# result is the oracle's bit for bit — also where the slab is a single coefficient (Add touches element 0 only), where it
# is larger than the product on some axis, and for intervals.
def _binomial_like(v, w, nd, p, x0=0.0, y0=0.0):
    """subst = (x0 + eps_v) * (1 - p + p * (y0 + eps_w)) minus its constant term, as a compact tensor"""
    s = np.zeros([2 if ax in (v, w) else 1 for ax in range(nd)])
    idx = lambda i, j: tuple((i if ax == v else (j if ax == w else 0)) for ax in range(nd))
    c = 1 - p + p * y0
    s[idx(0, 0)] = 0.0           # the evaluator strips the constant term (gf:609-627)
    s[idx(1, 0)] = c
    s[idx(0, 1)] = x0 * p
    s[idx(1, 1)] = p
    return s


FUSED_CASES = [
    # (shape of a, degrees, v, w, p)
    ((9, 7), [12, 12], 0, 1, 0.3),
    ((6, 5, 4), [9, 9, 6], 0, 1, 0.25),
    ((6, 5, 4), [9, 9, 6], 1, 2, 0.4),
    ((5, 4, 6), [7, 7, 7], 2, 0, 0.125),
    ((4, 3, 3, 4), [6, 5, 5, 6], 0, 3, 0.3),
    ((12, 1, 5), [14, 3, 8], 0, 2, 0.7),       # slabs with a unit axis; the substitution brings axis 2 in
    ((7, 3), [7, 3], 0, 1, 0.5),               # degrees cap the growth at once
    # stencils FLAT along the last axis with rows of even length: the two-outputs-per-thread form (shallow_pair_min = 0 below)
    ((6, 5, 4), [9, 9, 4], 0, 1, 0.3),
    ((5, 6, 3, 8), [7, 8, 3, 8], 1, 0, 0.3),
    ((8, 9, 10), [10, 11, 12], 0, 1, 0.6),
    ((6, 7, 2), [8, 8, 2], 1, 0, 0.45),
]


@pytest.mark.parametrize("interval", [False, True], ids=["f64", "interval"])
@pytest.mark.parametrize("case", range(len(FUSED_CASES)))
def test_general_horner_step_fused_bit_exact(case, interval, OTP, GTP, OTPI, GTPI):
    import genfer_amd

    shape, deg, v, w, p = FUSED_CASES[case]
    rng = np.random.default_rng(100 + case)
    a = rng.random(shape) * (rng.random(shape) < 0.8)   # some exact zeros (signed-zero and short-circuit paths)
    a[tuple(-1 for _ in shape)] = 0.75                   # the top slab is not empty
    s = _binomial_like(v, w, len(shape), p, x0=0.5, y0=0.25)
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    if interval:
        ai = np.stack([a * (1 - 1e-16), a * (1 + 1e-16)])
        si = np.stack([s * (1 - 1e-16), s * (1 + 1e-16)])
    else:
        ai, si = a, s
    L = genfer_amd.lib()
    want = O.new(ai, deg).subst_var(v, O.new(si, deg))
    results = {}
    for terms in (-1.0, 0.0):   # default (fused) / off (product, gather, add, witness as separate launches)
        assert L.gft_set_option(b"shallow_max_terms", terms) == 0
        try:
            before = genfer_amd.op_stats()
            g = G.new(ai, deg).subst_var(v, G.new(si, deg))
            _check(want, g)
            after = genfer_amd.op_stats()
            results[terms] = (after["fused_horner_steps"] - before["fused_horner_steps"], after["launches"] - before["launches"])
        finally:
            L.gft_set_option(b"shallow_max_terms", -1.0)
    # the fused step's two forms (one / two outputs per thread) on the device tier, whatever the size
    L.gft_set_option(b"host_max_elems", 0.0)
    try:
        for pair_min in (0.0, -2.0):
            assert L.gft_set_option(b"shallow_pair_min", pair_min) == 0
            _check(want, G.new(ai, deg).subst_var(v, G.new(si, deg)))
    finally:
        L.gft_set_option(b"shallow_pair_min", -1.0)
        L.gft_set_option(b"host_max_elems", -1.0)
    assert results[0.0][0] == 0
    # on the device tier the loop really took the fused kernel, and with fewer launches
    if results[-1.0][0]:
        assert results[-1.0][1] < results[0.0][1], results


def test_general_horner_fused_runs_on_the_device_tier(OTP, GTP):
    """A tensor above the host tier's size: the steps after the first non-linear accumulator are fused launches."""
    import genfer_amd

    rng = np.random.default_rng(7)
    a = rng.random((40, 40, 6))
    deg = [48, 48, 8]
    s = _binomial_like(0, 1, 3, 0.3, x0=1.0, y0=1.0)
    before = genfer_amd.op_stats()
    g = GTP.new(a, deg).subst_var(0, GTP.new(s, deg))
    after = genfer_amd.op_stats()
    _check(OTP.new(a, deg).subst_var(0, OTP.new(s, deg)), g)
    # (under the default dispatch the first ~18 accumulators are host-resident: those steps run on the host tier)
    if os.environ.get("GFT_SHALLOW_MAX_TERMS") != "0" and os.environ.get("GFT_FUSE_HORNER") != "0":  # (the verification matrix switches the path off: values only then)
        assert after["fused_horner_steps"] - before["fused_horner_steps"] >= 15, (before, after)


@pytest.mark.parametrize("interval", [False, True], ids=["f64", "interval"])
@pytest.mark.parametrize("xs,ys,deg", [((30, 28, 9), (2, 2, 1), [31, 29, 9]), ((30, 28, 9), (1, 2, 2), [31, 29, 10]),
                                        ((1, 1, 24), (24, 24, 1), [24, 24, 24]), ((24, 1, 1), (24, 24, 24), [24, 24, 24]),
                                        ((40, 50), (3, 2), [41, 50]), ((5, 4, 3, 20), (2, 1, 2, 3), [6, 4, 4, 20]),
                                        # flat along the last axis, rows of even length: two outputs per thread (16-byte accesses)
                                        ((30, 28, 10), (2, 2, 1), [31, 29, 10]), ((90, 50), (3, 1), [92, 50]), ((30, 28, 10), (2, 2, 1), [30, 28, 10]),
                                        ((12, 11, 6, 8), (2, 1, 2, 1), [13, 11, 7, 8]), ((26, 24, 12), (2, 3, 1), [40, 40, 12]),
                                        # not flat: every x element of the last axis loaded once for the pair
                                        ((70, 69), (2, 2), [71, 70]), ((30, 28, 9), (1, 2, 2), [31, 29, 10]), ((64, 80), (2, 3), [64, 80]), ((20, 21, 13), (2, 1, 4), [21, 21, 16])])
def test_shallow_products_bit_exact(xs, ys, deg, interval, OTP, GTP, OTPI, GTPI, tier):
    """Products whose outputs receive few terms each (one operand a stencil, or an outer product) run on the reference-
    order kernel: bit-exact against the oracle, both operand orders."""
    import genfer_amd

    rng = np.random.default_rng(3)
    x, y = rng.random(xs) - 0.3, rng.random(ys) - 0.3
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    if interval:
        x, y = np.stack([x - 1e-3, x + 1e-3]), np.stack([y - 1e-3, y + 1e-3])
    before = genfer_amd.op_stats()["shallow_products"]
    _check(O.new(x, deg) * O.new(y, deg), G.new(x, deg) * G.new(y, deg))
    _check(O.new(y, deg) * O.new(x, deg), G.new(y, deg) * G.new(x, deg))
    if tier == "device" and os.environ.get("GFT_SHALLOW_MAX_TERMS") is None:
        assert genfer_amd.op_stats()["shallow_products"] == before + 2, "the products did not take the shallow kernel"
    # the kernel's two forms — one output per thread, two neighbours along the last axis (rows of even length) — whatever the size
    L = genfer_amd.lib()
    try:
        for pair_min in (0.0, -2.0):
            assert L.gft_set_option(b"shallow_pair_min", pair_min) == 0
            _check(O.new(x, deg) * O.new(y, deg), G.new(x, deg) * G.new(y, deg))
            _check(O.new(y, deg) * O.new(x, deg), G.new(y, deg) * G.new(x, deg))
    finally:
        L.gft_set_option(b"shallow_pair_min", -1.0)


@pytest.mark.parametrize("interval", [False, True], ids=["f64", "interval"])
@pytest.mark.parametrize("xs,ys,deg", [((100, 1, 1), (100, 40, 30), [100, 40, 30]),     # three_populations' product, truncated at the operands' length
                                        ((70, 1), (50, 33), [90, 33]),                   # rank 2, the result shorter than the full product
                                        ((1, 40, 1), (12, 40, 20), [12, 60, 20]),        # a line along the MIDDLE axis
                                        ((200, 1), (64, 40), [220, 40]),                 # a long line: 32-column tiles
                                        ((60, 1, 1), (20, 8, 8), [60, 8, 8]),            # a line longer than the other operand, the result cut at the line's length
                                        ((17, 1, 1, 1), (5, 3, 4, 6), [21, 3, 4, 6])])  # rank 4, the shortest line the form takes
def test_line_products_bit_exact(xs, ys, deg, interval, OTP, GTP, OTPI, GTPI, tier):
    """A product one operand of which is a line along an outer axis (round 5, k_conv_line: a tile of the other operand in LDS,
    the reference's order per output): bit-exact against the oracle in both operand orders — the line as x (ascending line
    index) and as y (descending) — f64 and interval, with signed zeros and a non-finite value in the operands."""
    rng = np.random.default_rng(5)
    x, y = rng.random(xs) - 0.3, rng.random(ys) - 0.3
    x.flat[::7] = 0.0
    y.flat[::5] = -0.0
    y[tuple(min(1, n - 1) for n in ys)] = np.inf
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    if interval:
        x, y = np.stack([x - 1e-3, x + 1e-3]), np.stack([y - 1e-3, y + 1e-3])
    _check(O.new(x, deg) * O.new(y, deg), G.new(x, deg) * G.new(y, deg))
    _check(O.new(y, deg) * O.new(x, deg), G.new(y, deg) * G.new(x, deg))


# ---- round 5: recorded operations -------------------------------------------------------------------------------
def rand(shape, seed, lo=0.0, hi=1.0):
    from conftest import splitmix64_uniform

    return (lo + (hi - lo) * splitmix64_uniform(seed, int(np.prod(shape)))).reshape(shape)


def check(o, g):
    _check(o, g)


def _if_statement(T, g, v, x, cs, m, half, deg_p1):
    """One arm of `if .. {observe k ~ Poisson(l * X_v)}`: the observation chain on the predecessor g, the scaling substitution
    X_v -> m * X_v (`subst - constant_term(subst)`: for intervals a loop with a few-ulp constant, for f64 a power table), the
    branch probability."""
    obs = g.observe_chain(v, x, cs, deg_p1)
    sub = T.var_with_degrees_p1(v, x, [deg_p1] * 2) * T.from_scalar(m)
    sub = sub - T.from_scalar(sub.constant_term())
    return obs.subst_var(v, sub) * T.from_scalar(half)


@pytest.mark.parametrize("interval", [False, True])
def test_recorded_chains_epilogues_and_riders_bit_exact(interval, OTP, GTP, OTPI, GTPI, tier):
    """A ladder of mixture-style `if`s through the handle API: observation chains are RECORDED and launched with the
    Add of the two arms as their epilogue, a recorded chain on an old predecessor rides along with a later observation launch,
    and with intervals the Horner loops' linearity scans are answered by the "no exact zero" proof and recorded loops ride on
    other loops' launches.  Every result must equal the oracle's bit for bit, with each switch on and off, and the counters
    must show that the fused forms were really taken."""
    import genfer_amd

    L = genfer_amd.lib()
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    n = 64  # (4096 coefficients: device tensors under the default dispatch too)
    base = rand((n, n), 501, 0.05, 1.0)
    arr = np.stack([base, base * (1 + 1e-15)]) if interval else base
    sc = (lambda v: (v, v)) if interval else (lambda v: v)

    def run(T, levels=4):
        g = [T.new(arr, [n, n])]
        for lvl in range(levels):  # the predecessor of the first arm is one level OLDER than that of the second: old news to the main chain
            d = n - 3 * (lvl + 1)
            a1 = _if_statement(T, g[max(0, len(g) - 2)].truncate_to_degree_p1(d + 2), 0, sc(0.9), [sc(0.1), sc(0.05)], sc(0.9048374180359595), sc(0.5), d)
            a2 = _if_statement(T, g[-1].truncate_to_degree_p1(d + 2), 1, sc(0.8), [sc(0.1), sc(0.05)], sc(0.9048374180359595), sc(0.5), d)
            g.append(a1 + a2)
        return g[-1]

    want = run(O)
    # (tier "host" = the default dispatch, under which the substitutions `m * (x + eps_v) - m * x` stay lazy host handles as in
    # the interpreter's runs; tier "device" materialises them, which launches the recordings early: same bits, no fusion)
    try:
        for opts in ({}, {"lazy_observe": 0}, {"batch_dag": 0}, {"lazy_sum": 0}, {"lazy_horner": 0}, {"nz_proofs": 0},
                     {"lazy_observe": 0, "lazy_sum": 0, "lazy_horner": 0, "nz_proofs": 0}, {"batch_dag": 0, "lazy_sum": 0}):
            for k, v in opts.items():
                assert L.gft_set_option(k.encode(), float(v)) == 0
            try:
                before = genfer_amd.op_stats()
                got = run(G)
                check(want, got)
                after = genfer_amd.op_stats()
                d = {k: after[k] - before[k] for k in after}
                switched = any(os.environ.get(k) for k in ("GFT_BATCH", "GFT_LAZY_OBSERVE", "GFT_LAZY_HORNER", "GFT_LAZY_SUM", "GFT_NZ_PROOFS",
                                                           "GFT_DEFER", "GFT_HORNER_LOOP_MAX"))
                if not opts and tier == "host" and not switched:  # (the verification matrix runs this test under every switch: bits only)
                    if not interval:
                        assert d["fused_observe_adds"] >= 3, d   # the Adds ran as the observation kernels' epilogues
                    if interval:
                        assert d["scans_proven"] >= 4 and d["linear_scans"] == 0, d  # no scan, no round trip
                if opts.get("lazy_observe") == 0:
                    assert d["fused_observe_adds"] == 0, d
            finally:
                for k in opts:
                    L.gft_set_option(k.encode(), 1.0)
    finally:
        pass


@pytest.mark.parametrize("interval", [False, True])
def test_recorded_sums_nested_add_bit_exact(interval, OTP, GTP, OTPI, GTPI):
    """hmm's `if`: both arms end in mul_linear (State ~ Bernoulli(p): c * t + m * shift(t), a two-chain Add) and the merge adds
    them.  The arms' Adds are RECORDED and the merge evaluates all three in one launch (K<E>::chain_nest); anybody else who
    reads a recorded sum launches it.  Bit for bit the oracle's result either way."""
    import genfer_amd

    L = genfer_amd.lib()
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    shape = (2, 48, 48)  # (4608 coefficients: device tensors under the default dispatch too)
    a, b = rand(shape, 511, 0.1, 1.0), rand(shape, 512, 0.1, 1.0)
    mk = (lambda t: np.stack([t, t * (1 + 1e-15)])) if interval else (lambda t: t)
    sc = (lambda v: (v, v)) if interval else (lambda v: v)
    deg = [2, 48, 48]

    def run(T):
        ta, tb = T.new(mk(a), deg), T.new(mk(b), deg)
        lin1 = T.var_with_degrees_p1(0, sc(0.0), deg) * T.from_scalar(sc(0.2)) + T.from_scalar(sc(0.8))   # 0.8 + 0.2 x_0
        lin2 = T.var_with_degrees_p1(0, sc(0.0), deg) * T.from_scalar(sc(0.7)) + T.from_scalar(sc(0.3))
        s1, s2 = ta * lin1, tb * lin2          # mul_linear: recorded two-chain Adds
        merged = s1 + s2                       # the nested Add
        again = s1 - tb                        # a second consumer of a recording (plain operand on the right)
        return merged, again, s2

    want = run(O)
    try:
        for lazy in (1.0, 0.0):
            assert L.gft_set_option(b"lazy_sum", lazy) == 0
            before = genfer_amd.op_stats()
            got = run(G)
            for w, g_ in zip(want, got):
                check(w, g_)
            d = genfer_amd.op_stats()
            if lazy and not any(os.environ.get(k) for k in ("GFT_BATCH", "GFT_LAZY_SUM", "GFT_DEFER")):
                assert d["nested_adds"] - before["nested_adds"] >= 2, (d, before)
    finally:
        L.gft_set_option(b"lazy_sum", 1.0)


@pytest.mark.parametrize("interval", [False, True])
def test_dependent_recordings_are_issued_after_their_producer(interval, OTP, GTP, OTPI, GTPI, tier):
    """A recording whose INPUT is another recording (B = observe(A), r2 = subst_var(r1)) must never share a launch with the one
    that writes that input (round-5 advisor finding on the riders of that round; in the launch graph of round 6 the two are on
    different levels by construction).  Reading the producer first, then the dependents, must give the oracle's bits — with
    the graph on and off."""
    import genfer_amd

    L = genfer_amd.lib()
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    n = 64
    base = rand((n, n), 521, 0.05, 1.0)
    arr = np.stack([base, base * (1 + 1e-15)]) if interval else base
    sc = (lambda v: (v, v)) if interval else (lambda v: v)

    def run(T):
        out = []
        g = T.new(arr, [n, n])
        a = g.observe_chain(0, sc(0.9), [sc(0.1), sc(0.05)], n - 4)      # recorded
        b = a.observe_chain(1, sc(0.8), [sc(0.2), sc(0.07)], n - 8)      # recorded on a recording
        c = b.observe_chain(0, sc(0.7), [sc(0.3)], n - 10)               # ... and one more level
        out += [a.array().copy(), b.array().copy(), c.array().copy()]    # producer first, then the dependents
        # nested linear Horner loops on an operand that is old news to the main chain (recorded when proven: intervals)
        p = T.new(arr, [n, n])
        for _ in range(3):
            _ = (g + g).constant_term()                                  # unrelated stream operations: p ages
        lin = T.var_with_degrees_p1(0, sc(0.25), [n, n]) * T.from_scalar(sc(0.5))
        r1 = p.subst_var(0, lin)
        r2 = r1.subst_var(0, lin)
        r3 = r2.subst_var(1, T.var_with_degrees_p1(1, sc(0.125), [n, n]) * T.from_scalar(sc(0.75)))
        out += [r1.array().copy(), r2.array().copy(), r3.array().copy()]
        return out

    want = run(O)
    try:
        for opts in ({}, {"batch_dag": 0}):
            for k, v in opts.items():
                assert L.gft_set_option(k.encode(), float(v)) == 0
            got = run(G)
            for i, (w, g_) in enumerate(zip(want, got)):
                assert w.shape == g_.shape, (i, w.shape, g_.shape)
                assert np.array_equal(w, g_), (opts, i, float(np.max(np.abs(w - g_))))
    finally:
        L.gft_set_option(b"batch_dag", 1.0)


@pytest.mark.parametrize("interval", [False, True])
def test_launch_graph_batches_bit_exact(interval, OTP, GTP, OTPI, GTPI, tier):
    """The deferred launch graph: B independent operations of one kind on same-shape tensors — what the evaluator
    issues for the B input points of one depth — are recorded and issued as ONE launch per kind and level (blockIdx.y = item).
    Every item must carry the bits of the oracle's own sequence for that item, with the graph on and off, and with it on the
    counters must show batches: observation chains (plain, and with the Add of two arms as epilogue), two-chain Adds, nested
    Adds of mul_linear sums, and (intervals) proven linear Horner loops."""
    import genfer_amd

    L = genfer_amd.lib()
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    n, B = 48, 9
    base = rand((n, n), 601, 0.05, 1.0)
    other = rand((n, n), 602, 0.05, 1.0)
    mk = (lambda t: np.stack([t, t * (1 + 1e-15)])) if interval else (lambda t: t)
    sc = (lambda v: (v, v)) if interval else (lambda v: v)
    xs = [0.3 + 0.07 * k for k in range(B)]

    def run(T):
        g, h = T.new(mk(base), [n, n]), T.new(mk(other), [n, n])
        out = []
        # level 1: B plain observation chains on g, B on h (two axes)
        a = [g.observe_chain(0, sc(x), [sc(0.1), sc(0.05)], n - 4) for x in xs]
        b = [h.observe_chain(1, sc(x), [sc(0.2), sc(0.07)], n - 4) for x in xs]
        # level 2: a chain on each result with the Add of the two arms as its epilogue (the `if` of mixture)
        c = [a[k].observe_chain(0, sc(0.9), [sc(0.3)], n - 6) * T.from_scalar(sc(0.5)) +
             b[k].observe_chain(1, sc(0.8), [sc(0.4)], n - 6) * T.from_scalar(sc(0.5)) for k in range(B)]
        # two-chain Adds of scaled tensors
        d = [a[k] * T.from_scalar(sc(0.25)) + b[k] * T.from_scalar(sc(0.75)) for k in range(B)]
        # observation chains whose INPUT is a deferred chain (a scaled tensor): its materialisation is a node of the graph too
        # (SettleRec: hmm's `if` arms) — B of them are one launch, then the B chains another
        s_ = [(a[k] * T.from_scalar(sc(0.6))).observe_chain(0, sc(0.7), [sc(0.2)], n - 8) for k in range(B)]
        # nested Adds of mul_linear sums (hmm's `State ~ Bernoulli(p)` on both arms)
        lin1 = T.var_with_degrees_p1(0, sc(0.0), [n, n]) * T.from_scalar(sc(0.2)) + T.from_scalar(sc(0.8))
        lin2 = T.var_with_degrees_p1(0, sc(0.0), [n, n]) * T.from_scalar(sc(0.7)) + T.from_scalar(sc(0.3))
        e = [a[k] * lin1 + b[k] * lin2 for k in range(B)]
        # linear Horner loops (recorded where the loop is proven: intervals with a non-zero constant)
        # (f64: an unproven loop reads its verdict back — it issues its operand's recordings where it stands; built AFTER the root)
        mkf = lambda: [c[k].subst_var(0, T.var_with_degrees_p1(0, sc(0.25 + 0.01 * k), [n, n]) * T.from_scalar(sc(0.5))) for k in range(B)]
        f = mkf() if interval else None
        # ONE value that depends on every item — as the program's result depends on every input point —, read first: the graph
        # executes level by level across the items; the items themselves are in memory afterwards and are compared one by one
        def total(lst):
            r = lst[0]
            for t in lst[1:]:
                r = r + t
            return r
        root = total(c + d + e)
        out.append(root.array().copy())
        out.append(total(s_).array().copy())
        if f is None:
            f = mkf()
        froot = total(f)
        out.append(froot.array().copy())
        for lst in (c, d, e, s_, f):
            out += [t.array().copy() for t in lst]
        return out

    want = run(O)
    try:
        for batch in (1.0, 0.0):
            assert L.gft_set_option(b"batch_dag", batch) == 0
            before = genfer_amd.op_stats()
            got = run(G)
            for i, (w, g_) in enumerate(zip(want, got)):
                assert w.shape == g_.shape, (batch, i, w.shape, g_.shape)
                assert np.array_equal(w, g_), (batch, i, float(np.max(np.abs(w - g_))))
            d = genfer_amd.op_stats()
            delta = {k: d[k] - before[k] for k in d}
            # (tier "host" = the default dispatch: scalars and affine substitutions stay host values, as in the interpreter's runs; with
            # everything forced onto the device their products read values back, which issues the recordings one by one)
            if batch and tier == "host" and not any(os.environ.get(k) for k in ("GFT_BATCH", "GFT_LAZY_OBSERVE", "GFT_LAZY_SUM", "GFT_LAZY_HORNER", "GFT_DEFER", "GFT_NZ_PROOFS")):
                assert delta["batch_launches"] >= 4 and delta["batch_items"] >= 4 * B, delta
                assert delta["chains_materialised"] >= B, delta  # (the B recorded materialisations)
            if not batch:
                assert delta["batch_launches"] == 0, delta
    finally:
        L.gft_set_option(b"batch_dag", 1.0)
