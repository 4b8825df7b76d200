"""Seeded random expression trees over the TaylorPoly API on the GPU vs the CPU oracle, bit for bit.

Every tree mixes constructors (dense tensors, scalars, variables — i.e. lazy handles), algebra (+ - * / neg pow),
structure ops (derivative, shift_down, truncation, extend_to_dim, coefficients_of_term, subst_var with dense, linear
and scaled-variable substitutions) and the value inspections the interpreter performs in between (constant_term,
extract_linear, is_zero) — the places where laziness, memoised verdicts and the fused paths interact.  Sizes stay
below the tiled crossover, so everything is reference-order arithmetic and the comparison is exact (NaN == NaN)."""
import os

import numpy as np
import pytest

from conftest import splitmix64_uniform

NV = 3  # variables


def _rand(shape, seed, lo=-1.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * splitmix64_uniform(seed, n)).reshape(shape)


class OracleRejected(Exception):
    """The oracle refused the operation (a reference assertion): the tree is abandoned, nothing to compare."""


def both(fo, fg):
    """Apply the oracle's op first; if it raises, abandon the tree; then the GPU op, which must NOT raise."""
    try:
        ro = fo()
    except Exception as e:  # noqa: BLE001 - TaylorError of the oracle binding
        raise OracleRejected(str(e))
    return ro, fg()


class Gen:
    def __init__(self, seed, interval):
        self.rng = np.random.default_rng(seed)
        self.seed = seed * 1000
        self.interval = interval

    def scal(self, v):
        return (v, v + abs(v) * 1e-9) if self.interval and self.rng.random() < 0.5 else ((v, v) if self.interval else v)

    def leaf(self, O, G):
        r = self.rng.integers(0, 5)
        deg = [int(self.rng.integers(3, 7)) for _ in range(NV)]
        if r == 0:
            shape = tuple(int(self.rng.integers(1, d + 1)) for d in deg)
            self.seed += 1
            a = _rand(shape, self.seed)
            if self.rng.random() < 0.3:
                a[tuple(0 for _ in shape)] = [0.0, 1.0, -1.0][int(self.rng.integers(0, 3))]
            arr = np.stack([a, a + np.abs(a) * 1e-9]) if self.interval else a
            return both(lambda: O.new(arr, deg), lambda: G.new(arr, deg))
        if r == 1:
            v = [0.0, 1.0, -1.0, 0.5, -2.25, 3.0][int(self.rng.integers(0, 6))]
            s = self.scal(v)
            return both(lambda: O.from_scalar(s), lambda: G.from_scalar(s))
        if r == 2:
            v = int(self.rng.integers(0, NV))
            x0 = [0.0, 0.0, 1.0, -0.5][int(self.rng.integers(0, 4))]
            s = self.scal(x0)
            return both(lambda: O.var_with_degrees_p1(v, s, deg), lambda: G.var_with_degrees_p1(v, s, deg))
        if r == 3:
            v = int(self.rng.integers(0, NV))
            n = int(self.rng.integers(2, 7))
            return both(lambda: O.var_at_zero(v, n), lambda: G.var_at_zero(v, n))
        v = int(self.rng.integers(0, NV))
        s = self.scal([0.0, 0.75][int(self.rng.integers(0, 2))])
        n = int(self.rng.integers(1, 7))
        return both(lambda: O.var(v, s, n), lambda: G.var(v, s, n))

    def tree(self, O, G, depth):
        if depth == 0 or self.rng.random() < 0.15:
            return self.leaf(O, G)
        op = int(self.rng.integers(0, 16))
        ao, ag = self.tree(O, G, depth - 1)
        if op <= 3:
            bo, bg = self.tree(O, G, depth - 1)
            if op == 0:
                return both(lambda: ao + bo, lambda: ag + bg)
            if op == 1:
                return both(lambda: ao - bo, lambda: ag - bg)
            if op == 2:
                return both(lambda: ao * bo, lambda: ag * bg)
            # division by something with a safely non-zero constant term
            c = self.scal(2.5)
            return both(lambda: ao / (bo * bo + O.from_scalar(c)), lambda: ag / (bg * bg + G.from_scalar(c)))
        nv = ao.num_vars()
        v = int(self.rng.integers(0, max(nv, 1)))
        if op == 4:
            return both(lambda: -ao, lambda: -ag)
        if op == 5:
            return both(lambda: ao.pow(2), lambda: ag.pow(2))
        if op == 6 and nv:
            n = int(self.rng.integers(0, 2))
            if n < ao.len_of(v):
                return both(lambda: ao.derivative(v, n), lambda: ag.derivative(v, n))
        if op == 7 and nv:
            if 1 < ao.len_of(v):
                return both(lambda: ao.shift_down(v, 1), lambda: ag.shift_down(v, 1))
        if op == 8:
            d = int(self.rng.integers(1, 6))
            return both(lambda: ao.truncate_to_degree_p1(d), lambda: ag.truncate_to_degree_p1(d))
        if op == 9 and nv:
            # the interpreter's Subst arm: read the constant, subtract it, substitute
            so, sg = self.tree(O, G, depth - 1)
            co, cg = so.constant_term(), sg.constant_term()
            assert _same_scalar(co, cg)
            so, sg = both(lambda: so - O.from_scalar(co), lambda: sg - G.from_scalar(cg))
            return both(lambda: ao.subst_var(v, so), lambda: ag.subst_var(v, sg))
        if op == 10 and nv:
            k = int(self.rng.integers(0, 3))
            return both(lambda: ao.coefficients_of_term(v, k), lambda: ag.coefficients_of_term(v, k))
        if op == 11 and nv:
            c = self.scal([0.5, -1.5, 1.0, 0.0][int(self.rng.integers(0, 4))])
            x0 = self.scal([0.0, 0.25][int(self.rng.integers(0, 2))])
            n = max(2, int(ao.len_of(v)) if ao.len_of(v) < 64 else 4)
            lo, lg = both(lambda: O.var(v, x0, n) * O.from_scalar(c), lambda: G.var(v, x0, n) * G.from_scalar(c))
            assert lo.extract_linear() == lg.extract_linear() or _nan_pair(lo.extract_linear(), lg.extract_linear())
            return both(lambda: ao.subst_var(v, lo), lambda: ag.subst_var(v, lg))
        if op == 12 and nv:
            orders = sorted({int(t) for t in self.rng.integers(0, 4, size=2)})
            return both(lambda: ao.taylor_polynomial_terms(v, orders), lambda: ag.taylor_polynomial_terms(v, orders))
        if op == 13:
            d = int(self.rng.integers(2, 6))
            return both(lambda: ao.extend_to_dim(nv + 1, d), lambda: ag.extend_to_dim(nv + 1, d))
        if op == 14 and nv:
            return both(lambda: ao.remove_last_variable(), lambda: ag.remove_last_variable())
        if op == 15 and nv:
            n, d = int(self.rng.integers(0, 3)), int(self.rng.integers(1, 6))
            if n < ao.len_of(v):
                return both(lambda: ao.derivative(v, n).truncate_to_degree_p1(d), lambda: ag.derivative_truncated(v, n, d))
        return ao, ag


def _nan_pair(a, b):
    return str(a) == str(b)


def _same_scalar(a, b):
    a, b = np.atleast_1d(np.asarray(a, dtype=float)), np.atleast_1d(np.asarray(b, dtype=float))
    return a.shape == b.shape and bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b)) | (a == b)))


def _check(o, g):
    assert o.shape() == g.shape() and o.degrees_p1() == g.degrees_p1(), (o.shape(), g.shape(), o.degrees_p1(), g.degrees_p1())
    a, b = np.asarray(o.array()), np.asarray(g.array())
    ok = (a == b) | (np.isnan(a) & np.isnan(b))
    assert np.all(ok), (a[~ok][:4], b[~ok][:4])


@pytest.mark.gpu
@pytest.mark.parametrize("interval", [False, True])
@pytest.mark.parametrize("seed", range(1, 1 + int(os.environ.get("GFT_FUZZ_SEEDS", "120"))))
def test_random_expression_trees(seed, interval, OTP, GTP, OTPI, GTPI):
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    gen = Gen(seed, interval)
    compared = 0
    for _ in range(4):
        try:
            o, g = gen.tree(O, G, 5 if seed % 3 == 0 else 4)
        except OracleRejected:
            continue
        _check(o, g)
        compared += 1
    assert compared >= 1 or seed > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 25))
def test_random_trees_mid_size_positive(seed, OTP, GTP):
    """Larger positive tensors (up to 28^3 / 72^2): products cross the tiled crossover (FMA, different summation order,
    inner split for rank 2), so the contract is 1e-10 relative per coefficient; shapes and degrees stay exact."""
    rng = np.random.default_rng(1000 + seed)
    nd = int(rng.integers(2, 4))
    side = 28 if nd == 3 else 72

    def leaf(k):
        shape = tuple(int(rng.integers(side // 2, side + 1)) for _ in range(nd))
        a = _rand(shape, 5000 + 10 * seed + k, 0.1, 1.0)
        deg = [side + 4] * nd
        return OTP.new(a, deg), GTP.new(a, deg)

    def tree(depth, k=0):
        if depth == 0:
            return leaf(k)
        op = int(rng.integers(0, 9))
        ao, ag = tree(depth - 1, 2 * k + 1)
        if op >= 6:
            # exp / log / div recurrences on a well-conditioned argument: c0 + t / (8 * max-norm bound)
            so, sg = OTP.from_scalar(1.0 / (8.0 * side ** nd)), GTP.from_scalar(1.0 / (8.0 * side ** nd))
            to, tg = ao * so, ag * sg
            if op == 6:
                return to.exp(), tg.exp()
            one_o, one_g = OTP.from_scalar(1.0), GTP.from_scalar(1.0)
            if op == 7:
                return (one_o + to).log(), (one_g + tg).log()
            return ao / (one_o + to), ag / (one_g + tg)
        if op <= 1:
            bo, bg = tree(depth - 1, 2 * k + 2)
            return (ao * bo, ag * bg) if op == 0 else (ao + bo, ag + bg)
        v = int(rng.integers(0, nd))
        if op == 2:
            return ao.derivative(v, 1), ag.derivative(v, 1)
        if op == 3:
            return ao.shift_down(v, 1), ag.shift_down(v, 1)
        if op == 4:
            d = int(rng.integers(side // 2, side))
            return ao.truncate_to_degree_p1(d), ag.truncate_to_degree_p1(d)
        lin = np.zeros([2 if ax == v else 1 for ax in range(nd)])
        lin.flat[0], lin.flat[1] = 0.25, 0.5
        deg = list(ao.degrees_p1())
        return ao.subst_var(v, OTP.new(lin, deg)), ag.subst_var(v, GTP.new(lin, deg))

    o, g = tree(2)
    assert o.shape() == g.shape() and o.degrees_p1() == g.degrees_p1()
    a, b = np.asarray(o.array()), np.asarray(g.array())
    assert np.all(np.abs(a - b) <= 1e-10 * np.abs(a)), np.max(np.abs(a - b) / np.abs(a))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(1, 1 + int(os.environ.get("GFT_FUZZ_SHAPES", "60"))))
def test_conv_tiled_random_shapes(seed):
    """The tiled kernel (forced, incl. the inner split for long last axes, compact operands, and every lane tile
    8x8 / 4x16 / 2x32 / 1x64 besides the planner's own choice) against the
    reference-order kernel on random ragged shapes of rank 2-4, mixed-sign data, slab range + accumulate:
    |err| <= 1e-10 * (|x| (*) |y|) coefficient-wise (SURVEY 8d normwise bound)."""
    import torch

    import genfer_amd

    rng = np.random.default_rng(9000 + seed)
    nd = int(rng.integers(2, 5))
    # shapes are drawn INSIDE the tiled kernel's domain (rank 3/4 directly, rank 2 and long last axes through the
    # inner split; unit axes are collapsed by the library first, so a rank-4 shape with one unit axis is a rank-3
    # problem); what lies outside is covered by test_conv_outside_tiled_domain_falls_back_bit_exactly
    hi = {2: 160, 3: 36, 4: 14}[nd]
    if os.environ.get("GFT_FUZZ_BIG"):  # occasional deep run: many tiles, stream-K ranges cut through tiles, reductions
        hi = {2: 700, 3: 100, 4: 30}[nd]
    zs = [int(rng.integers(2, hi + 1)) for _ in range(nd)]
    if nd == 2:
        zs[1] = int(rng.integers(2, 260))  # rank 2: directly (lane tile 1 x 64 rows) up to 128, through the split beyond
    if nd == 3 and seed % 5 == 0:
        zs = [int(rng.integers(2, 10)), int(rng.integers(2, 12)), int(rng.integers(129, 200))]  # rank-3 inner split
    if nd == 4 and seed % 4 == 0:
        zs[int(rng.integers(0, 3))] = 1  # collapses to rank 3
    # compact operands, but never a result larger than the product's support (xs + ys - 1): `Mul` cannot produce
    # that (sum_shape, mt:150-170), and the tiled planner refuses output tiles nothing contributes to
    xs = [int(rng.integers(1, z + 1)) for z in zs]
    ys = [int(rng.integers(max(1, z - a + 1), z + 1)) for z, a in zip(zs, xs)]
    x = _rand(xs, 100 + seed)
    y = _rand(ys, 200 + seed)
    L = genfer_amd.lib()
    tx, ty = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    ax, ay = tx.abs(), ty.abs()
    lo = int(rng.integers(0, zs[0]))
    hi_s = int(rng.integers(lo + 1, zs[0] + 1))
    z0 = torch.from_numpy(_rand(zs, 300 + seed)).cuda()

    tile = [0, 3, 4, 5, 6][seed % 5]

    def run(mode, a, b, acc):
        out = z0.clone() if acc else torch.full(zs, float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()  # torch fills on ITS stream; the library runs on its own non-blocking stream
        L.gft_set_conv_mode(mode)
        L.gft_set_option(b"tiled_tile", float(tile))
        try:
            genfer_amd.conv_raw(a.data_ptr(), xs, b.data_ptr(), ys, out.data_ptr(), zs, lo if acc else 0, hi_s if acc else zs[0], acc)
        finally:
            L.gft_set_conv_mode(0)
            L.gft_set_option(b"tiled_tile", 0.0)
        L.gft_synchronize()
        return out

    torch.cuda.synchronize()
    got = run(2, tx, ty, False)  # raises if the shape were outside the tiled domain: the draw above must not leave it
    want = run(1, tx, ty, False)
    bound = run(1, ax, ay, False)
    assert bool(torch.all((got - want).abs() <= 1e-10 * bound + 0.0))
    got = run(2, tx, ty, True)
    want = run(1, tx, ty, True)
    assert bool(torch.equal(got[:lo], z0[:lo])) and bool(torch.equal(got[hi_s:], z0[hi_s:]))
    assert bool(torch.all((got[lo:hi_s] - want[lo:hi_s]).abs() <= 1e-10 * (bound[lo:hi_s] + z0[lo:hi_s].abs())))


@pytest.mark.gpu
@pytest.mark.parametrize("xs,ys,zs", [
    ((300,), (250,), (400,)),                                    # rank 1
    ((3, 4, 5, 6, 7), (4, 4, 4, 4, 4), (5, 6, 7, 8, 9)),         # rank 5
    ((2, 3, 2, 3, 2, 3), (3, 2, 3, 2, 3, 2), (4, 4, 4, 4, 4, 4)),  # rank 6
])
def test_conv_outside_tiled_domain_falls_back_bit_exactly(xs, ys, zs):
    """Shapes the tiled kernel does not take (rank 1): forcing it is an error; small products of any rank: the automatic
    dispatch computes them on the reference-order kernels — bit-identical to the one-thread-per-output kernel (itself
    bit-exact against the oracle)."""
    import torch

    import genfer_amd

    L = genfer_amd.lib()
    x, y = _rand(xs, 71), _rand(ys, 72)
    tx, ty = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()

    def run(mode, a=None, b=None):
        a, b = (tx if a is None else a), (ty if b is None else b)
        out = torch.full(zs, float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        L.gft_set_conv_mode(mode)
        try:
            genfer_amd.conv_raw(a.data_ptr(), xs, b.data_ptr(), ys, out.data_ptr(), zs)
        finally:
            L.gft_set_conv_mode(0)
        L.gft_synchronize()
        return out

    if len(zs) < 2:
        with pytest.raises(genfer_amd.TaylorError, match="not supported"):
            run(2)
    else:  # rank >= 5: in the tiled domain since round 3 (host loop over the leading axes), within the tiled contract
        bound = run(1, tx.abs(), ty.abs())  # mixed-sign data: normwise against |x| (*) |y|
        assert bool(torch.all((run(2) - run(1)).abs() <= 1e-10 * bound))
    assert bool(torch.equal(run(0), run(1)))  # small products: the automatic dispatch stays on the reference-order kernels
