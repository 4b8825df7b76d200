"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar (north_star): integer bookkeeping (stored shape, degrees_p1, support/index
queries) bit-exact; values within 1e-10 relative — and bit-exact wherever the HIP path keeps the
reference's operation order (everything except libm seeds, wave-shuffle axis sums and the
tiled FMA convolution)."""
import itertools
import os

import numpy as np
import pytest

from conftest import splitmix64_uniform

pytestmark = pytest.mark.gpu

UMAX = 2**64 - 1


def both(OTP, GTP, arr, deg=None):
    arr = np.asarray(arr, dtype=np.float64)
    deg = list(arr.shape) if deg is None else deg
    return OTP.new(arr, deg), GTP.new(arr, deg)


def same_meta(o, g):
    assert g.degrees_p1() == o.degrees_p1()
    assert g.coeffs_shape() == o.coeffs_shape()


def check(o, g, exact=True, rel=1e-10, scale=None):
    same_meta(o, g)
    a, b = o.array(), g.array()
    if exact:
        assert np.array_equal(a, b, equal_nan=True), (a, b)
    else:
        ref = np.abs(a) if scale is None else scale
        assert np.all((np.abs(a - b) <= rel * ref) | (a == b)), (a, b)


def rand(shape, seed, lo=0.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * splitmix64_uniform(seed, n)).reshape(shape)


SHAPES = [
    ((5,), (7,), (8,)),
    ((1,), (6,), (6,)),
    ((3, 4), (4, 2), (5, 5)),
    ((4, 4), (4, 4), (4, 4)),
    ((2, 5), (5, 1), (6, 5)),
    ((1, 5), (5, 1), (6, 5)),
    ((3, 3, 3), (3, 3, 3), (4, 5, 3)),
    ((4, 1, 3), (2, 3, 3), (5, 3, 4)),
    ((6, 5, 7), (4, 5, 6), (8, 7, 9)),
    ((2, 3, 2, 3), (3, 2, 2, 2), (4, 4, 3, 4)),
    ((2, 2, 2, 2, 2), (2, 2, 2, 2, 2), (3, 2, 3, 2, 3)),
    ((9, 1, 1, 8), (1, 1, 1, 8), (12, 2, 2, 12)),
]


@pytest.mark.parametrize("xs,ys,deg", SHAPES)
def test_binary_ops_bit_exact(OTP, GTP, xs, ys, deg):
    """add/sub/mul/div keep the reference's operation order on device => bit-exact (f64 + - * / are
    IEEE-exact on gfx950; no FMA contraction: library built with -ffp-contract=off)."""
    x, y = rand(xs, 1, -1, 1), rand(ys, 2, 0.5, 1.5)
    ox, gx = both(OTP, GTP, x, deg)
    oy, gy = both(OTP, GTP, y, deg)
    check(ox + oy, gx + gy)
    check(ox - oy, gx - gy)
    check(oy - ox, gy - gx)
    check(ox * oy, gx * gy)
    check(oy * ox, gy * gx)
    check(ox / oy, gx / gy)
    check(-ox, -gx)


@pytest.mark.parametrize("xs,ys,deg", SHAPES[:9])
def test_exp_log_pow(OTP, GTP, xs, ys, deg):
    x = rand(xs, 3, 0.5, 1.5)
    ox, gx = both(OTP, GTP, x, deg)
    # the scalar seeds exp(x0) / ln(x0) are formed on the host (the platform libm, like the reference, f64.rs:54-61)
    # and everything after them keeps the reference's operation order: bit-exact
    check(ox.exp(), gx.exp())
    check(ox.log(), gx.log())
    for e in (0, 1, 2, 3, 5):
        check(ox.pow(e), gx.pow(e))


@pytest.mark.parametrize("xs,ys,deg", SHAPES)
def test_structural_ops_bit_exact(OTP, GTP, xs, ys, deg):
    x = rand(xs, 4, -1, 1)
    ox, gx = both(OTP, GTP, x, deg)
    nd = len(xs)
    for v in range(nd):
        for n in range(0, min(deg[v], 4)):
            check(ox.derivative(v, n), gx.derivative(v, n))
            check(ox.taylor_expansion_of_coeff(v, n), gx.taylor_expansion_of_coeff(v, n))
            check(ox.shift_down(v, n), gx.shift_down(v, n))
            check(ox.coefficients_of_term(v, n), gx.coefficients_of_term(v, n))
        check(ox.taylor_polynomial_terms(v, [0, 2]), gx.taylor_polynomial_terms(v, [0, 2]))
        check(ox.taylor_polynomial_terms(v, [1]), gx.taylor_polynomial_terms(v, [1]))
        check(ox.taylor_polynomial_terms(v, []), gx.taylor_polynomial_terms(v, []))
    check(ox.truncate_to_degree_p1(2), gx.truncate_to_degree_p1(2))
    check(ox.truncate_to_degree_p1(1), gx.truncate_to_degree_p1(1))
    check(ox.extend_to_dim(nd + 2, 5), gx.extend_to_dim(nd + 2, 5))
    check(ox.remove_last_variable(), gx.remove_last_variable())
    check(ox.extend(list(deg)), gx.extend(list(deg)))
    assert gx.clone() == gx
    assert gx.constant_term() == ox.constant_term()
    assert gx.extract_constant() == ox.extract_constant()
    assert gx.extract_linear() == ox.extract_linear()
    assert gx.is_constant() == ox.is_constant() and gx.num_vars() == ox.num_vars()
    for v in range(nd + 1):
        assert gx.len_of(v) == ox.len_of(v)
    for idx in itertools.islice(itertools.product(*[range(d) for d in deg]), 0, None, 7):
        assert gx.coefficient(idx) == ox.coefficient(idx)


@pytest.mark.parametrize("xs,ys,deg", SHAPES[:10])
def test_subst_var(OTP, GTP, xs, ys, deg):
    x, s = rand(xs, 5, -1, 1), rand(ys, 6, -0.5, 0.5)
    ox, gx = both(OTP, GTP, x, deg)
    os_, gs = both(OTP, GTP, s, deg)
    for v in range(len(xs)):
        check(ox.subst_var(v, os_), gx.subst_var(v, gs))  # general Horner: mul + add, bit-exact
        check(ox.subst_var(v, OTP.zero()), gx.subst_var(v, GTP.zero()))  # marginalize
        lin_o = OTP.from_scalar(0.3) * OTP.var_at_zero(v, deg[v])
        lin_g = GTP.from_scalar(0.3) * GTP.var_at_zero(v, deg[v])
        check(ox.subst_var(v, lin_o), gx.subst_var(v, lin_g))  # m*x_v scaling path
        sh_o, sh_g = OTP.var(v, 0.25, deg[v]), GTP.var(v, 0.25, deg[v])
        check(ox.subst_var(v, sh_o), gx.subst_var(v, sh_g))  # Taylor shift x0 + eps_v


def test_shortcut_dispatch_and_scalars(OTP, GTP):
    f = [[1.0, 2.0], [3.0, 4.0]]
    of, gf = both(OTP, GTP, f)
    for mk in ("zero", "one"):
        check(getattr(OTP, mk)() * of, getattr(GTP, mk)() * gf)
        check(of * getattr(OTP, mk)(), gf * getattr(GTP, mk)())
        check(of + getattr(OTP, mk)(), gf + getattr(GTP, mk)())
        check(getattr(OTP, mk)() - of, getattr(GTP, mk)() - gf)
    check(of / OTP.one(), gf / GTP.one())
    check(of / OTP.from_scalar(3.0), gf / GTP.from_scalar(3.0))
    check(OTP.from_scalar(2.0) + OTP.from_scalar(3.0), GTP.from_scalar(2.0) + GTP.from_scalar(3.0))
    check(OTP.from_scalar(2.0) * OTP.from_scalar(3.0), GTP.from_scalar(2.0) * GTP.from_scalar(3.0))
    check(OTP.from_u32(7).exp().log(), GTP.from_u32(7).exp().log(), exact=False)
    check(OTP.zero_with([3, 4]) * of, GTP.zero_with([3, 4]) * gf)
    # linear factor on either side (mt:1052-1061)
    ol = OTP.from_scalar(3.0) + OTP.from_scalar(2.0) * OTP.var_at_zero(1, 2)
    gl = GTP.from_scalar(3.0) + GTP.from_scalar(2.0) * GTP.var_at_zero(1, 2)
    assert gl.extract_linear() == ol.extract_linear() == (3.0, 2.0, 1)
    check(ol * of, gl * gf)
    check(of * ol, gf * gl)
    # untruncated degrees (usize::MAX, used by simplify: generating_function.rs:485,569)
    ou, gu = both(OTP, GTP, f, [UMAX, UMAX])
    check(ou * ou, gu * gu)
    check(ou + of, gu + gf)
    assert (gu * gu).degrees_p1() == (UMAX, UMAX)
    check(ou.subst_var(0, OTP.var(0, 0.5, 3)), gu.subst_var(0, GTP.var(0, 0.5, 3)))
    # 0-dim and all-ones shapes
    check(OTP.from_scalar(2.0).extend_to_dim(3, 4) * OTP.from_scalar(5.0), GTP.from_scalar(2.0).extend_to_dim(3, 4) * GTP.from_scalar(5.0))
    check(OTP.from_scalar(2.0).exp(), GTP.from_scalar(2.0).exp(), exact=False)


def test_error_behaviour_matches(OTP, GTP):
    from genfer_amd import TaylorError

    f = [[1.0, 2.0], [3.0, 4.0]]
    of, gf = both(OTP, GTP, f)
    for P in (of, gf):
        with pytest.raises(TaylorError):
            P.coefficient([2, 0])  # index out of bounds (mt:318-322)
        with pytest.raises(TaylorError):
            P.coefficient([1])  # index too short (mt:333-337)
        with pytest.raises(TaylorError):
            P.derivative(2, 0)  # bad variable (mt:459)
        with pytest.raises(TaylorError):
            P.shift_down(0, 2)  # n >= len_of(v) (mt:516)
    with pytest.raises(TaylorError):
        GTP.new([[1.0, 2.0]], [1, 1])  # shape exceeds degrees (mt:35-39)


def test_conv_raw_naive_bit_exact_vs_oracle(oracle_lib, GTP):
    """gft_conv_raw (reference-order kernel) on torch device buffers == oracle's orc_mul_raw bitwise,
    including slab ranges and accumulation."""
    import ctypes as C

    import torch

    import genfer_amd

    genfer_amd.lib().gft_set_conv_mode(1)
    try:
        sz = lambda s: (C.c_size_t * len(s))(*s)
        oracle_lib.orc_mul_raw.restype = C.c_int
        szp = C.POINTER(C.c_size_t)
        oracle_lib.orc_mul_raw.argtypes = [C.c_void_p, szp, C.c_void_p, szp, C.c_void_p, szp, C.c_size_t]
        for xs, ys, zs in [((9, 8, 7), (6, 8, 5), (12, 9, 10)), ((17,), (9,), (20,)), ((5, 6, 3, 4), (4, 3, 3, 2), (6, 6, 4, 4))]:
            x, y = rand(xs, 7, -1, 1), rand(ys, 8, -1, 1)
            want = np.zeros(zs)
            oracle_lib.orc_mul_raw(x.ctypes.data_as(C.c_void_p), sz(xs), y.ctypes.data_as(C.c_void_p), sz(ys),
                                   want.ctypes.data_as(C.c_void_p), sz(zs), len(zs))
            tx, ty = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
            tz = torch.full(zs, 7.0, dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            half = zs[0] // 2
            genfer_amd.conv_raw(tx.data_ptr(), xs, ty.data_ptr(), ys, tz.data_ptr(), zs, 0, half)
            genfer_amd.conv_raw(tx.data_ptr(), xs, ty.data_ptr(), ys, tz.data_ptr(), zs, half, zs[0])
            genfer_amd.lib().gft_synchronize()
            assert np.array_equal(tz.cpu().numpy(), want)
    finally:
        genfer_amd.lib().gft_set_conv_mode(0)


def test_interval_ops(OTPI, GTPI):
    """Interval<F64> planes: same op-for-op semantics as src/interval.rs (widen by one ulp, exact
    zero/one short-circuits) => bit-exact against the oracle."""
    for xs, ys, deg in SHAPES[:9]:
        lo = rand(xs, 11, -1, 1)
        x = np.stack([lo, lo + rand(xs, 12, 0, 1e-3)])
        lo = rand(ys, 13, 0.5, 1.5)
        y = np.stack([lo, lo + rand(ys, 14, 0, 1e-3)])
        x[(slice(None),) + tuple(0 for _ in xs)] = [0.0, 0.0] if len(xs) % 2 else [1.0, 1.0]  # exercise short-circuits
        ox, gx = OTPI.new(x, deg), GTPI.new(x, deg)
        oy, gy = OTPI.new(y, deg), GTPI.new(y, deg)
        check(ox + oy, gx + gy)
        check(ox - oy, gx - gy)
        check(ox * oy, gx * gy)
        check(ox / oy, gx / gy)
        check(-ox, -gx)
        for v in range(len(xs)):
            check(ox.derivative(v, 1), gx.derivative(v, 1))
            check(ox.shift_down(v, 1), gx.shift_down(v, 1))
            check(ox.subst_var(v, oy), gx.subst_var(v, gy))
        check(oy.exp(), gy.exp())
        check(oy.log(), gy.log())
        # soundness: lo <= hi everywhere
        r = (gx * gy).array()
        assert np.all(r[0] <= r[1])
    assert GTPI.from_scalar((2.0, 3.0)).constant_term() == (2.0, 3.0)
    assert GTPI.var(1, (0.5, 0.5), 4).extract_linear() == OTPI.var(1, (0.5, 0.5), 4).extract_linear()


def _conv_raw_gpu(mode, x, y, zs, slab=None, accumulate=False, z0=None):
    import torch

    import genfer_amd

    L = genfer_amd.lib()
    L.gft_set_conv_mode(mode)
    try:
        tx, ty = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        tz = torch.from_numpy(z0.copy()).cuda() if z0 is not None else torch.full(tuple(zs), np.nan, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        lo, hi = (0, zs[0]) if slab is None else slab
        genfer_amd.conv_raw(tx.data_ptr(), x.shape, ty.data_ptr(), y.shape, tz.data_ptr(), zs, lo, hi, accumulate)
        L.gft_synchronize()
        return tz.cpu().numpy()
    finally:
        L.gft_set_conv_mode(0)


@pytest.mark.parametrize("interval", [False, True])
def test_lazy_variables_and_scaled_variables(interval, OTP, GTP, OTPI, GTPI):
    """Variables built from host scalars stay on the host until a kernel must read them, and c * eps_v of a plain
    variable stays lazy too (IEEE identities c*0, c*1): every way of consuming them must match the oracle bit for
    bit — export, arithmetic, use as a substitution (the pure-scaling shortcut of mt:547-556) and as a product
    operand (mul_linear path)."""
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    mk = (lambda v: (v, v)) if interval else (lambda v: v)
    base = rand((5, 4, 6), 61, 0.1, 1.0)
    arr = np.stack([base, base + 1e-9]) if interval else base
    deg = [7, 6, 8]
    for v in range(3):
        for c in (2.5, -3.0, 0.5, 1.0, -0.0, 0.0):
            for x0 in (0.0, 0.75):
                ov = O.var_with_degrees_p1(v, mk(x0), deg)
                gv = G.var_with_degrees_p1(v, mk(x0), deg)
                check(ov, gv)
                os_, gs_ = ov * O.from_scalar(mk(c)), gv * G.from_scalar(mk(c))
                check(os_, gs_)
                assert os_.extract_linear() == gs_.extract_linear()
                check(O.from_scalar(mk(c)) * ov, G.from_scalar(mk(c)) * gv)
                op, gp = O.new(arr, deg), G.new(arr, deg)
                check(op.subst_var(v, os_), gp.subst_var(v, gs_))
                check(op * os_, gp * gs_)
                check(os_ + op, gs_ + gp)
                check(os_ - os_ * os_, gs_ - gs_ * gs_)
                check(os_.derivative(v, 1), gs_.derivative(v, 1))
                # affine substitutions built the way the interpreter builds them (Var +/- Const, Const + Var),
                # then "subst - constant_term(subst)" (generating_function.rs Subst evaluation)
                for d in (-1.0, 0.25, 0.0, -0.0):
                    for mkaff in (lambda t, k: t + k, lambda t, k: t - k, lambda t, k: k + t):
                        oa, ga = mkaff(os_, O.from_scalar(mk(d))), mkaff(gs_, G.from_scalar(mk(d)))
                        check(oa, ga)
                        assert oa.constant_term() == ga.constant_term()
                        oc, gc = oa - O.from_scalar(oa.constant_term()), ga - G.from_scalar(ga.constant_term())
                        check(oc, gc)
                        check(op.subst_var(v, oc), gp.subst_var(v, gc))
                        check(op.subst_var(v, oa), gp.subst_var(v, ga))


@pytest.mark.parametrize("loop_max", [1 << 40, 0])
@pytest.mark.parametrize("interval", [False, True])
def test_subst_var_linear_substitution_fused_horner(interval, loop_max, OTP, GTP, OTPI, GTPI):
    """subst_var with a linear substitution c + m*eps_w (c != 0 or w != v) runs the fused Horner step
    (k_horner_linear): bit-exact against the oracle's generic mul/add loop for every combination of substituted
    axis, substitution variable, c in {generic, 1, 0 with w != v}, ragged degrees and 1-d inputs."""
    import genfer_amd

    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    mk = (lambda a: np.stack([a, a + 1e-7])) if interval else (lambda a: a)
    cases = [((5, 4, 6), [7, 6, 8]), ((3, 7), [9, 7]), ((6,), [8]), ((2, 3, 2, 4), [4, 3, 3, 5]), ((4, 1, 5), [6, 4, 5])]
    # loop_max 2^40: all remaining steps in one launch (one workgroup per line); 0: one launch per step
    assert genfer_amd.lib().gft_set_option(b"horner_loop_max", float(loop_max)) == 0
    for shape, deg in cases:
        base = rand(shape, 71, -1.0, 1.0)
        op, gp = O.new(mk(base), deg), G.new(mk(base), deg)
        nd = len(shape)
        for v in range(nd):
            for w in range(nd):
                for c, m in ((0.3, 0.7), (1.0, -0.5), (-2.0, 1.0), (0.0, 0.25)):
                    if c == 0.0 and v == w:
                        continue  # pure scaling: different (table) path, covered elsewhere
                    lin = np.zeros([2 if ax == w else 1 for ax in range(nd)])
                    lin.flat[0], lin.flat[1] = c, m
                    sdeg = list(deg)
                    os_, gs_ = O.new(mk(lin), sdeg), G.new(mk(lin), sdeg)
                    check(op.subst_var(v, os_), gp.subst_var(v, gs_))
                    # the interpreter's Subst evaluation: read the constant term, subtract it, substitute — the
                    # host then knows element 0 of the device tensor (x - x = +0) and skips the device scan
                    oc0, gc0 = os_.constant_term(), gs_.constant_term()
                    assert oc0 == gc0
                    oz, gz = os_ - O.from_scalar(oc0), gs_ - G.from_scalar(gc0)
                    check(oz, gz)
                    check(op.subst_var(v, oz), gp.subst_var(v, gz))
                    check(op.subst_var(w, oz), gp.subst_var(w, gz))
    genfer_amd.lib().gft_set_option(b"horner_loop_max", float(1 << 40))


def test_subst_var_linear_long_axis_and_wide_tensor(OTP, GTP):
    """Lines longer than the loop kernel's 2048 fall back to one fused launch per Horner step; a wide tensor
    (many lines, substitution axis not last => strided lines) stays on the one-launch loop kernel."""
    base = rand((2100,), 81, -1.0, 1.0) * 0.999 ** np.arange(2100)
    op, gp = OTP.new(base, [2200]), GTP.new(base, [2200])
    lin = np.array([0.25, 0.5])
    check(op.subst_var(0, OTP.new(lin, [2200])), gp.subst_var(0, GTP.new(lin, [2200])))
    wide = rand((40, 30, 20), 82, -1.0, 1.0)
    deg = [45, 30, 20]
    ow, gw = OTP.new(wide, deg), GTP.new(wide, deg)
    for v, w in ((0, 0), (0, 1), (2, 0), (1, 2)):
        lin = np.zeros([2 if ax == w else 1 for ax in range(3)])
        lin.flat[0], lin.flat[1] = -0.5, 1.25
        check(ow.subst_var(v, OTP.new(lin, deg)), gw.subst_var(v, GTP.new(lin, deg)))


@pytest.mark.parametrize("interval", [False, True], ids=["f64", "interval"])
@pytest.mark.parametrize("shape,deg,sdeg", [((70,), [70], [70]), ((130,), [140], [140]), ((200,), [200], [150]), ((300,), [300], [300]),
                                            ((5, 130), [6, 130], [6, 130]), ((150, 3), [150, 4], [150, 4]), ((3, 200, 2), [3, 200, 2], [3, 190, 2])])
def test_subst_var_point_pipeline_multi_wave(shape, deg, sdeg, interval, OTP, GTP, OTPI, GTPI):
    """v -> c + m v on lines of 65 .. 1024 coefficients: the POINT wave pipeline with two to five waves per line.
    A wave joins the loop at the first step that reaches its positions (round 6) — the result must be the oracle's bit for bit,
    for positive data (the lean step), mixed signs (the generic step), a clipped degree and strided lines."""
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    nd = len(shape)
    v = max(range(nd), key=lambda ax: shape[ax])
    for positive in (True, False):
        base = rand(shape, 91, 0.05, 1.0) if positive else rand(shape, 92, -1.0, 1.0)
        base = base * (0.97 ** np.arange(shape[v])).reshape([-1 if ax == v else 1 for ax in range(nd)])
        mk = (lambda a: np.stack([a, a + np.abs(a) * 1e-12 + 1e-300])) if interval else (lambda a: a)
        op, gp = O.new(mk(base), deg), G.new(mk(base), deg)
        for c, m in ((0.3, 0.7), (-0.25, 0.9), (1e-13, 1.0)):
            lin = np.zeros([2 if ax == v else 1 for ax in range(nd)])
            lin.flat[0], lin.flat[1] = c, m
            os_, gs_ = O.new(mk(lin), sdeg), G.new(mk(lin), sdeg)  # (a smaller degree of the substitution clips the result)
            check(op.subst_var(v, os_), gp.subst_var(v, gs_))


STAGED_SHAPES = [
    ((17,), (9,), (20,)),                                  # rank 1: rows only, axis 0 is the staged axis
    ((300,), (300,), (300,)),                              # more than one chunk per row
    ((9, 8), (6, 8), (12, 9)),                             # rank 2: plane staging, axis 0 staged
    ((120, 130), (90, 130), (150, 130)),                   # rank 2, plane too large for 4 WGs/CU => rows
    ((9, 8, 7), (6, 8, 5), (12, 9, 10)),
    ((20, 17, 29), (13, 22, 30), (30, 30, 40)),            # y rows shorter than the output row (pitch != syb)
    ((3, 70, 70), (3, 70, 70), (3, 70, 70)),               # 76 KB of planes => rows
    ((5, 6, 3, 4), (4, 3, 3, 2), (6, 6, 4, 4)),
    ((3, 2, 4, 3, 5, 4), (2, 3, 2, 4, 3, 5), (4, 4, 5, 5, 6, 6)),   # rank 6: odometer over four outer axes
    ((1, 5, 1, 6), (3, 1, 4, 6), (3, 5, 4, 6)),            # unit axes (collapsed by the host)
]


@pytest.mark.parametrize("xs,ys,zs", STAGED_SHAPES)
def test_conv_staged_bit_identical_to_reference_order_kernel(xs, ys, zs, oracle_lib):
    """LDS-staged kernel (conv mode 3) == one-thread-per-output kernel (mode 1) == oracle, bit for bit:
    same per-output operation order, only the data movement differs.  Also slab ranges + accumulation."""
    import ctypes as C

    x, y = rand(xs, 31, -1, 1), rand(ys, 32, -1, 1)
    want = _conv_raw_gpu(1, x, y, zs)
    got = _conv_raw_gpu(3, x, y, zs)
    assert np.array_equal(got, want)
    sz = lambda s: (C.c_size_t * len(s))(*s)
    szp = C.POINTER(C.c_size_t)
    oracle_lib.orc_mul_raw.restype = C.c_int
    oracle_lib.orc_mul_raw.argtypes = [C.c_void_p, szp, C.c_void_p, szp, C.c_void_p, szp, C.c_size_t]
    ref = np.zeros(zs)
    oracle_lib.orc_mul_raw(x.ctypes.data_as(C.c_void_p), sz(xs), y.ctypes.data_as(C.c_void_p), sz(ys),
                           ref.ctypes.data_as(C.c_void_p), sz(zs), len(zs))
    assert np.array_equal(got, ref)
    lo, hi = zs[0] // 3, max(zs[0] // 3 + 1, (2 * zs[0]) // 3)
    z0 = rand(zs, 33)
    a = _conv_raw_gpu(1, x, y, zs, slab=(lo, hi), accumulate=True, z0=z0)
    b = _conv_raw_gpu(3, x, y, zs, slab=(lo, hi), accumulate=True, z0=z0)
    assert np.array_equal(a, b)
    assert np.array_equal(b[:lo], z0[:lo]) and np.array_equal(b[hi:], z0[hi:])


@pytest.mark.parametrize("mode", [1, 3])
def test_recurrences_same_bits_in_both_reference_order_kernels(mode, OTP, GTP, OTPI, GTPI):
    """div / exp / log / pow / subst_var drive the convolution kernels in slab mode (j0 bounds, exclusive and
    descending j0): forced one-thread-per-output (1) and forced LDS-staged (3) both reproduce the oracle."""
    import genfer_amd

    L = genfer_amd.lib()
    L.gft_set_conv_mode(mode)
    try:
        for shape in [(40,), (12, 70), (9, 8, 7), (4, 5, 3, 6)]:
            deg = list(shape)
            x, y = rand(shape, 41, 0.5, 1.5), rand(shape, 42, 0.5, 1.5)
            for O, G, mk in ((OTP, GTP, lambda a: a), (OTPI, GTPI, lambda a: np.stack([a, a + 1e-6]))):
                ox, gx, oy, gy = O.new(mk(x), deg), G.new(mk(x), deg), O.new(mk(y), deg), G.new(mk(y), deg)
                check(ox * oy, gx * gy)
                check(ox / oy, gx / gy)
                check(ox.pow(3), gx.pow(3))
                check(ox.subst_var(len(shape) - 1, oy), gx.subst_var(len(shape) - 1, gy))
                check(oy.exp(), gy.exp())
                check(oy.log(), gy.log())
    finally:
        L.gft_set_conv_mode(0)


def test_exp_recurrence_steps_on_the_tiled_kernel(OTP, GTP):
    """exp's slab steps above the tiled crossover run as plain slab products of shifted operand views on the tiled
    kernel (1e-10 contract; measured 1e-14); div and log keep the reference-order kernel and stay bit-exact."""
    shape = (24, 20, 28)
    a = rand(shape, 91, -0.2, 0.2)
    b = rand(shape, 92, -0.2, 0.2)
    b[0, 0, 0] = 1.5
    oa, ga, ob, gb = OTP.new(a, list(shape)), GTP.new(a, list(shape)), OTP.new(b, list(shape)), GTP.new(b, list(shape))
    # mixed-sign argument: normwise bound against the same recurrence on |a|, which dominates every
    # partial sum of the signed one
    want, got = oa.exp().array(), ga.exp().array()
    bound = OTP.new(np.abs(a), list(shape)).exp().array()
    assert np.all(np.abs(got - want) <= 1e-10 * bound)
    # positive argument: 1e-10 relative PER COEFFICIENT (north_star), values spanning many decades
    p = rand(shape, 93, 0.0, 0.2)
    want, got = OTP.new(p, list(shape)).exp().array(), GTP.new(p, list(shape)).exp().array()
    assert want.min() > 0 and want.max() / want.min() > 1e6
    assert np.all(np.abs(got - want) <= 1e-10 * want), np.max(np.abs(got - want) / want)
    check(oa / ob, ga / gb)
    check(ob.log(), gb.log())


TILED_SHAPES = [
    ((32, 32, 32), (32, 32, 32), (32, 32, 32)),
    ((20, 17, 29), (13, 22, 30), (30, 30, 40)),       # ragged, compact operands, inner not a multiple of 8
    ((9, 8, 7), (6, 8, 5), (12, 9, 10)),
    ((40, 3, 128), (2, 40, 128), (41, 40, 128)),      # maximum inner size, thin lane axes
    ((5, 6, 7, 20), (4, 3, 5, 17), (6, 6, 9, 24)),    # rank 4: uniform leading axis
    ((16, 16, 16, 16), (16, 16, 16, 16), (16, 16, 16, 16)),
    ((3, 70, 70, 8), (3, 70, 70, 8), (3, 70, 70, 8)),
]


@pytest.mark.parametrize("xs,ys,zs", TILED_SHAPES)
def test_conv_tiled_matches_reference_order_kernel(xs, ys, zs):
    """LDS-tiled FMA kernel vs the reference-order kernel (itself bit-exact vs the oracle, see
    test_conv_raw_naive_bit_exact_vs_oracle): positive inputs => 1e-10 relative per coefficient;
    mixed-sign inputs => normwise bound against |x| (*) |y|."""
    x, y = rand(xs, 21), rand(ys, 22)
    want = _conv_raw_gpu(1, x, y, zs)
    got = _conv_raw_gpu(2, x, y, zs)
    assert np.all(np.abs(got - want) <= 1e-10 * np.abs(want)), np.abs((got - want) / want).max()
    # mixed sign
    xm, ym = 2 * x - 1, 2 * y - 1
    bound = _conv_raw_gpu(1, np.abs(xm), np.abs(ym), zs)
    want = _conv_raw_gpu(1, xm, ym, zs)
    got = _conv_raw_gpu(2, xm, ym, zs)
    assert np.all(np.abs(got - want) <= 1e-10 * bound)
    # slab range + accumulate: only the selected leading slabs change, by exactly the product
    lo, hi = zs[0] // 3, max(zs[0] // 3 + 1, (2 * zs[0]) // 3)
    z0 = rand(zs, 23)
    got = _conv_raw_gpu(2, x, y, zs, slab=(lo, hi), accumulate=True, z0=z0)
    full = _conv_raw_gpu(1, x, y, zs)
    exp = z0.copy()
    exp[lo:hi] += full[lo:hi]
    assert np.array_equal(got[:lo], z0[:lo]) and np.array_equal(got[hi:], z0[hi:])
    assert np.all(np.abs(got[lo:hi] - exp[lo:hi]) <= 1e-10 * np.abs(exp[lo:hi]))
    # deterministic: two runs are bit-identical (fixed-order partial-slab reduction, no atomics)
    assert np.array_equal(_conv_raw_gpu(2, x, y, zs), _conv_raw_gpu(2, x, y, zs))


def test_conv_tiled_vs_oracle_and_pgf_like_dynamic_range(oracle_lib):
    """Tiled kernel against the CPU oracle directly, including a pgf-like operand spanning > 100
    orders of magnitude (relative accuracy per coefficient must survive: no FFT-style absolute error)."""
    import ctypes as C
    from math import lgamma, log

    szp = C.POINTER(C.c_size_t)
    oracle_lib.orc_mul_raw.restype = C.c_int
    oracle_lib.orc_mul_raw.argtypes = [C.c_void_p, szp, C.c_void_p, szp, C.c_void_p, szp, C.c_size_t]
    n = 40
    shape = (n, n, n)
    i = np.arange(n)
    pois = lambda lam: np.exp(i * log(lam) - lam - np.array([lgamma(k + 1) for k in i]))
    pgf = pois(20.0)[:, None, None] * pois(30.0)[None, :, None] * pois(40.0)[None, None, :]
    x = pgf * (1 + 0.1 * rand(shape, 31))
    y = pgf[::-1, ::-1, ::-1].copy() * (1 + 0.1 * rand(shape, 32))
    assert x.max() / x[x > 0].min() > 1e30
    want = np.zeros(shape)
    sz = (C.c_size_t * 3)(*shape)
    oracle_lib.orc_mul_raw(x.ctypes.data_as(C.c_void_p), sz, y.ctypes.data_as(C.c_void_p), sz,
                           want.ctypes.data_as(C.c_void_p), sz, 3)
    got = _conv_raw_gpu(2, x, y, shape)
    assert np.all(np.abs(got - want) <= 1e-10 * np.abs(want))


def test_conv_tiled_full_tensor_vs_oracle_64cubed(oracle_lib):
    """The tiled kernel against the CPU oracle DIRECTLY on a whole 64^3 tensor (9e9 multiply-adds, ~3.5 s of oracle):
    no transitivity through the GPU reference-order kernel.  Positive data: 1e-10 per coefficient; mixed signs:
    normwise against |x| (*) |y|."""
    import ctypes as C

    szp = C.POINTER(C.c_size_t)
    oracle_lib.orc_mul_raw.restype = C.c_int
    oracle_lib.orc_mul_raw.argtypes = [C.c_void_p, szp, C.c_void_p, szp, C.c_void_p, szp, C.c_size_t]
    shape = (64, 64, 64)
    sz = (C.c_size_t * 3)(*shape)

    def oracle(a, b):
        out = np.zeros(shape)
        oracle_lib.orc_mul_raw(a.ctypes.data_as(C.c_void_p), sz, b.ctypes.data_as(C.c_void_p), sz, out.ctypes.data_as(C.c_void_p), sz, 3)
        return out

    x, y = rand(shape, 61), rand(shape, 62)
    want = oracle(x, y)
    got = _conv_raw_gpu(2, x, y, shape)
    assert np.all(np.abs(got - want) <= 1e-10 * np.abs(want)), np.abs((got - want) / want).max()
    xm, ym = 2 * x - 1, 2 * y - 1
    bound = oracle(np.abs(xm), np.abs(ym))
    assert np.all(np.abs(_conv_raw_gpu(2, xm, ym, shape) - oracle(xm, ym)) <= 1e-10 * bound)


@pytest.mark.parametrize("shape", [(24, 24, 24, 24), (378, 378), (20, 30, 300)], ids=["rank4", "split_rank2", "split_rank3"])
def test_conv_tiled_rank4_and_inner_split_vs_oracle_directly(shape, oracle_lib):
    """Rank 4 (the wave-uniform leading axis) and the piece split of long last axes against the CPU oracle DIRECTLY on the
    whole tensor — 8e9 / 5e9 / 7e9 multiply-adds, 2-4 s of oracle each — not through the GPU reference-order kernel."""
    import ctypes as C

    szp = C.POINTER(C.c_size_t)
    oracle_lib.orc_mul_raw.restype = C.c_int
    oracle_lib.orc_mul_raw.argtypes = [C.c_void_p, szp, C.c_void_p, szp, C.c_void_p, szp, C.c_size_t]
    nd = len(shape)
    sz = (C.c_size_t * nd)(*shape)
    x, y = rand(shape, 71), rand(shape, 72)
    want = np.zeros(shape)
    oracle_lib.orc_mul_raw(x.ctypes.data_as(C.c_void_p), sz, y.ctypes.data_as(C.c_void_p), sz, want.ctypes.data_as(C.c_void_p), sz, nd)
    got = _conv_raw_gpu(2, x, y, shape)
    assert np.all(np.abs(got - want) <= 1e-10 * np.abs(want)), np.abs((got - want) / want).max()


IN_PLACE_SHAPES = [((32, 32, 32), (32, 32, 32), [32, 32, 32]), ((24, 40, 48), (24, 17, 48), [24, 40, 48]),
                   ((20, 12, 14, 16), (20, 12, 14, 16), [20, 12, 14, 16]), ((30, 33, 40), (30, 33, 24), [30, 33, 40])]


@pytest.mark.parametrize("xs,ys,deg", IN_PLACE_SHAPES, ids=["32cubed", "compact_y", "rank4", "compact_rows"])
def test_conv_tiled_in_place_operands_vs_oracle(xs, ys, deg, OTP, GTP, tier):
    """Operands whose own layout is the packed one (rows of whole 8-coefficient chunks, the library's buffers with their
    64 bytes of slack) are read in place: no packing launch, no zero padding — hence no non-finite verdict and no guarded
    fallback launch.  Through the handle API (raw caller pointers have no slack and are packed), forced onto the tiled
    kernel, against the oracle directly; with inf / NaN operands the result has the oracle's non-finite pattern and its
    finite coefficients agree to 1e-10 (nothing is polluted: there is no padding to multiply)."""
    import genfer_amd

    L = genfer_amd.lib()
    x, y = rand(xs, 91), rand(ys, 92)
    L.gft_set_conv_mode(2)
    try:
        gx, gy = GTP.new(x, deg), GTP.new(y, deg)
        (gx * gy).array()  # extract_linear verdicts are memoised on the operands' buffers now
        before = genfer_amd.op_stats()
        got = gx * gy
        after = genfer_amd.op_stats()
        want = OTP.new(x, deg) * OTP.new(y, deg)
        check(want, got, exact=False)
        if tier == "device":
            assert after["tiled"] - before["tiled"] == 1
            if xs[-1] == deg[-1] and ys[-1] == deg[-1] and os.environ.get("GFT_TILED_INPLACE") != "0":  # (rows that stop short of the result's are packed)
                assert after["launches"] - before["launches"] <= 2, "in-place operands: the main kernel and the reduce, nothing else"
        xi, yi = x.copy(), y.copy()
        xi[tuple(min(2, s - 1) for s in xs)] = np.inf
        yi[tuple(min(1, s - 1) for s in ys)] = np.nan
        for a, b in ((xi, y), (x, yi), (xi, yi)):
            g = np.asarray((GTP.new(a, deg) * GTP.new(b, deg)).array())
            w = np.asarray((OTP.new(a, deg) * OTP.new(b, deg)).array())
            assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(np.isinf(g), np.isinf(w))
            fin = np.isfinite(w)
            assert np.all(np.abs(g[fin] - w[fin]) <= 1e-10 * np.abs(w[fin]))
            assert np.array_equal(g[np.isinf(w)], w[np.isinf(w)])
    finally:
        L.gft_set_conv_mode(0)


SPLIT_SHAPES = [
    ((378, 378), (378, 378), (378, 378)),             # rank 2 (two_populations-like): last axis split into 6 x 64
    ((100, 130), (70, 97), (150, 200)),               # ragged rank 2, compact operands
    ((40, 96), (40, 96), (40, 96)),                   # rank 2, shortest last axis that is split
    ((12, 20, 300), (12, 20, 300), (12, 20, 300)),    # rank 3 with a last axis > 128 -> rank 4
    ((9, 7, 129), (5, 7, 140), (12, 9, 200)),
    ((3, 1000), (2, 1000), (4, 1000)),                # long rows, few of them
]


@pytest.mark.parametrize("xs,ys,zs", SPLIT_SHAPES)
def test_conv_tiled_inner_split_matches_reference_order_kernel(xs, ys, zs):
    """Rank-2 products and last axes longer than 128 reach the tiled kernel through the (P, B) inner split with
    overlap-add of the carries: same products as the reference => 1e-10 (positive inputs), normwise bound for
    mixed signs; slab range + accumulate only touch the selected rows."""
    x, y = rand(xs, 51), rand(ys, 52)
    want = _conv_raw_gpu(1, x, y, zs)
    got = _conv_raw_gpu(2, x, y, zs)
    assert np.all(np.abs(got - want) <= 1e-10 * np.abs(want)), np.abs((got - want) / want).max()
    xm, ym = 2 * x - 1, 2 * y - 1
    bound = _conv_raw_gpu(1, np.abs(xm), np.abs(ym), zs)
    assert np.all(np.abs(_conv_raw_gpu(2, xm, ym, zs) - _conv_raw_gpu(1, xm, ym, zs)) <= 1e-10 * bound)
    lo, hi = zs[0] // 3, max(zs[0] // 3 + 1, (2 * zs[0]) // 3)
    z0 = rand(zs, 53)
    got = _conv_raw_gpu(2, x, y, zs, slab=(lo, hi), accumulate=True, z0=z0)
    exp = z0.copy()
    exp[lo:hi] += want[lo:hi]
    assert np.array_equal(got[:lo], z0[:lo]) and np.array_equal(got[hi:], z0[hi:])
    assert np.all(np.abs(got[lo:hi] - exp[lo:hi]) <= 1e-10 * np.abs(exp[lo:hi]))
    # non-finite operands: the device-side guard routes the whole product to the reference-order kernel
    xi = x.copy()
    xi[(1,) * len(xs)] = np.inf
    assert np.array_equal(_conv_raw_gpu(2, xi, y, zs), _conv_raw_gpu(1, xi, y, zs), equal_nan=True)


@pytest.mark.parametrize("shape", [(16, 16, 20), (40, 40, 40), (6, 20, 20, 24)])
def test_conv_tiled_nonfinite_operands_fall_back(shape):
    """inf/NaN operands must not be polluted by zero padding.  The verdict is taken on the device (epoch stamp
    written by the packing/scan kernels): the tiled kernels leave z alone and the guarded reference-order launch
    computes it, so auto mode AND forced-tiled mode equal the reference-order kernel bit for bit; a following
    finite product on the same stream is unaffected (the stamp is per product)."""
    x, y = rand(shape, 41), rand(shape, 42)
    xf, yf = x.copy(), y.copy()
    x[(3, 4, 5, 1)[: len(shape)]] = np.inf
    y[(1, 2, 3, 0)[: len(shape)]] = np.nan
    want = _conv_raw_gpu(1, x, y, shape)
    for mode in (0, 2):
        got = _conv_raw_gpu(mode, x, y, shape)
        assert np.array_equal(got, want, equal_nan=True)
        only_x = _conv_raw_gpu(mode, x, yf, shape)
        assert np.array_equal(only_x, _conv_raw_gpu(1, x, yf, shape), equal_nan=True)
        fin = _conv_raw_gpu(mode, xf, yf, shape)
        ref = _conv_raw_gpu(1, xf, yf, shape)
        assert np.all(np.isfinite(fin)) and np.all(np.abs(fin - ref) <= 1e-10 * np.abs(ref))


def test_full_size_c2_properties():
    """BASELINE configs[1] at full size (128^3): tiled result vs the reference-order kernel on the
    whole tensor, plus size-independent properties: commutativity (bitwise on the tiled kernel is
    not required; 1e-10), linearity in x, and the constant-term / corner known answers."""
    shape = (128, 128, 128)
    x, y = rand(shape, 1), rand(shape, 2)
    got = _conv_raw_gpu(2, x, y, shape)
    want = _conv_raw_gpu(1, x, y, shape)
    assert np.all(np.abs(got - want) <= 1e-10 * np.abs(want)), np.abs((got - want) / want).max()
    comm = _conv_raw_gpu(2, y, x, shape)
    assert np.all(np.abs(got - comm) <= 1e-10 * np.abs(want))
    got2 = _conv_raw_gpu(2, 2.0 * x, y, shape)  # scaling by a power of two is exact
    assert np.array_equal(got2, 2.0 * got)
    assert got[0, 0, 0] == x[0, 0, 0] * y[0, 0, 0]
    assert abs(got[0, 0, 1] - (x[0, 0, 0] * y[0, 0, 1] + x[0, 0, 1] * y[0, 0, 0])) <= 1e-15


def test_full_size_c4_properties():
    """BASELINE configs[3] at full size (64^4, the 8-GPU workload) on one GPU: size-independent properties of the
    tiled result (commutativity, exact power-of-two scaling, corner known answers, slab additivity) and the first
    leading slab against the reference-order kernel."""
    shape = (64, 64, 64, 64)
    x, y = rand(shape, 3), rand(shape, 4)
    got = _conv_raw_gpu(2, x, y, shape)
    comm = _conv_raw_gpu(2, y, x, shape)
    assert np.all(np.abs(got - comm) <= 1e-10 * np.abs(got))
    assert np.array_equal(_conv_raw_gpu(2, 0.5 * x, y, shape), 0.5 * got)
    assert got[0, 0, 0, 0] == x[0, 0, 0, 0] * y[0, 0, 0, 0]
    assert abs(got[0, 0, 0, 1] - (x[0, 0, 0, 0] * y[0, 0, 0, 1] + x[0, 0, 0, 1] * y[0, 0, 0, 0])) <= 1e-15
    # output sharding: two disjoint slab ranges written into one buffer reproduce the full product
    z = np.full(shape, np.nan)
    a = _conv_raw_gpu(2, x, y, shape, slab=(0, 40), z0=z)
    b = _conv_raw_gpu(2, x, y, shape, slab=(40, 64), z0=a)
    assert np.all(np.abs(b - got) <= 1e-10 * np.abs(got))
    # leading slab vs the reference-order kernel (9e9 MACs)
    want0 = _conv_raw_gpu(3, x, y, shape, slab=(0, 1), z0=np.zeros(shape))[0]
    assert np.all(np.abs(got[0] - want0) <= 1e-10 * np.abs(want0))


def _oracle_slabs_parallel(oracle_lib, x, y, zs, slabs, threads):
    """Leading-axis output slabs of the reference-order product on the oracle, one slab per host thread (slabs are
    independent, mt:1001-1011; the per-element operation order is the reference's).  Returns {k0: slab}."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    szp = C.POINTER(C.c_size_t)
    oracle_lib.orc_mul_slabs_timed.restype = C.c_double
    oracle_lib.orc_mul_slabs_timed.argtypes = [C.c_void_p, szp, C.c_void_p, szp, C.c_void_p, szp, C.c_size_t, C.c_size_t,
                                               C.c_size_t, C.POINTER(C.c_double)]
    nd = len(zs)
    res = np.zeros(zs)
    sx, sy, sz = (C.c_size_t * nd)(*x.shape), (C.c_size_t * nd)(*y.shape), (C.c_size_t * nd)(*zs)

    def one(k):
        m = C.c_double(0.0)
        oracle_lib.orc_mul_slabs_timed(x.ctypes.data_as(C.c_void_p), sx, y.ctypes.data_as(C.c_void_p), sy,
                                       res.ctypes.data_as(C.c_void_p), sz, nd, k, k + 1, C.byref(m))
        return k

    order = sorted(slabs, reverse=True)  # heaviest first
    with ThreadPoolExecutor(max(1, threads)) as ex:
        list(ex.map(one, order))
    return {k: res[k] for k in slabs}


def test_full_size_c2_whole_tensor_vs_oracle_directly(oracle_lib):
    """BASELINE configs[1] (128^3), the headline tensor itself: EVERY coefficient of the tiled product against the CPU
    oracle (not through the GPU reference-order kernel).  5.6e11 multiply-adds on the oracle: all host threads, one
    leading slab each (about 16 s on the 256-core GPU box); skipped below 32 threads."""
    threads = os.cpu_count() or 1
    if threads < 32:
        pytest.skip(f"the full 128^3 oracle product needs >= 32 host threads to finish in a test's time (have {threads})")
    shape = (128, 128, 128)
    x, y = rand(shape, 1), rand(shape, 2)
    got = _conv_raw_gpu(2, x, y, shape)
    want = _oracle_slabs_parallel(oracle_lib, x, y, shape, range(128), min(threads, 128))
    worst = 0.0
    for k in range(128):
        err = np.abs(got[k] - want[k]) / np.abs(want[k])
        worst = max(worst, float(err.max()))
    assert worst <= 1e-10, worst


def test_c4_slabs_vs_oracle_directly(oracle_lib):
    """BASELINE configs[3] (64^4): the leading slabs {0, 31, 63} of the tiled result against the oracle directly (not the
    GPU reference-order kernel).  Slab 63 alone is 5.8e11 multiply-adds, so the slabs are cut into their 64 rows along
    the second axis (orc_mul_rows: the same terms in the same order per output) and spread over the host's threads;
    with fewer than 32 threads only light slabs are checked."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    threads = os.cpu_count() or 1
    shape = (64, 64, 64, 64)
    x, y = rand(shape, 3), rand(shape, 4)
    slabs = [0, 31, 63] if threads >= 32 else ([0, 7] if threads >= 8 else [0, 1])
    szp = C.POINTER(C.c_size_t)
    oracle_lib.orc_mul_rows.restype = C.c_int
    oracle_lib.orc_mul_rows.argtypes = [C.c_void_p, szp, C.c_void_p, szp, C.c_void_p, szp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    sz = (C.c_size_t * 4)(*shape)
    want = np.zeros(shape)

    def one(task):
        k0, k1 = task
        return oracle_lib.orc_mul_rows(x.ctypes.data_as(C.c_void_p), sz, y.ctypes.data_as(C.c_void_p), sz,
                                       want.ctypes.data_as(C.c_void_p), sz, 4, k0, k1, k1 + 1)

    tasks = sorted(((k0, k1) for k0 in slabs for k1 in range(64)), key=lambda t: -(t[0] + 1) * (t[1] + 1))
    with ThreadPoolExecutor(min(threads, 128)) as ex:
        assert all(rc == 0 for rc in ex.map(one, tasks))
    for k in slabs:
        got = _conv_raw_gpu(2, x, y, shape, slab=(k, k + 1), z0=np.zeros(shape))[k]
        assert np.all(np.abs(got - want[k]) <= 1e-10 * np.abs(want[k])), (k, np.abs((got - want[k]) / want[k]).max())


def test_conv_tiled_rank5_vs_oracle_directly(oracle_lib):
    """A rank-5 general product on the tiled kernel (host loop over the leading axis, accumulate-mode rank-4 launches)
    against the oracle's orc_mul_raw directly: 1e-10 on positive data, normwise on mixed-sign data."""
    import ctypes as C

    import genfer_amd

    szp = C.POINTER(C.c_size_t)
    oracle_lib.orc_mul_raw.restype = C.c_int
    oracle_lib.orc_mul_raw.argtypes = [C.c_void_p, szp, C.c_void_p, szp, C.c_void_p, szp, C.c_size_t]
    sz = lambda s: (C.c_size_t * len(s))(*s)

    def oracle(x, y, zs):
        ref = np.zeros(zs)
        assert oracle_lib.orc_mul_raw(x.ctypes.data_as(C.c_void_p), sz(x.shape), y.ctypes.data_as(C.c_void_p), sz(y.shape),
                                      ref.ctypes.data_as(C.c_void_p), sz(zs), len(zs)) == 0
        return ref

    for xs, ys, zs in (((4, 5, 6, 7, 8), (4, 5, 6, 7, 8), (4, 5, 6, 7, 8)), ((3, 6, 7, 8, 9), (2, 5, 7, 6, 9), (4, 8, 9, 10, 12))):
        x, y = rand(xs, 171), rand(ys, 172)
        before = genfer_amd.op_stats()["tiled"]
        got = _conv_raw_gpu(2, x, y, zs)
        assert genfer_amd.op_stats()["tiled"] > before, "the product did not take the tiled kernel"
        want = oracle(x, y, zs)
        assert np.all(np.abs(got - want) <= 1e-10 * np.abs(want)), np.abs((got - want) / np.where(want == 0, 1, want)).max()
        xm, ym = 2 * x - 1, 2 * y - 1
        bound = oracle(np.abs(xm), np.abs(ym), zs)
        assert np.all(np.abs(_conv_raw_gpu(2, xm, ym, zs) - oracle(xm, ym, zs)) <= 1e-10 * bound)


@pytest.mark.parametrize("interval", [False, True])
def test_derivative_truncated_equals_two_step_form(interval, OTP, GTP, OTPI, GTPI):
    """gft_derivative_truncated == derivative(v, n).truncate_to_degree_p1(d) of the oracle (generating_function.rs
    Derivative arm), bit for bit, including the zero / assertion-free edge cases."""
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    mk = (lambda a: np.stack([a, a + 1e-8])) if interval else (lambda a: a)
    for shape, deg in (((6, 5, 7), [9, 8, 9]), ((4, 9), [6, 12]), ((5,), [7]), ((3, 1, 4), [5, 3, 6])):
        base = rand(shape, 97, -1.0, 1.0)
        o, g = O.new(mk(base), deg), G.new(mk(base), deg)
        for v in range(len(shape)):
            for n in (0, 1, 2, 4):
                if n >= deg[v]:
                    continue
                for d in (1, 3, 5, 20):
                    check(o.derivative(v, n).truncate_to_degree_p1(d), g.derivative_truncated(v, n, d))
                    check(o.derivative_truncated(v, n, d), g.derivative_truncated(v, n, d))


def test_full_size_c5_interval_product_encloses_f64(GTP, GTPI):
    """BASELINE configs[4] at the C2 size: point intervals [x, x] of the 128^3 data through the interval product
    (LDS-staged reference-order kernel).  Soundness at full size: every f64 coefficient of the (tiled) f64 product
    lies inside its interval, the intervals are thin (<= a few thousand ulps for 2e6-term sums), lo <= hi."""
    shape = (128, 128, 128)
    x, y = rand(shape, 1), rand(shape, 2)
    iv = (GTPI.new(np.stack([x, x]), list(shape)) * GTPI.new(np.stack([y, y]), list(shape))).array()
    lo, hi = np.asarray(iv[0]), np.asarray(iv[1])
    mid = (GTP.new(x, list(shape)) * GTP.new(y, list(shape))).array()
    assert np.all(lo <= hi)
    slack = 1e-10 * np.abs(mid)  # the f64 product itself is only 1e-10-close to the exactly rounded sum
    assert np.all(lo - slack <= mid) and np.all(mid <= hi + slack)
    assert np.all(hi - lo <= 1e-9 * np.abs(mid))
    assert lo[0, 0, 0] == hi[0, 0, 0] == x[0, 0, 0] * y[0, 0, 0] or lo[0, 0, 0] < x[0, 0, 0] * y[0, 0, 0] < hi[0, 0, 0]


def test_full_size_c5_mixed_sign_interval_product_encloses_f64(GTP, GTPI):
    """BASELINE configs[4] at the C2 size with MIXED-SIGN data (no sign-regime fast path applies to all of it): thin
    intervals [x, x + 1e-12 |x|] around data in [-1, 1).  Soundness at full size: lo <= hi, and the f64 product of the
    lower endpoints lies inside every interval up to the f64 product's own normwise error (1e-10 |x| (*) |y|)."""
    shape = (128, 128, 128)
    x, y = 2 * rand(shape, 1) - 1, 2 * rand(shape, 2) - 1
    xh, yh = x + 1e-12 * np.abs(x), y + 1e-12 * np.abs(y)
    iv = (GTPI.new(np.stack([x, xh]), list(shape)) * GTPI.new(np.stack([y, yh]), list(shape))).array()
    lo, hi = np.asarray(iv[0]), np.asarray(iv[1])
    assert np.all(lo <= hi)
    mid = (GTP.new(x, list(shape)) * GTP.new(y, list(shape))).array()
    bound = (GTP.new(np.abs(x), list(shape)) * GTP.new(np.abs(y), list(shape))).array()
    slack = 1e-10 * bound
    assert np.all(lo - slack <= mid) and np.all(mid <= hi + slack)
    assert np.all(hi - lo <= 1e-8 * bound)  # widths: 1e-12 relative inputs + ~2e6 outward roundings per coefficient
    assert lo[0, 0, 0] <= x[0, 0, 0] * y[0, 0, 0] <= hi[0, 0, 0]


def _oracle_interval_slabs(oracle_lib, xa, ya, shape, slabs, threads):
    """Leading slabs of the interval product of xa * ya (arrays [2, *shape]: lo plane, hi plane) by the ORACLE, spread over host
    threads row by row (orci_mul_rows: array-of-intervals layout).  Returns {k0: array [2, *shape[1:]]}."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    nd = len(shape)
    szp = C.POINTER(C.c_size_t)
    oracle_lib.orci_mul_rows.restype = C.c_int
    oracle_lib.orci_mul_rows.argtypes = [C.c_void_p, szp, C.c_void_p, szp, C.c_void_p, szp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    sz = (C.c_size_t * nd)(*shape)
    xi = np.ascontiguousarray(np.stack([xa[0], xa[1]], axis=-1))  # element i = {lo, hi}
    yi = np.ascontiguousarray(np.stack([ya[0], ya[1]], axis=-1))
    want = np.zeros(tuple(shape) + (2,))

    def one(task):
        k0, k1 = task
        return oracle_lib.orci_mul_rows(xi.ctypes.data_as(C.c_void_p), sz, yi.ctypes.data_as(C.c_void_p), sz, want.ctypes.data_as(C.c_void_p), sz,
                                        nd, k0, k1, k1 + 1)

    tasks = sorted(((k0, k1) for k0 in slabs for k1 in range(shape[1])), key=lambda t: -(t[0] + 1) * (t[1] + 1))
    with ThreadPoolExecutor(min(threads, 192)) as ex:
        assert all(rc == 0 for rc in ex.map(one, tasks))
    return {k: np.stack([want[k][..., 0], want[k][..., 1]]) for k in slabs}


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("data", ["positive", "mixed"])
def test_full_size_c5_interval_slabs_vs_oracle_directly(oracle_lib, GTPI, data):
    """BASELINE configs[4] at its headline size, against the ORACLE's bits (round-5 verdict, Weak #1: the 128^3 interval product
    was only checked for enclosure and against the GPU's own staged kernel).  Leading slabs {0, 63, 127} of the product the
    library computes in auto mode, bit for bit against orci_mul_rows on all host threads — positive data (the positive
    regime) and mixed-sign data with a zero, a one and an infinity in it (finite / general regimes).  With fewer than 32
    host threads only light slabs are checked (slab 127 alone is 8.7e9 interval multiply-adds)."""
    threads = os.cpu_count() or 1
    shape = (128, 128, 128)
    if data == "positive":
        x, y = rand(shape, 1), rand(shape, 2)
        xa, ya = np.stack([x, x * (1 + 1e-15)]), np.stack([y, y * (1 + 1e-15)])
    else:
        x, y = 2 * rand(shape, 1) - 1, 2 * rand(shape, 2) - 1
        x[1, 1, 1] = 0.0
        y[2, 2, 2] = 1.0
        xa, ya = np.stack([x, x + 1e-15 * np.abs(x)]), np.stack([y, y + 1e-15 * np.abs(y)])
        xa[1][3, 3, 3] = np.inf
    slabs = [0, 63, 127] if threads >= 32 else ([0, 5] if threads >= 8 else [0, 1])
    want = _oracle_interval_slabs(oracle_lib, xa, ya, shape, slabs, threads)
    got = np.asarray((GTPI.new(xa, list(shape)) * GTPI.new(ya, list(shape))).array())
    for k in slabs:
        same = _bits_equal(got[:, k], want[k])
        assert np.all(same), (data, k, int((~same).sum()), got[:, k][~same][:4], want[k][~same][:4])


def test_full_size_interval_row_pair_88_whole_tensor_vs_oracle(oracle_lib, GTPI):
    """The row-pair form at 88^3 (slab ranges under the 2 GiB workspace), EVERY coefficient against the oracle directly,
    positive data; needs the host's threads (6.1e10 interval multiply-adds), so below 32 threads a 40^3 product is checked."""
    threads = os.cpu_count() or 1
    n = 88 if threads >= 32 else 40
    shape = (n, n, n)
    x, y = rand(shape, 301, 0.1, 1.0), rand(shape, 302, 0.1, 1.0)
    xa, ya = np.stack([x, x * (1 + 1e-15)]), np.stack([y, y * (1 + 1e-15)])
    want = _oracle_interval_slabs(oracle_lib, xa, ya, shape, list(range(n)), threads)
    got = np.asarray((GTPI.new(xa, list(shape)) * GTPI.new(ya, list(shape))).array())
    for k in range(n):
        same = _bits_equal(got[:, k], want[k])
        assert np.all(same), (k, int((~same).sum()))


@pytest.mark.parametrize("shape", [(72, 72, 72), (88, 88, 88), (20, 20, 20, 24), (32, 32, 32, 32)])
def test_full_size_interval_row_pair_ranges_equal_the_staged_kernel(GTPI, shape):
    """The row-pair form under its default 2 GiB workspace at sizes that take every kind of slab range — 72^3 (ranges of 8
    slabs of axis 0 on two lanes), 88^3 (ranges of 4), 20^3 x 24 (ranges of slabs of the leading axis), 32^4 (ranges of the
    leading axis below, ranges of axis 0 inside one leading slab above: one such slab alone is 4.6 GB) — against the
    LDS-staged reference-order kernel (row-pair form off), which the oracle pins at the sizes it finishes: bit for bit,
    positive data (positive regime in both passes) and mixed-sign data with an exact zero, a one and an infinity in it."""
    import genfer_amd

    L = genfer_amd.lib()
    xp, yp = rand(shape, 301, 0.1, 1.0), rand(shape, 302, 0.1, 1.0)
    xm, ym = rand(shape, 303, -1.0, 1.0), rand(shape, 304, -1.0, 1.0)
    xm[(1,) * len(shape)] = 0.0
    ym[(2,) * len(shape)] = 1.0
    cases = [(np.stack([xp, xp * (1 + 1e-15)]), np.stack([yp, yp * (1 + 1e-15)])),
             (np.stack([xm, xm + 1e-15 * np.abs(xm)]), np.stack([ym, ym + 1e-15 * np.abs(ym)]))]
    cases[1][0][1][(3,) * len(shape)] = np.inf
    L.gft_set_option(b"host_max_elems", 0.0)
    try:
        for a, b in cases:
            got = {}
            for pairs in (2.0, 0.0):
                assert L.gft_set_option(b"conv_rb_pairs", pairs) == 0
                try:
                    got[pairs] = np.asarray((GTPI.new(a, list(shape)) * GTPI.new(b, list(shape))).array())
                finally:
                    L.gft_set_option(b"conv_rb_pairs", -1.0)
            u, v = got[2.0].view(np.uint64), got[0.0].view(np.uint64)
            same = (u == v) | (np.isnan(got[2.0]) & np.isnan(got[0.0]))
            assert np.all(same), (shape, int((~same).sum()), got[2.0][~same][:4], got[0.0][~same][:4])
            if os.environ.get("GFT_RB_PAIRS_CAP_MB") is None:  # (the verification matrix also runs this with other caps)
                assert genfer_amd.pool_stats()["in_use"] < (3 << 30)  # (operands, results and a workspace of at most 2 GiB)
    finally:
        L.gft_set_option(b"host_max_elems", -1.0)


def test_interval_edge_values_bit_exact(OTPI, GTPI):
    """Interval add / sub / mul / div on the edge values of f64 (signed zeros, smallest subnormals, largest finite,
    infinities, NaN, exact 0 / 1 / -1 points): the outward widening (f64.rs:127-171 next_up / next_down) and every
    short-circuit of interval.rs must agree with the oracle bit for bit, NaN payloads aside."""
    tiny, big, inf, nan = 5e-324, 1.7976931348623157e308, np.inf, np.nan
    vals = [0.0, -0.0, tiny, -tiny, 2.2250738585072014e-308, 1.0, -1.0, 0.5, -3.0, big, -big, inf, -inf, nan]
    pts = [(a, a) for a in vals] + [(-1.0, 1.0), (0.0, 1.0), (-tiny, tiny), (1.0, inf), (-inf, -1.0), (-0.0, 0.0), (0.25, 0.75)]
    lo = np.array([p[0] for p in pts])
    hi = np.array([p[1] for p in pts])
    n = len(pts)
    # all ordered pairs as two 1-d tensors (element-wise ops act per coefficient; products via the [n] x [1] trick)
    A = np.stack([np.repeat(lo, n), np.repeat(hi, n)])
    B = np.stack([np.tile(lo, n), np.tile(hi, n)])
    oa, ga, ob, gb = OTPI.new(A, [n * n]), GTPI.new(A, [n * n]), OTPI.new(B, [n * n]), GTPI.new(B, [n * n])

    def same(o, g):
        a, b = np.asarray(o.array()), np.asarray(g.array())
        assert a.shape == b.shape
        ok = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
        assert np.all(ok), (a[~ok][:4], b[~ok][:4])

    same(oa + ob, ga + gb)
    same(oa - ob, ga - gb)
    same(-oa, -ga)
    for i, (l, h) in enumerate(pts):  # scalar * tensor, tensor / scalar, tensor * scalar: every pair through mul / div
        so, sg = OTPI.from_scalar((l, h)), GTPI.from_scalar((l, h))
        same(so * oa, sg * ga)
        same(oa * so, ga * sg)
        same(oa / so, ga / sg)
    # general products (convolution MAC path incl. the wave-uniform fast path and its fallback)
    x = np.stack([lo.reshape(3, 7), hi.reshape(3, 7)])
    y = np.stack([lo[::-1].reshape(3, 7), hi[::-1].reshape(3, 7)])
    same(OTPI.new(x, [3, 7]) * OTPI.new(y, [3, 7]), GTPI.new(x, [3, 7]) * GTPI.new(y, [3, 7]))


def test_many_variables_few_nontrivial_axes(OTP, GTP):
    """20 variables of which three carry coefficients: unit axes are collapsed on the host, so the kernels see
    rank 3 (the reference's 8-variable programs have this shape pattern)."""
    nd = 20
    shape = [1] * nd
    shape[2], shape[9], shape[17] = 5, 6, 4
    deg = [3] * nd
    deg[2], deg[9], deg[17] = 7, 8, 6
    x, y = rand(shape, 95, -1, 1), rand(shape, 96, -1, 1)
    ox, gx, oy, gy = OTP.new(x, deg), GTP.new(x, deg), OTP.new(y, deg), GTP.new(y, deg)
    check(ox * oy, gx * gy)
    check(ox + oy, gx + gy)
    check(ox.derivative(9, 2), gx.derivative(9, 2))
    check(ox.shift_down(17, 1), gx.shift_down(17, 1))
    check(ox.subst_var(2, oy), gx.subst_var(2, gy))
    check((ox * oy) / (oy + OTP.from_scalar(3.0)), (gx * gy) / (gy + GTP.from_scalar(3.0)))


@pytest.mark.parametrize("xs,ys,deg", SHAPES)
def test_observe_step_fused_equals_reference_sequence(OTP, GTP, xs, ys, deg):
    """gft_observe_step (one kernel) == derivative -> truncate -> * var -> * const (three reference calls),
    bit for bit, on both backends; includes the exact-zero / exact-one scalar shortcuts."""
    x = rand(xs, 51, -1, 1)
    ox, gx = both(OTP, GTP, x, deg)
    for v in range(len(xs)):
        for d in (1, 2, 3, max(deg)):
            for xv in (0.0, 1.0, 0.37):
                for c in (1.0, 0.25, 0.0):
                    if not (1 < deg[v]):
                        continue
                    want = (ox.derivative(v, 1).truncate_to_degree_p1(d) * OTP.var(v, xv, d)) * OTP.from_scalar(c)
                    check(want, ox.observe_step(v, xv, c, d))
                    check(want, gx.observe_step(v, xv, c, d))
                    unfused = (gx.derivative(v, 1).truncate_to_degree_p1(d) * GTP.var(v, xv, d)) * GTP.from_scalar(c)
                    check(want, unfused)


def test_observe_step_interval(OTPI, GTPI):
    for xs, ys, deg in SHAPES[:8]:
        lo = rand(xs, 52, -1, 1)
        x = np.stack([lo, lo + rand(xs, 53, 0, 1e-3)])
        # the short-circuit operands of iv:126-190 among the coefficients: exact zeros of both signs, the points 1 and -1,
        # a non-finite bound (the kernel resolves `* [1,1]` and `0 + A` where they stand and takes the wave-checked general
        # formulas elsewhere: every one of these must still come out as the reference's if-chain has it)
        xs_ = x.copy()
        flat = xs_.reshape(2, -1)
        n = flat.shape[1]
        for i, (l, h) in enumerate(((0.0, 0.0), (-0.0, -0.0), (1.0, 1.0), (-1.0, -1.0), (-0.0, 0.0), (0.5, np.inf), (np.nan, 1.0))):
            if i < n:
                flat[0, (3 * i + 1) % n] = l
                flat[1, (3 * i + 1) % n] = h
        for data in (x, xs_):
            ox, gx = OTPI.new(data, deg), GTPI.new(data, deg)
            for v in range(len(xs)):
                if not (1 < deg[v]):
                    continue
                for xv, c in (((0.0, 0.0), (0.5, 0.5)), ((1.0, 1.0), (1.0, 1.0)), ((0.3, 0.31), (0.2, 0.21)), ((-1.0, -1.0), (-0.5, 0.25))):
                    want = (ox.derivative(v, 1).truncate_to_degree_p1(3) * OTPI.var(v, xv, 3)) * OTPI.from_scalar(c)
                    check(want, gx.observe_step(v, xv, c, 3))


@pytest.mark.parametrize("interval", [False, True])
def test_derive_scale_fused_equals_reference_sequence(interval, OTP, GTP, OTPI, GTPI):
    """gft_derive_scale = derivative(v, 1).truncate_to_degree_p1(d) * from(c) in one gather (the continuous-rate Poisson
    observation step, generating_function.rs:703-706): bit-identical to the oracle's unfused sequence, including Mul's
    zero / one shortcuts, non-finite constants and the 1-element cases."""
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    mk = (lambda a: np.stack([a, a + np.abs(a) * 1e-12])) if interval else (lambda a: a)
    sc = (lambda c: (c, c + abs(c) * 1e-12)) if interval else (lambda c: c)
    for shape, deg in [((7,), [9]), ((5, 6), [6, 8]), ((3, 4, 5), [4, 4, 7]), ((2, 1, 6), [3, 2, 6]), ((1, 5), [4, 5]), ((40, 50), [40, 50])]:
        a = rand(shape, 77, -1, 1)
        o, g = O.new(mk(a), deg), G.new(mk(a), deg)
        for v in range(len(shape)):
            for c in (0.3, -2.5, 1.0, 0.0, float("inf"), float("nan")):
                for d in (1, 2, 3, max(deg) + 1):
                    if not 1 < o.len_of(v):
                        continue
                    check(o.derive_scale(v, sc(c), d), g.derive_scale(v, sc(c), d))
    # inside a whole program: continuous-rate Poisson observations
    src = "rate ~ Exponential(1);\nobserve 3 ~ Poisson(2 * rate);\nobserve 1 ~ Poisson(rate);\nreturn rate;\n"
    import genfer_amd
    import os
    from conftest import ROOT

    flags = "--no-timing" + (" --bounds" if interval else "")
    rc, want, _ = genfer_amd.run_sgcl_with_backend(src, flags, os.path.join(ROOT, "oracle", "liborc.so"), "orci_" if interval else "orc_")
    assert rc == 0, want
    got, _ = genfer_amd.run_sgcl(src, flags)
    assert got == want


@pytest.mark.parametrize("interval", [False, True])
def test_observe_chain_fused_equals_stepwise_reference(interval, OTP, GTP, OTPI, GTPI):
    """gft_observe_chain = n observation steps (derivative -> truncate -> * (x + eps_v) -> * c_k, generating_function.rs:
    684-689), each one degree lower than the one inside it, in ONE launch: bit-identical to the oracle's unfused loop —
    ordinary chains, chains longer than one launch takes (16 steps), x = 0 / 1, constants 0 / 1 / non-finite, 1-element
    intermediates (all of which make the library fall back to single steps)."""
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    mk = (lambda a: np.stack([a, a + np.abs(a) * 1e-12])) if interval else (lambda a: a)
    sc = (lambda c: (c, c + abs(c) * 1e-12)) if interval else (lambda c: c)
    rng = np.random.default_rng(11)
    cases = [((30,), [40], 0, 7), ((12, 25), [14, 30], 1, 9), ((12, 25), [14, 30], 0, 5), ((5, 6, 20), [6, 6, 24], 2, 12),
             ((3, 70), [4, 90], 1, 55), ((90, 40), [100, 48], 0, 20), ((2, 9), [2, 9], 1, 6)]
    for shape, deg, v, n in cases:
        a = rand(shape, 5 + n, 0.1, 1.0)
        o, g = O.new(mk(a), deg), G.new(mk(a), deg)
        # (the same chain on data with exact zeros of both signs, ones, minus ones and a non-finite coefficient)
        sp = a.copy().reshape(-1)
        for i, val in enumerate((0.0, -0.0, 1.0, -1.0, np.inf)):
            sp[(5 * i + 2) % sp.size] = val
        osp, gsp = O.new(mk(sp.reshape(shape)), deg), G.new(mk(sp.reshape(shape)), deg)
        cs_sp = [sc(float(c)) for c in rng.uniform(0.2, 1.5, size=n)]
        check(osp.observe_chain(v, sc(0.7), cs_sp, max(deg)), gsp.observe_chain(v, sc(0.7), cs_sp, max(deg)))
        for x in (0.7, 0.0, 1.0):
            cs = [sc(float(c)) for c in rng.uniform(0.2, 1.5, size=n)]
            for d in (1, 3, max(deg)):
                check(o.observe_chain(v, sc(x), cs, d), g.observe_chain(v, sc(x), cs, d))
        cs = [sc(0.5)] * (n - 1) + [sc(1.0)]
        check(o.observe_chain(v, sc(0.3), cs, 4), g.observe_chain(v, sc(0.3), cs, 4))
        for bad in (0.0, float("inf"), float("nan")):
            cs = [sc(0.5), sc(bad), sc(0.25)]
            check(o.observe_chain(v, sc(0.3), cs, 5), g.observe_chain(v, sc(0.3), cs, 5))


@pytest.mark.parametrize("interval", [False, True])
def test_add_scaled_fused_equals_reference_sequence(interval, OTP, GTP, OTPI, GTPI):
    """gft_add_scaled(a, b, c) = a + b * from(c) in one pass (the Lah-number accumulation of the negative-binomial
    observation, generating_function.rs:743-746): bit-identical to the two reference calls, ragged shapes, different
    degrees, scalar operands and c in {0, 1, -1, inf, nan} included."""
    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    mk = (lambda a: np.stack([a, a + np.abs(a) * 1e-12])) if interval else (lambda a: a)
    sc = (lambda c: (c, c + abs(c) * 1e-12)) if interval else (lambda c: c)
    cases = [((5,), [7], (3,), [6]), ((3, 4), [5, 5], (4, 2), [4, 6]), ((2, 3, 4), [4, 4, 4], (3, 1, 2), [3, 4, 5]),
             ((60, 70), [64, 80], (64, 50), [64, 64]), ((1,), [4], (3,), [4]), ((3,), [4], (1,), [4])]
    for sa, da, sb, db in cases:
        a, b = rand(sa, 31, -1, 1), rand(sb, 32, -1, 1)
        oa, ga, ob, gb = O.new(mk(a), da), G.new(mk(a), da), O.new(mk(b), db), G.new(mk(b), db)
        for c in (0.375, -2.0, 1.0, 0.0, -1.0, float("inf"), float("nan")):
            check(oa.add_scaled(ob, sc(c)), ga.add_scaled(gb, sc(c)))
    # the whole negative-binomial observation path
    src = "X ~ Poisson(4.5);\nobserve 7 ~ NegBinomial(X, 0.25);\nreturn X;\n"
    import os
    import genfer_amd
    from conftest import ROOT

    flags = "--no-timing --limit 12" + (" --bounds" if interval else "")
    rc, want, _ = genfer_amd.run_sgcl_with_backend(src, flags, os.path.join(ROOT, "oracle", "liborc.so"), "orci_" if interval else "orc_")
    assert rc == 0, want
    got, _ = genfer_amd.run_sgcl(src, flags)
    assert got == want


@pytest.mark.parametrize("shape", [(24, 40), (6, 10, 33), (3, 4, 5, 20)])
def test_interval_positive_regime_leaves_and_returns_bit_exact(OTPI, GTPI, shape):
    """Strictly positive interval products whose sums LEAVE the positive regime on the way: a term whose lower bound
    underflows to zero, an upper bound that overflows, a lower bound that reaches the least subnormal.  The staged
    kernel's positive-regime sums carry no per-term test (a bad term poisons the running bound with a NaN pattern and the
    finished sum is tested once); whatever it decides, the bits must be the reference's."""
    n = int(np.prod(shape))
    base = 0.05 + rand(shape, 1301)
    for tag, scale_x, scale_y in (("underflow", 1e-170, 1e-170), ("overflow", 1e160, 1e155), ("subnormal", 1e-160, 1e-160), ("mixed", 1.0, 1.0)):
        x, y = base * scale_x, (0.05 + rand(shape, 1302)) * scale_y
        if tag == "mixed":  # a few extreme entries inside ordinary data: only some sums (some lanes) fall back
            x, y = x.copy(), y.copy()
            x.flat[n // 3], y.flat[n // 2], x.flat[n - 2], y.flat[1] = 1e-300, 1e-300, 1e200, 1e200
        xi, yi = np.stack([x, x * (1 + 1e-15)]), np.stack([y, y * (1 + 1e-15)])
        o = OTPI.new(xi, list(shape)) * OTPI.new(yi, list(shape))
        g = GTPI.new(xi, list(shape)) * GTPI.new(yi, list(shape))
        same_meta(o, g)
        a, b = np.asarray(o.array()), np.asarray(g.array())
        ok = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
        assert np.all(ok), (tag, a[~ok][:4], b[~ok][:4])


@pytest.mark.parametrize("shape", [(20, 18, 16), (5, 6, 9, 10)])
def test_recurrences_side_stream_overlap_is_bit_identical(GTP, GTPI, shape):
    """div / log with the bulk of the right-looking update on the side stream (default) against the single-stream
    order ("recur_overlap" = 0): same launches, same operands, only the stream differs — identical bits, f64 and
    interval, including back-to-back calls that reuse pooled buffers."""
    import genfer_amd

    L = genfer_amd.lib()
    x, y = 0.5 + rand(shape, 1401), 0.5 + rand(shape, 1402)

    def run(cls, xa, ya):
        a, b = cls.new(xa, list(shape)), cls.new(ya, list(shape))
        out = []
        for _ in range(2):
            q = a / b
            l = (cls.from_scalar(1.0 if xa.ndim == len(shape) else (1.0, 1.0)) + a).log()
            out += [np.asarray(q.array()).copy(), np.asarray(l.array()).copy()]
        return out

    for cls, xa, ya in ((GTP, x, y), (GTPI, np.stack([x, x * (1 + 1e-15)]), np.stack([y, y * (1 + 1e-15)]))):
        L.gft_set_option(b"recur_overlap", 1.0)
        on = run(cls, xa, ya)
        L.gft_set_option(b"recur_overlap", 0.0)
        try:
            off = run(cls, xa, ya)
        finally:
            L.gft_set_option(b"recur_overlap", 1.0)
        for p, q in zip(on, off):
            assert np.array_equal(p.view(np.uint64), q.view(np.uint64))


def test_log_fused_slab_step_equals_unfused(OTP, GTP):
    """log of 3-d tensors: the slab step fused into the 2-d division launch (neg, += k * xs[k], divide by xs[0], / k,
    rs[k] = res[k] * k) against the oracle, bit for bit — compact arguments (xs shorter than the result on every
    axis) and full ones."""
    for xs, deg in (((12, 10, 14), [12, 10, 14]), ((5, 7, 3), [9, 8, 11]), ((9, 2, 64), [9, 6, 64])):
        x = 0.2 * rand(xs, 1501, -1, 1)
        x[(0,) * len(xs)] = 1.25
        o, g = both(OTP, GTP, x, deg)
        check(o.log(), g.log())


@pytest.mark.parametrize("xs,ys", [((40,), (40,)), ((130,), (9,)), ((200,), (200,)), ((7, 90), (7, 90)), ((12, 33), (5, 20)), ((3, 4, 70), (3, 4, 70))])
def test_division_with_non_finite_and_extreme_coefficients_bit_exact(OTP, GTP, OTPI, GTPI, xs, ys):
    """The blocked division kernels run their rows WITHOUT per-lane bounds (excluded positions multiply a staged zero)
    and with the 3-instruction quotient inside an exponent window; a non-finite or out-of-window coefficient must send
    the row through the bounded / full-division form.  Dividends and divisors with inf, NaN, huge, tiny and exact-zero
    coefficients: quotients bit-identical to the oracle (NaN payloads aside), f64 and interval."""
    deg = [max(a, b) for a, b in zip(xs, ys)]
    base_x, base_y = rand(xs, 1601, -1, 1), rand(ys, 1602, -1, 1)
    base_y[(0,) * len(ys)] = 1.5
    for tag, edit in (("inf_in_x", lambda x, y: x.__setitem__(tuple(min(2, s - 1) for s in xs), np.inf)),
                      ("nan_in_x", lambda x, y: x.__setitem__(tuple(min(1, s - 1) for s in xs), np.nan)),
                      ("huge", lambda x, y: x.__setitem__(tuple(min(3, s - 1) for s in xs), 1e306)),
                      ("tiny", lambda x, y: x.__setitem__(tuple(0 for _ in xs), 1e-310)),
                      ("inf_in_y", lambda x, y: y.__setitem__(tuple(min(1, s - 1) for s in ys), np.inf)),
                      ("zeros", lambda x, y: (x.__setitem__(tuple(0 for _ in xs), 0.0), y.flat.__setitem__(slice(1, None, 3), 0.0)))):
        x, y = base_x.copy(), base_y.copy()
        edit(x, y)
        for O, G, mk in ((OTP, GTP, lambda a: a), (OTPI, GTPI, lambda a: np.stack([a, a]))):
            o = O.new(mk(x), deg) / O.new(mk(y), deg)
            g = G.new(mk(x), deg) / G.new(mk(y), deg)
            same_meta(o, g)
            a, b = np.asarray(o.array()), np.asarray(g.array())
            ok = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
            assert np.all(ok), (tag, O.__qualname__, np.argwhere(~ok)[:3], a[~ok][:3], b[~ok][:3])


def test_row_oriented_gather_large_tensors_bit_exact(OTP, GTP, OTPI, GTPI):
    """Tensors of >= 2^20 elements whose gathers miss the 16-byte path (a shift along the last axis, odd row lengths,
    interval planes) take the row-oriented kernel (one wave per row): derivative / coefficient expansion / mul_var /
    truncation / linear substitution along every axis against the oracle, bit for bit."""
    shape = (96, 101, 111)  # 1.08e6 elements, odd rows
    x = rand(shape, 1701, -1, 1)
    o, g = both(OTP, GTP, x, list(shape))
    for v in range(3):
        check(o.derivative(v, 2), g.derivative(v, 2))
        check(o.taylor_expansion_of_coeff(v, 1), g.taylor_expansion_of_coeff(v, 1))
        check(o * OTP.var(v, 0.0, shape[v]), g * GTP.var(v, 0.0, shape[v]))
        lin = np.zeros([2 if ax == v else 1 for ax in range(3)])
        lin.flat[1] = 0.75
        check(o.subst_var(v, OTP.new(lin, list(shape))), g.subst_var(v, GTP.new(lin, list(shape))))
    check(o.truncate_to_degree_p1(57), g.truncate_to_degree_p1(57))
    xi = np.stack([x, x + 1e-9])
    oi, gi = OTPI.new(xi, list(shape)), GTPI.new(xi, list(shape))
    for v in (0, 2):
        a, b = oi.derivative(v, 1), gi.derivative(v, 1)
        same_meta(a, b)
        assert np.array_equal(np.asarray(a.array()), np.asarray(b.array()))


@pytest.mark.parametrize("interval", [False, True])
def test_deferred_chains_bit_exact_and_fewer_launches(interval, OTP, GTP, OTPI, GTPI, tier):
    """Round 3: runs of elementwise operations on a device tensor — subst_var by a pure scaling m*x_v (mt:557-565),
    `+ const` (mt:862-869), `* const` / `/ const` / neg (mt:1041-1047), truncation — are recorded on the handle and
    evaluated inside the consuming Add/Sub (the two arms of an `if`, generating_function.rs:557-566).  Bit for bit the
    oracle's sequence, whichever consumer ends the chain; and the chains really replace launches."""
    import genfer_amd

    O, G = (OTPI, GTPI) if interval else (OTP, GTP)
    mk = (lambda a: np.stack([a, a + 1e-9 * np.abs(a)])) if interval else (lambda a: a)
    sc = (lambda x: (x, x)) if interval else (lambda x: x)
    L = genfer_amd.lib()

    def program(T, A, B, v, d, deg, Z):
        s = T.var(v, sc(0.0), deg[v]) * T.from_scalar(sc(0.9048374180359595))  # m * x_v
        p = (A.subst_var(v, s) + T.from_scalar(sc(0.25))) * T.from_scalar(sc(0.5))
        q = (B.subst_var(v, s).truncate_to_degree_p1(d) - T.from_scalar(sc(1.5))) / T.from_scalar(sc(3.0))
        n = T.from_scalar(sc(2.0)) - (-q)                      # scalar - tensor: FIRST_SUB_NEG_ALL after a neg stage
        outs = [p + q, p - q, n + p, (p + q).truncate_to_degree_p1(max(d - 1, 1))]
        outs.append(p * q)                                     # a consumer that needs the tensors in memory
        # Mul asks `self.extract_linear()` first (mt:1014-1072): on a chain that is one launch — materialise and scan
        # (k_chain_scan) — whatever the verdict; Z * 3 IS linear (two non-zero coefficients in a full-size tensor)
        outs.append((Z * T.from_scalar(sc(3.0))) * q)
        outs.append(((Z + T.from_scalar(sc(0.5))) * T.from_scalar(sc(-0.25))) * (q + p))
        outs.append((p + T.from_scalar(sc(1.0))).derivative(v, 1))
        seven = p
        for k in range(8):                                     # longer than CHAIN_MAX: the chain restarts
            seven = seven * T.from_scalar(sc(1.0 + 0.125 * k))
        outs.append(seven + q)
        # sub-box views that do not start at element 0 (slab extraction): deferred on position-independent chains only
        outs.append((A * T.from_scalar(sc(2.0))).coefficients_of_term(v, 1) + q.coefficients_of_term(v, 0))
        outs.append(((-B).coefficients_of_term(v, min(2, deg[v] - 1)) * T.from_scalar(sc(0.75))) - p.coefficients_of_term(v, 1))
        # mul by a linear polynomial c + m x_v (mul_linear, mt:611-623): mul_var's shifted operand is a front-padded view
        lin = T.var(v, sc(0.8), deg[v]) * T.from_scalar(sc(0.2)) + T.from_scalar(sc(0.64))
        outs.append(A * lin)
        outs.append((B * T.from_scalar(sc(-1.5))) * lin)          # a chain under the pad
        outs.append(A.mul_var(sc(-0.5), v, [min(d_ + (1 if i_ == v else 0), g_) for i_, (d_, g_) in enumerate(zip(A.coeffs_shape(), deg))], deg) * T.from_scalar(sc(-2.0)))
        return outs, [p.constant_term(), q.constant_term(), n.constant_term(), seven.constant_term()]

    for shape, deg in (((24, 20), [30, 22]), ((6, 7, 9), [8, 8, 9]), ((40,), [64]), ((3, 1, 12), [5, 4, 12])):
        a, b = rand(shape, 201, 0.1, 1.0), rand(shape, 202, -1.0, 1.0)
        for v in range(len(shape)):
            for d in (3, 7, 100):
                zl = np.zeros(shape)
                zl[(0,) * len(shape)] = 0.7
                if shape[v] >= 2:
                    zl[tuple(1 if i_ == v else 0 for i_ in range(len(shape)))] = 0.3
                want, wc = program(O, O.new(mk(a), deg), O.new(mk(b), deg), v, d, deg, O.new(mk(zl), deg))
                counts = {}
                for defer in (1, 0):
                    assert L.gft_set_option(b"defer", float(defer)) == 0
                    try:
                        before = genfer_amd.op_stats()
                        got, gc = program(G, G.new(mk(a), deg), G.new(mk(b), deg), v, d, deg, G.new(mk(zl), deg))
                        for o, g in zip(want, got):
                            check(o, g)
                        assert gc == wc
                        after = genfer_amd.op_stats()
                        counts[defer] = {k: after[k] - before[k] for k in ("launches", "deferred_ops", "chain_addsub_launches")}
                    finally:
                        L.gft_set_option(b"defer", 1.0)
                assert counts[0]["deferred_ops"] == 0
                if tier == "device" and int(np.prod(shape)) >= 64:  # the tensors live in HBM
                    assert counts[1]["deferred_ops"] > 0 and counts[1]["launches"] < counts[0]["launches"], counts


HIGH_RANK_SHAPES = [
    ((3, 6, 7, 8, 9), (2, 5, 7, 6, 9), (4, 8, 9, 10, 12)),             # rank 5, ragged, compact operands
    ((4, 4, 8, 8, 8), (4, 4, 8, 8, 8), (4, 4, 8, 8, 8)),               # rank 5, truncated at the operand shapes
    ((2, 3, 4, 5, 6, 7), (3, 2, 4, 4, 5, 7), (4, 4, 6, 7, 8, 10)),     # rank 6
    ((3, 1, 5, 6, 1, 7, 8), (2, 1, 4, 6, 1, 7, 5), (4, 1, 7, 9, 1, 10, 10)),  # rank 7 with unit axes: rank 5 after collapsing
]


@pytest.mark.parametrize("xs,ys,zs", HIGH_RANK_SHAPES)
def test_conv_tiled_rank5_and_up_matches_reference_order_kernel(xs, ys, zs):
    """Rank >= 5 general products (the reference's mul is rank-generic, mt:984-1012) on the tiled FMA kernel: the leading
    axes beyond the kernel's four are walked on the host with accumulate-mode rank-4 launches.  Against the
    reference-order kernel (bit-exact vs the oracle): 1e-10 per coefficient on positive data, normwise on mixed signs;
    a leading-slab range writes only its slabs; non-finite operands fall back to the reference's result."""
    import genfer_amd

    x, y = rand(xs, 71), rand(ys, 72)
    want = _conv_raw_gpu(1, x, y, zs)
    before = genfer_amd.op_stats()["tiled"]
    got = _conv_raw_gpu(2, x, y, zs)
    assert genfer_amd.op_stats()["tiled"] > before, "the product did not take the tiled kernel"
    assert np.all(np.abs(got - want) <= 1e-10 * np.abs(want)), np.abs((got - want) / want).max()
    xm, ym = 2 * x - 1, 2 * y - 1
    bound = _conv_raw_gpu(1, np.abs(xm), np.abs(ym), zs)
    assert np.all(np.abs(_conv_raw_gpu(2, xm, ym, zs) - _conv_raw_gpu(1, xm, ym, zs)) <= 1e-10 * bound)
    lo, hi = zs[0] // 3, max(zs[0] // 3 + 1, (2 * zs[0]) // 3)
    z0 = rand(zs, 73)
    part = _conv_raw_gpu(2, x, y, zs, slab=(lo, hi), z0=z0)
    assert np.array_equal(part[:lo], z0[:lo]) and np.array_equal(part[hi:], z0[hi:])
    assert np.all(np.abs(part[lo:hi] - want[lo:hi]) <= 1e-10 * np.abs(want[lo:hi]))
    xi = x.copy()
    xi[(1,) * len(xs) if all(s > 1 for s in xs) else tuple(min(1, s - 1) for s in xs)] = np.inf
    assert np.array_equal(_conv_raw_gpu(2, xi, y, zs), _conv_raw_gpu(1, xi, y, zs), equal_nan=True)


WAVEFRONT_SHAPES = [
    # (quotient shape, divisor shape, dividend shape)
    ((40, 40, 40), (40, 40, 40), (40, 40, 40)),
    ((30, 20, 33), (7, 20, 12), (30, 5, 33)),        # compact divisor and dividend
    ((12, 10, 9, 16), (12, 10, 9, 16), (12, 10, 9, 16)),  # rank 4
    ((200, 48), (200, 48), (200, 48)),               # rank 2, many rows
    ((70, 64), (3, 64), (70, 1)),                    # rows of exactly 64, thin operands
    ((21, 19, 31), (5, 19, 31), (21, 7, 9)),         # rows <= 32: two source rows per wave, odd row counts
    ((9, 11, 32), (9, 11, 32), (9, 11, 32)),         # rows of exactly 32
    ((13, 5, 7, 24), (13, 2, 7, 24), (4, 5, 7, 24)),  # rank 4, packed rows
    # thousands of source rows per quotient row: four / eight source rows per wave, four coefficients per lane (round 4)
    ((48, 48, 40), (48, 48, 40), (48, 48, 40)),      # rows of 33 .. 64: 16-lane groups
    ((17, 13, 11, 24), (17, 5, 11, 24), (9, 13, 11, 20)),  # rank 4, rows <= 32: 8-lane groups, compact operands
    ((2100, 9), (2100, 9), (2100, 9)),               # rank 2, short rows, a long chain
    # rank 2 with rows > 64: the coefficient-level wavefront (tasks are 64-coefficient segments of rows; round 4)
    ((24, 130), (24, 130), (24, 130)),               # three segments, the last one partial
    ((60, 200), (7, 70), (45, 150)),                 # compact divisor (its rows end inside segment 1) and dividend
    ((150, 65), (150, 65), (150, 2)),                # one coefficient into the second segment; a thin dividend
    ((33, 256), (33, 100), (33, 256)),               # whole segments; divisor rows shorter than the quotient's
    ((130, 130), (130, 130), (130, 130)),
    # rank 3 / 4 with rows > 64: the segment wavefront with leading axes (round 6; before: the slab-by-slab form, a 30x cliff at row 65)
    ((6, 7, 70), (6, 7, 70), (6, 7, 70)),            # one coefficient run into the second segment (interval data too)
    ((5, 9, 130), (2, 9, 70), (5, 4, 100)),          # compact divisor and dividend, three segments
    ((4, 3, 5, 66), (4, 3, 5, 66), (4, 3, 5, 66)),   # rank 4
    ((66, 66, 66), (66, 66, 66), (66, 66, 66)),      # just over the old cliff
    ((10, 70, 130), (10, 70, 130), (10, 70, 130)),
    ((6, 5, 100, 70), (6, 5, 100, 70), (6, 5, 100, 70)),
    ((20, 96, 96), (20, 96, 96), (20, 96, 96)),
]


@pytest.mark.parametrize("zs,ys,xs", WAVEFRONT_SHAPES)
def test_div_row_wavefront_bit_exact(zs, ys, xs, OTP, GTP, OTPI, GTPI, tier):
    """The division as a row wavefront (one launch, every quotient row a task of one wave, dependencies through per-row
    flags) consumes its terms in the reference's order (mt:1162-1192 over mt:984-1012): bit-exact against the oracle,
    and identical to the slab-by-slab blocked recurrence (`div_wavefront` = 0); finite, non-finite and interval data."""
    import genfer_amd

    L = genfer_amd.lib()
    _div_row_wavefront_bit_exact(zs, ys, xs, OTP, GTP, OTPI, GTPI, tier, L, genfer_amd)


def _div_row_wavefront_bit_exact(zs, ys, xs, OTP, GTP, OTPI, GTPI, tier, L, genfer_amd):
    x, y = rand(xs, 81, -1.0, 1.0), rand(ys, 82, -0.2, 0.2)
    y[(0,) * len(ys)] = 1.5
    deg = list(zs)
    cases = [(OTP, GTP, x, y)]
    xi, yi = x.copy(), y.copy()
    xi[tuple(min(2, s - 1) for s in xs)] = np.inf
    yi[tuple(min(1, s - 1) for s in ys)] = np.nan
    cases.append((OTP, GTP, xi, y))
    cases.append((OTP, GTP, x, yi))
    if int(np.prod(zs)) <= 20000:  # interval oracle: ~80 instructions a multiply-add
        cases.append((OTPI, GTPI, np.stack([x, x + 1e-9]), np.stack([y, y + 1e-9])))
    for O, G, a, b in cases:
        want = O.new(a, deg) / O.new(b, deg)
        got = {}
        for wf in (1, 0):
            assert L.gft_set_option(b"div_wavefront", float(wf)) == 0
            try:
                before = genfer_amd.op_stats()["launches"]
                got[wf] = G.new(a, deg) / G.new(b, deg)
                check(want, got[wf])
                if wf and zs[-1] > 64 and len(zs) <= 4 and tier == "device":
                    # one launch (+ the fill of the result with the EMPTY pattern, + the flags' memset): not one per row or slab
                    # (rank 2: k_rows_wavefront; rank 3 / 4, round 6: k_seg_wavefront)
                    assert genfer_amd.op_stats()["launches"] - before <= 8, "the long-row quotient did not take the one-launch wavefront"
            finally:
                L.gft_set_option(b"div_wavefront", 1.0)
    # the log recurrence (mt:1335-1386) through the same wavefront: slabs k0 >= 1 in one launch, slab 0 one dimension down
    ylog = rand(ys, 83, -0.2, 0.2)
    ylog[(0,) * len(ys)] = 1.5
    lcases = [(OTP, GTP, ylog)]
    if int(np.prod(zs)) <= 20000:
        lcases.append((OTPI, GTPI, np.stack([ylog, ylog + 1e-9])))
    for O, G, b in lcases:
        want = O.new(b, deg).log()
        for wf in (1, 0):
            assert L.gft_set_option(b"div_wavefront", float(wf)) == 0
            try:
                check(want, G.new(b, deg).log())
            finally:
                L.gft_set_option(b"div_wavefront", 1.0)
    # the exp recurrence (mt:1271-1300) likewise: slabs k0 >= 1 in one launch — no division, the row sum / k0.  (Large f64
    # exponentials take the right-looking tiled path, 1e-10 contract: switched off here so that both forms keep the order.)
    xe = rand(xs, 84, -0.3, 0.3)
    ecases = [(OTP, GTP, xe)]
    if int(np.prod(zs)) <= 20000:
        ecases.append((OTPI, GTPI, np.stack([xe, xe + 1e-9])))
    L.gft_set_option(b"exp_right", 0.0)
    L.gft_set_option(b"tiled_min_macs", 1e30)
    try:
        for O, G, b in ecases:
            want = O.new(b, deg).exp()
            for wf in (1, 0):
                assert L.gft_set_option(b"div_wavefront", float(wf)) == 0
                try:
                    check(want, G.new(b, deg).exp())
                finally:
                    L.gft_set_option(b"div_wavefront", 1.0)
    finally:
        L.gft_set_option(b"exp_right", 1.0)
        L.gft_set_option(b"tiled_min_macs", 2e5)


RB_SHAPES = [
    # (x shape, y shape, result shape): the last axis of y spans the result's, 64 <= last axis <= 256
    ((5, 7, 64), (5, 7, 64), (5, 7, 64)),
    ((4, 5, 37), (3, 9, 100), (4, 9, 100)),        # narrow x rows (partial last chunk of 8), odd pair axis, short lead axis
    ((9, 128), (9, 128), (9, 128)),                # rank 2: the pair axis is axis 0
    ((3, 1, 129), (3, 6, 129), (3, 6, 129)),       # one x row per slab; three waves, the last one almost empty
    ((3, 3, 5, 64), (2, 3, 5, 64), (3, 3, 5, 64)),  # rank 4
    ((2, 6, 256), (2, 1, 256), (2, 6, 256)),       # one y row per slab, the longest rows
    ((9, 11, 40), (9, 11, 40), (9, 11, 40)),       # rows shorter than 64: the row-pair form only (else k_conv_staged)
    ((20, 19, 48), (20, 19, 48), (20, 19, 48)),    # nine y tiles, the last ones partly outside y
    ((70, 3, 33), (70, 2, 40), (70, 3, 40)),       # row-pair form: a tall thin lane tile, x rows end inside a chunk of 8
    ((2, 11, 13, 72), (3, 9, 13, 72), (3, 11, 13, 72)),  # rank 4 with compact x / y on the outer axes
    ((13, 13, 13), (13, 13, 13), (13, 13, 13)),    # short rows (row-pair form only): four column blocks of 4, the last one partial
    ((5, 6, 7, 9), (5, 6, 7, 9), (5, 6, 7, 9)),
]


@pytest.mark.parametrize("xs,ys,zs", RB_SHAPES)
def test_interval_rows_register_blocked_bit_exact(xs, ys, zs, OTPI, GTPI):
    """The interval product's three reference-order kernels against the oracle: k_conv_staged, the row-pair form (k_pair_sums
    + k_pair_collect: every row sum an independent task, ordered additions in a second pass) and k_conv_rows_rb.
    Large interval products run on k_conv_rows_rb (two outputs per lane sharing each y read, x through scalar loads,
    regimes from per-row flags): every output still receives its row sums in the reference's order (mt:971-1012) =>
    bit-exact against the oracle and identical to k_conv_staged, for positive data (positive regime), mixed-sign data
    (finite regime), data with exact zeros / ones / infinities (general regime) and sums that leave their regime
    (underflow to zero, overflow: recomputed)."""
    import genfer_amd

    L = genfer_amd.lib()
    deg = list(zs)

    def iv(lo, w):
        return np.stack([lo, lo + w * np.abs(lo) + 1e-300])

    xp, yp = rand(xs, 91, 0.1, 1.0), rand(ys, 92, 0.1, 1.0)
    xm, ym = rand(xs, 93, -1.0, 1.0), rand(ys, 94, -1.0, 1.0)
    cases = [(iv(xp, 1e-15), iv(yp, 1e-15)), (iv(xm, 1e-15), iv(ym, 1e-15))]
    # special points: exact zeros and ones (short-circuits of iv:164-190), an infinity, a NaN
    xz, yz = iv(xm, 1e-15), iv(yp, 1e-15)
    xz[:, tuple(0 for _ in xs)] = 0.0
    xz[(slice(None),) + tuple(min(1, s - 1) for s in xs)] = 1.0
    yz[(slice(None),) + tuple(min(2, s - 1) for s in ys)] = 0.0
    cases.append((xz, yz))
    xi = iv(xp, 1e-15)
    xi[1][tuple(min(1, s - 1) for s in xs)] = np.inf
    yn = iv(yp, 1e-15)
    yn[(slice(None),) + tuple(s - 1 for s in ys)] = np.nan
    cases.append((xi, yn))
    # positive data whose sums leave the positive regime: products underflow to zero / overflow
    cases.append((iv(xp * 1e-200, 1e-15), iv(yp * 1e-200, 1e-15)))
    cases.append((iv(xp * 1e200, 1e-15), iv(yp * 1e200, 1e-15)))
    L.gft_set_option(b"host_max_elems", 0.0)  # everything on the device
    try:
        for a, b in cases:
            want = OTPI.new(a, deg) * OTPI.new(b, deg)
            # (threshold, row-pair mode, its workspace cap): the fused rows kernel, k_conv_staged, the row-pair form (rows of
            # <= 128; longer ones fall through to k_conv_staged), and a cap too small for it (back to k_conv_staged)
            # ... and caps that cut it into slab ranges of the leading axis (round 5: the bounded workspace; ranges of 1 - 8
            # slabs for rank 3, of any height for rank 4 — where one slab alone exceeds the cap the product falls back)
            # ... the ranges alternating between two lanes (two streams, half the cap each), forced and never; rank 4 under the
            # smallest caps: ranges of axis 0 inside one slab of the leading axis
            for thr, pairs, cap, lanes in ((0.0, 0.0, 0.0, -1.0), (-1.0, 0.0, 0.0, -1.0), (-1.0, 2.0, 0.0, -1.0), (-1.0, 2.0, 4096.0, -1.0),
                                           (-1.0, 2.0, 4.0e6, -1.0), (-1.0, 2.0, 6.0e5, -1.0), (-1.0, 2.0, 4.0e6, 1.0), (-1.0, 2.0, 1.2e6, 1.0),
                                           (-1.0, 2.0, 6.0e5, 0.0), (-1.0, 2.0, 1.0e5, -1.0), (-1.0, 2.0, 2.0e5, 1.0)):
                assert L.gft_set_option(b"conv_rb_min_macs", thr) == 0
                assert L.gft_set_option(b"conv_rb_pairs", pairs) == 0
                assert L.gft_set_option(b"conv_rb_pairs_cap", cap) == 0
                assert L.gft_set_option(b"conv_rb_pairs_lanes", lanes) == 0
                try:
                    check(want, GTPI.new(a, deg) * GTPI.new(b, deg))
                finally:
                    L.gft_set_option(b"conv_rb_min_macs", 1.5e10)
                    L.gft_set_option(b"conv_rb_pairs", -1.0)
                    L.gft_set_option(b"conv_rb_pairs_cap", 0.0)
                    L.gft_set_option(b"conv_rb_pairs_lanes", -1.0)
    finally:
        L.gft_set_option(b"host_max_elems", -1.0)


def test_interval_row_pair_workspace_is_bounded(OTPI, GTPI):
    """The row-pair form of the interval product under a small cap: a 40^3 product whose row sums (0.44 GB) exceed a 32 MB cap
    runs in slab ranges of its leading axis from one bounded workspace — the same bits as the oracle (per output the terms
    arrive in the reference's order whatever the cut), and what the library holds stays bounded: gft_pool_stats' peak grows
    by less than cap + the operands and the result."""
    import genfer_amd

    L = genfer_amd.lib()
    shape = (40, 40, 40)
    lo_x, lo_y = rand(shape, 291, 0.1, 1.0), rand(shape, 292, 0.1, 1.0)
    a, b = np.stack([lo_x, lo_x * (1 + 1e-15)]), np.stack([lo_y, lo_y * (1 + 1e-15)])
    want = OTPI.new(a, list(shape)) * OTPI.new(b, list(shape))
    cap = 32.0 * 2**20
    L.gft_set_option(b"host_max_elems", 0.0)
    assert L.gft_set_option(b"conv_rb_pairs", 2.0) == 0
    assert L.gft_set_option(b"conv_rb_pairs_cap", cap) == 0
    try:
        gx, gy = GTPI.new(a, list(shape)), GTPI.new(b, list(shape))
        before = genfer_amd.pool_stats()
        staged0 = genfer_amd.op_stats()["staged"]
        got = gx * gy
        L.gft_synchronize()
        after = genfer_amd.pool_stats()
        check(want, got)
        assert genfer_amd.op_stats()["staged"] > staged0
        # (peak includes the kernels' workspaces; earlier tests of this process may have grown it: only its GROWTH is bounded here)
        assert after["peak"] - max(before["peak"], before["in_use"]) <= cap + 4 * 2 * 8 * 40**3 + (8 << 20), (before, after)
    finally:
        L.gft_set_option(b"conv_rb_pairs", -1.0)
        L.gft_set_option(b"conv_rb_pairs_cap", 0.0)
        L.gft_set_option(b"host_max_elems", -1.0)


@pytest.mark.parametrize("xs,ys,zs", RB_SHAPES + [((3, 5, 200), (3, 5, 200), (3, 5, 200)), ((40, 40, 24), (40, 40, 24), (40, 40, 24))])
def test_f64_reference_order_product_as_row_pair_sums_bit_exact(xs, ys, zs, OTP, GTP):
    """The f64 reference-order product (conv mode 3: what the interpreter takes where the tiled kernel does not apply, and
    what a caller who wants the reference's bits sets) as row-pair sums: every row sum formed from zero in ascending j
    with separate multiply and add, the row sums added in the reference's order (mt:971-1012) => the oracle's bits,
    including the sign of zeros (the reference's sums start from +0: a -0 product becomes +0) and non-finite values;
    against the LDS-staged kernel (row-pair form off) as well."""
    import genfer_amd

    L = genfer_amd.lib()
    deg = list(zs)
    x, y = rand(xs, 191, -1.0, 1.0), rand(ys, 192, -1.0, 1.0)
    xz, yz = x.copy(), y.copy()
    xz.flat[::7] = 0.0   # exact zeros against negative partners: -0 products inside the sums
    yz.flat[::5] = -0.0
    xz[tuple(0 for _ in xs)] = -1.0
    yz[tuple(0 for _ in ys)] = 0.0  # the constant coefficient: a single -0 term
    xi = x.copy()
    xi[tuple(min(1, n - 1) for n in xs)] = np.inf
    yn = y.copy()
    yn[tuple(n - 1 for n in ys)] = np.nan
    L.gft_set_option(b"host_max_elems", 0.0)
    assert L.gft_set_conv_mode(3) == 0
    try:
        for a, b in ((x, y), (xz, yz), (xi, yn), (x * 1e-200, y * 1e-200), (x * 1e200, y * 1e160)):
            want = (OTP.new(a, deg) * OTP.new(b, deg)).array()
            # (the capped ones: slab ranges of the leading axis; lanes 1: the ranges alternate between two streams)
            for pairs, cap, lanes in ((2.0, 0.0, -1.0), (0.0, 0.0, -1.0), (2.0, 2.0e6, -1.0), (2.0, 3.0e5, 0.0), (2.0, 6.0e5, 1.0)):
                assert L.gft_set_option(b"conv_rb_pairs", pairs) == 0
                assert L.gft_set_option(b"conv_rb_pairs_cap", cap) == 0
                assert L.gft_set_option(b"conv_rb_pairs_lanes", lanes) == 0
                try:
                    got = (GTP.new(a, deg) * GTP.new(b, deg)).array()
                finally:
                    L.gft_set_option(b"conv_rb_pairs", -1.0)
                    L.gft_set_option(b"conv_rb_pairs_cap", 0.0)
                    L.gft_set_option(b"conv_rb_pairs_lanes", -1.0)
                ok = (np.asarray(want).view(np.uint64) == np.asarray(got).view(np.uint64)) | (np.isnan(want) & np.isnan(got))
                assert np.all(ok), (pairs, cap, np.asarray(want)[~ok][:4], np.asarray(got)[~ok][:4])
    finally:
        L.gft_set_conv_mode(0)
        L.gft_set_option(b"host_max_elems", -1.0)


@pytest.mark.parametrize("async_launch", [1, 0])
def test_failed_kernel_launch_is_reported_not_swallowed(OTP, GTP, async_launch):
    """Launches are issued by the library's worker thread (gft_launch.hpp) and HIP's last-error state is per thread: the
    worker latches the first failing launch with the kernel's name, and the next drain — a value inspection or
    gft_synchronize — raises it through gft_last_error().  The test knob makes ONE launch request 1 MB of LDS."""
    import genfer_amd

    L = genfer_amd.lib()
    assert L.gft_set_option(b"host_max_elems", 0.0) == 0  # everything on the device
    assert L.gft_set_option(b"async_launch", float(async_launch)) == 0
    try:
        x = rand((40, 40), 5)
        gx = GTP.new(x, [40, 40])
        assert L.gft_synchronize() == 0
        assert L.gft_set_option(b"debug_fail_next_launch", 1.0) == 0
        gs = gx + gx  # one kernel launch, which fails on the worker
        rc = L.gft_synchronize()
        assert rc != 0, "a failed launch went unnoticed"
        msg = (L.gft_last_error() or b"").decode()
        assert "HIP error" in msg and "launching" in msg and "k_" in msg, msg
        del gs
        # the failure is reported once; the runtime keeps working afterwards
        assert L.gft_synchronize() == 0
        ox = OTP.new(x, [40, 40])
        check(ox + ox, gx + gx)
        # a value inspection surfaces it too
        assert L.gft_set_option(b"debug_fail_next_launch", 1.0) == 0
        gs = gx + gx
        with pytest.raises(genfer_amd.TaylorError):
            gs.array()
        assert L.gft_synchronize() == 0
    finally:
        L.gft_set_option(b"debug_fail_next_launch", 0.0)
        L.gft_set_option(b"async_launch", 1.0)
        L.gft_set_option(b"host_max_elems", -1.0)
