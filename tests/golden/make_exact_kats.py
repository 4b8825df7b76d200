#!/usr/bin/env python3
"""Generate tests/golden/exact_kats.json — independent known answers for the Taylor algebra.

Everything here is exact rational arithmetic (fractions.Fraction) on truncated multivariate
power series, written from the mathematical definitions (Cauchy product, series division,
exp/log via the logarithmic-derivative ODE, Horner substitution), NOT from the reference's
code and not through the oracle.  Inputs are small integers / dyadic rationals (exactly
representable in f64); expected outputs are stored as correctly rounded doubles
(float(Fraction) rounds to nearest).

Case kinds:
  mul        compact shapes + degree caps, d = 1..4; integer data => every f64 operation in any
             summation order is exact => expected results are bit-exact targets.
  add/sub    ditto.
  div/exp/log/subst   rational results => compared with a tolerance.
  iv_chain   q = (x*y + w) / d on small dyadic-rational tensors, exact.  The known answer is stored as the two doubles
             that bracket each exact coefficient (r_down <= r <= r_up, adjacent or equal) plus a magnitude scale: an
             Interval<F64> evaluation of the same chain on point intervals must ENCLOSE r (lo <= r_down, r_up <= hi —
             soundness of interval.rs's widen-by-one-ulp arithmetic) and be at most ~1e-12 * mag wide (tightness).
             Independent of the reference code, of the oracle and of every Interval implementation in this repo.

Run:  python tests/golden/make_exact_kats.py   (deterministic; seed fixed)
"""
import itertools
import json
import math
import os
import random
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
rng = random.Random(20231016)


def indices(shape):
    return itertools.product(*[range(s) for s in shape])


def zeros(shape):
    return {idx: Fraction(0) for idx in indices(shape)}


def rand_poly(shape, lo=-4, hi=5, denom=1):
    return {idx: Fraction(rng.randint(lo, hi), denom) for idx in indices(shape)}


def to_nested(p, shape):
    def rec(prefix, d):
        if d == len(shape):
            return float(p[tuple(prefix)])
        return [rec(prefix + [i], d + 1) for i in range(shape[d])]

    return rec([], 0)


def trunc_mul(x, xs, y, ys, zs):
    z = zeros(zs)
    for j in indices(xs):
        for m in indices(ys):
            k = tuple(a + b for a, b in zip(j, m))
            if all(kk < s for kk, s in zip(k, zs)):
                z[k] += x[j] * y[m]
    return z


def pad(p, shape, full):
    q = zeros(full)
    for idx in indices(shape):
        q[idx] = p[idx]
    return q


def series_div(x, y, full):
    """q with q*y = x (mod truncation `full`); y[0..0] != 0. Solve in graded-lex order."""
    q = zeros(full)
    y0 = y[tuple(0 for _ in full)]
    for k in indices(full):  # row-major order is compatible with the partial order
        acc = x[k]
        for j in indices(tuple(kk + 1 for kk in k)):
            if j == k:
                continue
            m = tuple(a - b for a, b in zip(k, j))
            acc -= q[j] * y[m]
        q[k] = acc / y0
    return q


def deriv_axis0_weighted(p, full):
    """(theta_0 p)[k] = k0 * p[k] (Euler operator along axis 0)."""
    return {k: k[0] * p[k] for k in indices(full)}


def cases():
    out = []
    # ---- mul / add / sub with compact shapes and degree caps --------------------------
    specs = [
        ((5,), (7,), (8,)),
        ((6,), (6,), (20,)),
        ((3, 4), (4, 2), (5, 5)),
        ((4, 4), (4, 4), (4, 4)),
        ((2, 5), (5, 1), (6, 5)),
        ((3, 3, 3), (3, 3, 3), (4, 5, 3)),
        ((4, 1, 3), (2, 3, 3), (5, 3, 4)),
        ((2, 3, 2, 3), (3, 2, 2, 2), (4, 4, 3, 4)),
        ((3, 3, 3, 3), (3, 3, 3, 3), (3, 3, 3, 3)),
    ]
    for xs, ys, deg in specs:
        x, y = rand_poly(xs), rand_poly(ys)
        zs = tuple(min(a + b - 1, d) for a, b, d in zip(xs, ys, deg))
        # inputs are first truncated to deg (mt:1028-1029)
        xt = tuple(min(a, d) for a, d in zip(xs, deg))
        yt = tuple(min(a, d) for a, d in zip(ys, deg))
        z = trunc_mul(x, xt, y, yt, zs)
        out.append(
            dict(kind="mul", x=to_nested(x, xs), y=to_nested(y, ys), x_deg=list(deg), y_deg=list(deg),
                 z=to_nested(z, zs), z_shape=list(zs), exact=True)
        )
        ms = tuple(min(max(a, b), d) for a, b, d in zip(xs, ys, deg))
        s = zeros(ms)
        dd = zeros(ms)
        for k in indices(ms):
            xv = x[k] if all(kk < a for kk, a in zip(k, xs)) else 0
            yv = y[k] if all(kk < a for kk, a in zip(k, ys)) else 0
            s[k] = xv + yv
            dd[k] = xv - yv
        out.append(
            dict(kind="addsub", x=to_nested(x, xs), y=to_nested(y, ys), x_deg=list(deg), y_deg=list(deg),
                 sum=to_nested(s, ms), diff=to_nested(dd, ms), shape=list(ms), exact=True)
        )

    # ---- div: full shapes, constant term nonzero --------------------------------------
    for full in [(8,), (4, 4), (3, 3, 3), (2, 3, 2, 2)]:
        x = rand_poly(full, denom=4)
        y = rand_poly(full, denom=2)
        y[tuple(0 for _ in full)] = Fraction(rng.choice([2, 3, -2, 5]))
        q = series_div(x, y, full)
        mag = zeros(full)  # magnitude scale: |x| + sum |q_j||y_m| for a normwise tolerance
        for k in indices(full):
            acc = abs(x[k])
            for j in indices(tuple(kk + 1 for kk in k)):
                m = tuple(a - b for a, b in zip(k, j))
                acc += abs(q[j] * y[m])
            mag[k] = acc / abs(y[tuple(0 for _ in full)])
        out.append(dict(kind="div", x=to_nested(x, full), y=to_nested(y, full), deg=list(full),
                        q=to_nested(q, full), scale=to_nested(mag, full), exact=False))

    # ---- exp / log: f = c0 + (stuff); check  exp: theta0(E) = theta0(f) * E  exactly is awkward
    # in rationals because exp(c0) is irrational, so use c0 = 0 for exp and c0 = 1 for log.
    for full in [(7,), (4, 4), (3, 3, 3)]:
        f = rand_poly(full, lo=-2, hi=3, denom=2)
        zero = tuple(0 for _ in full)
        f[zero] = Fraction(0)
        # exp(f) = sum f^n / n!  (f has zero constant term => finite sum under truncation)
        total_deg = sum(s - 1 for s in full)
        e = zeros(full)
        e[zero] = Fraction(1)
        term = dict(e)
        for n in range(1, total_deg + 1):
            term = trunc_mul(term, full, f, full, full)
            term = {k: v / n for k, v in term.items()}
            for k in e:
                e[k] += term[k]
        out.append(dict(kind="exp", f=to_nested(f, full), deg=list(full), e=to_nested(e, full), exact=False))
        # log(1+f) = sum (-1)^(n+1) f^n / n
        g = dict(f)
        g[zero] = Fraction(1)
        l = zeros(full)
        powf = zeros(full)
        powf[zero] = Fraction(1)
        for n in range(1, total_deg + 1):
            powf = trunc_mul(powf, full, f, full, full)
            for k in l:
                l[k] += Fraction((-1) ** (n + 1), n) * powf[k]
        out.append(dict(kind="log", g=to_nested(g, full), deg=list(full), l=to_nested(l, full), exact=False))

    # ---- subst_var (Horner) with integer data => exact ---------------------------------
    for full, v in [((4,), 0), ((3, 3), 0), ((3, 3), 1), ((3, 2, 3), 1)]:
        p = rand_poly(full, lo=-3, hi=3)
        s = rand_poly(full, lo=-2, hi=2)
        res = zeros(full)
        # sum_i p[.., i at v, ..] * s^i, truncated
        spow = zeros(full)
        spow[tuple(0 for _ in full)] = Fraction(1)
        for i in range(full[v]):
            slab = zeros(full)
            for k in indices(full):
                if k[v] == 0:
                    kk = list(k)
                    kk[v] = i
                    slab[k] = p[tuple(kk)]
            contrib = trunc_mul(slab, full, spow, full, full)
            for k in res:
                res[k] += contrib[k]
            spow = trunc_mul(spow, full, s, full, full)
        out.append(dict(kind="subst", p=to_nested(p, full), s=to_nested(s, full), v=v, deg=list(full),
                        r=to_nested(res, full), exact=True))
    # ---- interval enclosure chains: q = (x*y + w) / d, exact -------------------------------------
    def bracket(r):
        f = float(r)  # nearest double
        lo = f if Fraction(f) <= r else math.nextafter(f, -math.inf)
        hi = f if Fraction(f) >= r else math.nextafter(f, math.inf)
        assert Fraction(lo) <= r <= Fraction(hi)
        return lo, hi

    for full in [(6,), (4, 3), (3, 3, 2), (2, 2, 2, 2)]:
        x = rand_poly(full, lo=-6, hi=7, denom=8)
        y = rand_poly(full, lo=-6, hi=7, denom=4)
        w = rand_poly(full, lo=-9, hi=9, denom=16)
        d = rand_poly(full, lo=-3, hi=4, denom=2)
        zero = tuple(0 for _ in full)
        d[zero] = Fraction(rng.choice([3, -5, 7]), 2)
        num = trunc_mul(x, full, y, full, full)
        absnum = trunc_mul({k: abs(v) for k, v in x.items()}, full, {k: abs(v) for k, v in y.items()}, full, full)
        for k in num:
            num[k] += w[k]
            absnum[k] += abs(w[k])
        q = series_div(num, d, full)
        mag = zeros(full)
        for k in indices(full):
            acc = absnum[k]
            for j in indices(tuple(kk + 1 for kk in k)):
                if j == k:
                    continue
                m = tuple(a - b for a, b in zip(k, j))
                acc += mag[j] * abs(d[m])
            mag[k] = acc / abs(d[zero])
        lo, hi = zeros(full), zeros(full)
        for k in indices(full):
            lo[k], hi[k] = bracket(q[k])
        out.append(dict(kind="iv_chain", x=to_nested(x, full), y=to_nested(y, full), w=to_nested(w, full), d=to_nested(d, full),
                        deg=list(full), r_down=to_nested(lo, full), r_up=to_nested(hi, full), mag=to_nested(mag, full), exact=False))
    return out


if __name__ == "__main__":
    data = dict(
        _provenance="generated by tests/golden/make_exact_kats.py (exact Fraction arithmetic, seed 20231016); "
        "independent of both the reference code and the oracle",
        cases=cases(),
    )
    with open(os.path.join(HERE, "exact_kats.json"), "w") as f:
        json.dump(data, f)
    print("wrote", len(data["cases"]), "cases")
