// Host tier of the size-threshold dispatch (reference: every TaylorPoly operation is host code,
// src/multivariate_taylor.rs).  Genfer programs issue 10^5-10^6 operations on tensors of a few hundred elements
// (switchpoint: 3.8e5 operations, almost all <= 512 elements); a kernel launch costs ~4 us and a host round trip
// ~15 us, the same operation on 256 elements costs the host ~0.3 us.  So tensors whose operands are all host-resident
// and whose result stays below the crossover (gft_set_option "host_max_elems" / "host_max_macs") are computed right
// here, by plain loops over the SAME element functors the kernels use (gft_elem.hpp, one source for both passes,
// -ffp-contract=off): a result carries the same bits whichever side produced it.  Signatures mirror K<E> in
// gft_kernels.hpp without the stream; every loop runs in the order the corresponding kernel documents.
//
// This is part of the product (libgftaylor), not of the test oracle; nothing here includes oracle/.  The library
// still refuses to initialise without a gfx950 device.
#pragma once
#include <cstring>

#include "gft_kernels.hpp"

#include <cstdio>
namespace gft {
// GFT_TRACE_API: host-tier Horner steps by regime {positive constants, sign-known c, general} and elements that fell back
extern unsigned long long g_host_horner_stats[16];  // [0..3] steps by regime / semi misses; [4..7] elements of semi / fin runs that held / failed; [8..9] elements of lines without runs; [10..13] element form: stored by semi / fin / positive / general
extern int g_host_simd;  // the runs' vector clones: -1 the widest the CPU has (AVX-512, AVX2; the default), 1 AVX2 at most, 0 none (tests compare the builds of the same loop)
extern bool g_host_horner_runs;  // the finite regime of the host Horner step in runs of equal terms (tests switch it off to compare)


template <class E>
struct HK {
    typedef typename E::V V;

    static size_t numel(const Shape& s) {
        size_t n = 1;
        for (int i = 0; i < s.nd; ++i) n *= s.d[i];
        return n;
    }

    // k_gather
    static void gather(const double* src, size_t src_plane, double* out, size_t out_plane, const GatherArgs& a) {
        const size_t total = numel(a.out);
        for (size_t lin = 0; lin < total; ++lin) {
            size_t r = lin, soff = 0;
            bool valid = true;
            unsigned kaxis = 0;
            for (int ax = a.out.nd - 1; ax >= 0; --ax) {
                const unsigned d = a.out.d[ax];
                const unsigned k = (unsigned)(r % d);
                r /= d;
                const long long si = (long long)k + a.shift[ax];
                if (si < 0 || si >= (long long)a.src_len[ax]) valid = false;
                soff += (size_t)(si < 0 ? 0 : si) * a.src_stride[ax];
                if (ax == a.tab_axis) kaxis = k;
            }
            if (valid && a.keep && !a.keep[kaxis]) valid = false;
            V v = E::zero();
            if (valid) {
                v = E::ld(src, src_plane, soff);
                switch (a.op) {
                    case OP_MUL_S: v = E::mul(v, E::from(a.s)); break;
                    case OP_DIV_S: v = E::div(v, E::from(a.s)); break;
                    case OP_LMUL_S: v = E::mul(E::from(a.s), v); break;
                    case OP_NEG: v = E::neg(v); break;
                    case OP_MUL_TAB:
                    case OP_MUL_HTAB: v = E::mul(v, E::ld(a.tab, a.tab_plane, kaxis)); break;
                    case OP_MUL_TAB_LMUL_S: v = E::mul(E::from(a.s), E::mul(v, E::ld(a.tab, a.tab_plane, kaxis))); break;
                    case OP_MUL_POW: {
                        const V mv = a.tab ? E::ld(a.tab, a.tab_plane, 0) : E::from(a.s);
                        V f = E::one();
                        for (unsigned i = 0; i < kaxis; ++i) f = E::mul(f, mv);
                        v = E::mul(v, f);
                        break;
                    }
                    default: break;
                }
            }
            E::st(out, out_plane, lin, v);
        }
    }

    // k_addsub_padded: (0 + a) (+|-) b on leading blocks
    static void addsub_padded(const DView& out, const DView& a, const DView& b, int subtract) {
        const size_t total = numel(out.sh);
        for (size_t lin = 0; lin < total; ++lin) {
            size_t r = lin, aoff = 0, boff = 0, astr = 1, bstr = 1;
            bool ina = true, inb = true;
            for (int ax = out.sh.nd - 1; ax >= 0; --ax) {
                const unsigned d = out.sh.d[ax];
                const unsigned k = (unsigned)(r % d);
                r /= d;
                if (k >= a.sh.d[ax]) ina = false;
                if (k >= b.sh.d[ax]) inb = false;
                aoff += k * astr;
                boff += k * bstr;
                astr *= a.sh.d[ax];
                bstr *= b.sh.d[ax];
            }
            V v = E::zero();
            if (ina) v = E::add(v, E::ld(a.p, a.plane, aoff));
            if (inb) {
                const V w = E::ld(b.p, b.plane, boff);
                v = subtract ? E::sub(v, w) : E::add(v, w);
            }
            E::st(out.p, out.plane, lin, v);
        }
    }

    static void add_scaled_padded(const DView& out, const DView& a, const DView& b, Scalar2 c) {
        const size_t total = numel(out.sh);
        const V cv = E::from(c);
        for (size_t lin = 0; lin < total; ++lin) {
            size_t r = lin, aoff = 0, boff = 0, astr = 1, bstr = 1;
            bool ina = true, inb = true;
            for (int ax = out.sh.nd - 1; ax >= 0; --ax) {
                const unsigned d = out.sh.d[ax];
                const unsigned k = (unsigned)(r % d);
                r /= d;
                if (k >= a.sh.d[ax]) ina = false;
                if (k >= b.sh.d[ax]) inb = false;
                aoff += k * astr;
                boff += k * bstr;
                astr *= a.sh.d[ax];
                bstr *= b.sh.d[ax];
            }
            V v = E::zero();
            if (ina) v = E::add(v, E::ld(a.p, a.plane, aoff));
            if (inb) v = E::add(v, E::mul(cv, E::ld(b.p, b.plane, boff)));
            E::st(out.p, out.plane, lin, v);
        }
    }

    // k_copy_first
    static void copy_first(const double* src, size_t sp, double* dst, size_t dp, size_t n, int op, Scalar2 sv) {
        for (size_t i = 0; i < n; ++i) {
            V x = E::ld(src, sp, i);
            if (i == 0) {
                const V y = E::from(sv);
                x = (op == FIRST_ADD) ? E::add(x, y) : E::sub(x, y);
            }
            if (op == FIRST_SUB_NEG_ALL) x = E::neg(x);
            E::st(dst, dp, i, x);
        }
    }

    // two host scalars (k_scalar_imm's table)
    static V scalar_imm(int kind, V x, V y) {
        switch (kind) {
            case IMM_LMUL: return E::mul(y, x);
            case IMM_MUL: return E::mul(x, y);
            case IMM_DIV: return E::div(x, y);
            case IMM_NEG: return E::neg(x);
            case IMM_ADD: return E::add(x, y);
            case IMM_SUB: return E::sub(x, y);
            default: return E::neg(E::sub(x, y));
        }
    }

    // k_linear_scan: mask of the axes the tensor is "linear in"; c = t[0], m = t[e_v] of the first surviving axis
    static unsigned linear_scan(const DView& t, unsigned axes_mask, double c[2], double m[2]) {
        const size_t total = numel(t.sh);
        unsigned local = axes_mask;
        for (size_t lin = 0; lin < total && local != 0; ++lin) {
            if (E::is_zero(E::ld(t.p, t.plane, lin))) continue;
            size_t r = lin;
            int nonzero_axes = 0, which = -1;
            bool unit = true;
            for (int ax = t.sh.nd - 1; ax >= 0; --ax) {
                const unsigned d = t.sh.d[ax];
                const unsigned k = (unsigned)(r % d);
                r /= d;
                if (k != 0) {
                    nonzero_axes++;
                    which = ax;
                    if (k != 1) unit = false;
                }
            }
            if (nonzero_axes == 0) continue;
            if (nonzero_axes == 1 && unit) local &= (1u << which);
            else local = 0;
        }
        c[0] = c[1] = m[0] = m[1] = 0.0;
        if (local) {
            const int ax = __builtin_ffs((int)local) - 1;
            size_t stride = 1;
            for (int i = t.sh.nd - 1; i > ax; --i) stride *= t.sh.d[i];
            c[0] = t.p[0];
            m[0] = t.p[stride];
            if (E::W == 2) {
                c[1] = t.p[t.plane];
                m[1] = t.p[t.plane + stride];
            }
        }
        return local;
    }

    static V apply_map(V x, int op, unsigned u, V s) {
        switch (op) {
            case MAP_NEG: return E::neg(x);
            case MAP_DIV_U32: return E::div(x, E::from_u32(u));
            case MAP_MUL_U32: return E::mul(x, E::from_u32(u));
            case MAP_MUL_S: return E::mul(x, s);
            case MAP_DIV_S: return E::div(x, s);
            case MAP_LMUL_S: return E::mul(s, x);
            default: return x;
        }
    }
    static void map_inplace(double* p, size_t plane, size_t n, int op, unsigned u, Scalar2 s) {
        const V sv = E::from(s);
        for (size_t i = 0; i < n; ++i) E::st(p, plane, i, apply_map(E::ld(p, plane, i), op, u, sv));
    }

    // k_block_op
    static void block_op(const DView& dst, const DView& src, int op, unsigned u) {
        const size_t total = numel(src.sh);
        for (size_t lin = 0; lin < total; ++lin) {
            size_t r = lin, doff = 0, dstr = 1;
            for (int ax = src.sh.nd - 1; ax >= 0; --ax) {
                const unsigned d = src.sh.d[ax];
                const unsigned k = (unsigned)(r % d);
                r /= d;
                doff += k * dstr;
                dstr *= dst.sh.d[ax];
            }
            const V x = E::ld(src.p, src.plane, lin);
            V rv;
            if (op == BLK_ASSIGN) rv = x;
            else {
                const V cur = E::ld(dst.p, dst.plane, doff);
                rv = (op == BLK_ADD) ? E::add(cur, x) : E::add(cur, E::mul(E::from_u32(u), x));
            }
            E::st(dst.p, dst.plane, doff, rv);
        }
    }

    // k_exp_1d / k_log_1d (seed = exp / ln of xs[0], formed by the caller)
    static void exp_1d(const double* xs, size_t xp, unsigned nx, double* res, size_t rp, unsigned n, Scalar2 seed) {
        if (n == 0) return;
        E::st(res, rp, 0, E::from(seed));
        for (unsigned k = 1; k < n; ++k) {
            V sum = E::zero();
            const unsigned hi = nx < k + 1 ? nx : k + 1;
            for (unsigned j = 1; j < hi; ++j)
                sum = E::add(sum, E::mul(E::mul(E::ld(xs, xp, j), E::from_u32(j)), E::ld(res, rp, k - j)));
            E::st(res, rp, k, E::div(sum, E::from_u32(k)));
        }
    }
    static void log_1d(const double* xs, size_t xp, unsigned nx, double* res, size_t rp, unsigned n, Scalar2 seed) {
        if (n == 0) return;
        const V x0 = E::ld(xs, xp, 0);
        E::st(res, rp, 0, E::from(seed));
        for (unsigned k = 1; k < n; ++k) {
            V sum = E::zero();
            unsigned lo = (k + 1 > nx) ? (k + 1 - nx) : 0;
            if (lo < 1) lo = 1;
            for (unsigned j = lo; j < k; ++j)
                sum = E::add(sum, E::mul(E::mul(E::ld(xs, xp, k - j), E::ld(res, rp, j)), E::from_u32(j)));
            const V xk = k < nx ? E::ld(xs, xp, k) : E::zero();
            const V num = E::sub(E::mul(xk, E::from_u32(k)), sum);
            E::st(res, rp, k, E::div(E::div(num, x0), E::from_u32(k)));
        }
    }
    // k_div_1d_serial
    static void div_1d(const double* xs, size_t xp, unsigned nx, const double* ys, size_t yp, unsigned ny, double* res,
                       size_t rp, unsigned n) {
        const V y0 = E::ld(ys, yp, 0);
        for (unsigned k = 0; k < n; ++k) {
            V cur = E::zero();
            const unsigned lo = (k + 1 > ny) ? (k + 1 - ny) : 0;
            for (unsigned j = lo; j < k; ++j) cur = E::add(cur, E::mul(E::ld(res, rp, j), E::ld(ys, yp, k - j)));
            cur = E::neg(cur);
            if (k < nx) cur = E::add(cur, E::ld(xs, xp, k));
            E::st(res, rp, k, E::div(cur, y0));
        }
    }

    // k_factor_table
    static void factor_table(int op, unsigned n, unsigned len, const double* m, size_t mp, double* tab, size_t tp) {
        if (op == TAB_DERIV) {
            V ff = E::one();
            for (unsigned i = 1; i <= n; ++i) ff = E::mul(ff, E::from_u32(i));
            for (unsigned k = 0; k < len; ++k) {
                E::st(tab, tp, k, ff);
                ff = E::mul(ff, E::div(E::from_u32(n + k + 1), E::from_u32(k + 1)));
            }
        } else if (op == TAB_COEFF) {
            V f = E::one();
            E::st(tab, tp, 0, f);
            for (unsigned k = 1; k < len; ++k) {
                f = E::mul(f, E::div(E::from_u32(n + k), E::from_u32(k)));
                E::st(tab, tp, k, f);
            }
        } else if (op == TAB_POW) {
            V f = E::one();
            const V mv = E::ld(m, mp, 0);
            for (unsigned k = 0; k < len; ++k) {
                E::st(tab, tp, k, f);
                f = E::mul(f, mv);
            }
        } else {
            for (unsigned k = 0; k < len; ++k) E::st(tab, tp, k, E::from_u32(k));
        }
    }

    // k_sum_axis_seq (SUM_SEQ / SUM_UNROLL8)
    static void sum_axis(const double* in, size_t ip, unsigned outer, unsigned len, unsigned inner, size_t outer_stride,
                         double* out, size_t op, int mode) {
        const size_t total = (size_t)outer * inner;
        for (size_t lin = 0; lin < total; ++lin) {
            const unsigned i = (unsigned)(lin % inner);
            const size_t o = lin / inner;
            const size_t base = o * outer_stride + i;
            V acc = E::zero();
            if (mode == SUM_UNROLL8) {
                V p[8];
                for (int u = 0; u < 8; ++u) p[u] = E::zero();
                unsigned k = 0;
                for (; k + 8 <= len; k += 8)
                    for (int u = 0; u < 8; ++u) p[u] = E::add(p[u], E::ld(in, ip, base + (size_t)(k + u) * inner));
                acc = E::add(acc, E::add(p[0], p[4]));
                acc = E::add(acc, E::add(p[1], p[5]));
                acc = E::add(acc, E::add(p[2], p[6]));
                acc = E::add(acc, E::add(p[3], p[7]));
                for (; k < len; ++k) acc = E::add(acc, E::ld(in, ip, base + (size_t)k * inner));
            } else {
                for (unsigned k = 0; k < len; ++k) acc = E::add(acc, E::ld(in, ip, base + (size_t)k * inner));
            }
            E::st(out, op, lin, acc);
        }
    }

    // k_horner_linear: one Horner step res * (c + m eps_w) + coeff_i, the reference's element order (HornerArgs).
    // `--bounds` programs whose tensors stay small (switchpoint: 73 000 substitutions of ~100 steps each) live in this
    // loop, so it avoids what does not belong to the arithmetic: the multi-index is an odometer over the outer axes with
    // a plain loop along the last one (no division per element), and interval operands in the positive regime
    // (gft_elem.hpp: probability-like values) take mul_pos / add_pos — the same operations on the same values as the
    // general forms, hence the same bits — with the general form as the fallback per element.
    static void horner_linear(const double* res, size_t rp, const double* a, size_t ap, double* out, size_t op, const HornerArgs& g) {
        const int nd = g.out.nd;
        if (nd == 0) {
            horner_linear_elem(res, rp, a, ap, out, op, g, 0, 0, g.a_base, 0, true, true, true, false, 0, 0);
            return;
        }
        const V cv = E::from(g.c), mv = E::from(g.m);
        bool pos_consts = false;
        if constexpr (E::HAS_POS) pos_consts = E::pos_ok(mv) && (g.c_zero || g.c_one || E::pos_ok(cv));
        // The `--bounds` runs of v -> c + m v have a c that is a few ulps AROUND zero (an interval with a negative and a
        // positive bound): not the positive regime, but the signs of c's bounds are known, so against a positive x the
        // reference's product (iv:164-190) is [c.lo * x.hi, c.hi * x.hi] (or the analogous pair) and its outward steps are
        // the integer steps `bits +- 1` in the direction those signs dictate — the device's lean Horner step
        // (gft_kernels.hip LeanConsts<EIv>), here for the host tier: switchpoint `--bounds` spends 2.5 of its 2.8 s in this loop.
        int semi = 0;  // bit 0: usable, bit 1: lo uses x.hi, bit 2: hi uses x.lo, bit 3: dlo = +1, bit 4: dhi = +1, bit 5: m = [1,1]
        if constexpr (E::HAS_POS) {
            // (m = [1,1], a pure shift v -> c + v: the reference's product with it returns the other operand, iv:164-190)
            if (!pos_consts && !g.c_zero && !g.c_one && (E::pos_ok(mv) || E::is_one(mv)) && E::is_finite(cv) && cv.lo <= cv.hi && !E::maybe_special(cv) &&
                cv.lo != 0.0 && cv.hi != 0.0)
                semi = 1 | (!(cv.lo >= 0.0) ? 2 : 0) | (!(cv.hi >= 0.0) ? 4 : 0) | (cv.lo < 0.0 ? 8 : 0) | (cv.hi > 0.0 ? 16 : 0) | (E::is_one(mv) ? 32 : 0);
        }
        if (semi) pos_consts = false;
        // ... and where the DATA are not positive either (switchpoint's accumulators are error intervals [-t, +t] around zero)
        // the finite regime of the device kernels (gft_elem.hpp mul_fin / widen_fin): operands that are finite and no exact
        // 0 / +-1 point take no short-circuit, the outward steps need no NaN / inf guard, and ONE test of the result (not NaN)
        // validates every term.  bit 0: usable, bit 5: m = [1,1]
        int fin = 0;
        if constexpr (E::HAS_POS) {
            if (!g.c_zero && !g.c_one && E::fin_ok(cv) && cv.lo <= cv.hi && (E::fin_ok(mv) || E::is_one(mv)) && mv.lo <= mv.hi)
                fin = 1 | (E::is_one(mv) ? 32 : 0);
        }
        g_host_horner_stats[semi ? 1 : (pos_consts ? 0 : 2)]++;
        const int last = nd - 1;
        const unsigned nlast = g.out.d[last];
        size_t outer = 1;
        for (int ax = 0; ax < last; ++ax) outer *= g.out.d[ax];
        unsigned idx[MAXD] = {0};
        size_t lin = 0;
        for (size_t o = 0; o < outer; ++o) {
            bool in_p0 = true, in_r0 = true, in_c0 = true;
            size_t roff0 = 0, aoff0 = g.a_base;
            unsigned kw0 = 0;
            for (int ax = 0; ax < last; ++ax) {
                const unsigned k = idx[ax];
                if (k >= g.sh[ax]) in_p0 = false;
                if (k >= g.rs[ax]) in_r0 = false;
                if (k >= g.oc[ax]) in_c0 = false;
                if (ax == g.w) kw0 = k;
                roff0 += (size_t)k * g.rstr[ax];
                aoff0 += (size_t)k * g.astr[ax];
            }
            if constexpr (E::HAS_POS) {
                // (with a sign-known c the element form's first choice is the `semi` regime, which needs POSITIVE data: a line whose
                // first accumulator element is not positive — switchpoint's error intervals around zero — goes straight to the runs)
                bool runs = fin && g_host_horner_runs;
                bool semi_runs = false;  // ... and a line of positive data under a sign-known c: the `semi` regime in the same runs
                if (semi && g_host_horner_runs) {
                    const V x0 = in_r0 && g.rs[last] > 0 ? E::ld(res, rp, roff0) : E::one();
                    if (E::pos_ok(x0)) {
                        runs = true;
                        semi_runs = true;
                    }
                }
                if (runs) {
                    // (round 6) the FINITE regime in RUNS: along the line the three terms of a position — res[k - 1] * m, c * res[k],
                    // the coefficient — each exist on one interval of k, so the line is a handful of runs with the same terms, each a
                    // plain loop over contiguous-stride operands (no per-element flags: switchpoint --bounds is 1.3e8 elements of this)
                    const unsigned p_end = in_p0 ? (nlast < g.sh[last] ? nlast : g.sh[last]) : 0u;
                    const unsigned t2_end = (in_r0 && !g.c_zero) ? (p_end < g.rs[last] ? p_end : g.rs[last]) : 0u;
                    unsigned t1_lo = 0, t1_hi = 0;
                    if (g.w == last) {
                        t1_lo = 1;
                        t1_hi = p_end < g.upper + 1u ? p_end : g.upper + 1u;
                    } else if (kw0 >= 1 && kw0 - 1 < g.upper) {
                        t1_hi = p_end;
                    }
                    if (t1_hi < t1_lo) t1_hi = t1_lo;
                    unsigned t3_hi = 0;
                    if (g.coeff_scalar) t3_hi = lin == 0 ? 1u : 0u;
                    else if (in_c0) t3_hi = nlast < g.oc[last] ? nlast : g.oc[last];
                    unsigned k = 0;
                    while (k < nlast) {
                        const bool t1 = k >= t1_lo && k < t1_hi, t2 = k < t2_end, t3 = k < t3_hi;
                        unsigned q = nlast;  // the run ends where one of the three changes
                        if (k < t1_lo && t1_lo < q) q = t1_lo;
                        if (k < t1_hi && t1_hi < q) q = t1_hi;
                        if (k < t2_end && t2_end < q) q = t2_end;
                        if (k < t3_hi && t3_hi < q) q = t3_hi;
                        const int sel = (t1 ? 1 : 0) | (t2 ? 2 : 0) | (t3 ? 4 : 0);
                        const size_t roff = roff0 + (size_t)k * g.rstr[last], aoff = aoff0 + (size_t)k * g.astr[last];
                        // in chunks: one verdict per chunk, so a stray element outside the regime costs its chunk, not the line.  A line of
                        // a sign-known step typically starts positive (probabilities) and ends in error intervals around zero: the
                        // `semi` regime until a chunk fails it, the finite regime from there on (both are the reference's bits where
                        // their tests pass), single elements with every regime tried only where a chunk fails both.
                        constexpr unsigned CH = 16;
                        for (unsigned kk = k; kk < q; kk += CH) {
                            const unsigned n = q - kk < CH ? q - kk : CH;
                            const size_t ro = roff + (size_t)(kk - k) * g.rstr[last], ao = aoff + (size_t)(kk - k) * g.astr[last], li = lin + (kk - k);
                            bool ok = false;
                            if (semi_runs) {
                                ok = horner_run_sel<true>(sel, res, rp, a, ap, out, op, g, li, ro, ao, last, n, semi);
                                g_host_horner_stats[ok ? 4 : 5] += n;
                                if (!ok) semi_runs = false;  // (the rest of this line: straight to the finite regime)
                            }
                            if (!ok && fin) {
                                ok = horner_run_sel<false>(sel, res, rp, a, ap, out, op, g, li, ro, ao, last, n, fin);
                                g_host_horner_stats[ok ? 6 : 7] += n;
                            }
                            if (!ok) {
                                for (unsigned e = kk; e < kk + n; ++e) {
                                    const bool in_p = in_p0 && e < g.sh[last], in_r = in_r0 && e < g.rs[last], in_c = in_c0 && e < g.oc[last];
                                    horner_linear_elem(res, rp, a, ap, out, op, g, lin + (e - k), roff0 + (size_t)e * g.rstr[last], aoff0 + (size_t)e * g.astr[last],
                                                       g.w == last ? e : kw0, in_p, in_r, in_c, pos_consts, semi, fin);
                                }
                            }
                        }
                        lin += q - k;
                        k = q;
                    }
                    for (int ax = last - 1; ax >= 0; --ax) {
                        if (++idx[ax] < g.out.d[ax]) break;
                        idx[ax] = 0;
                    }
                    continue;
                }
            }
            g_host_horner_stats[semi ? 8 : 9] += nlast;
            for (unsigned k = 0; k < nlast; ++k, ++lin) {
                const bool in_p = in_p0 && k < g.sh[last], in_r = in_r0 && k < g.rs[last], in_c = in_c0 && k < g.oc[last];
                const unsigned kw = g.w == last ? k : kw0;
                const size_t roff = roff0 + (size_t)k * g.rstr[last], aoff = aoff0 + (size_t)k * g.astr[last];
                horner_linear_elem(res, rp, a, ap, out, op, g, lin, roff, aoff, kw, in_p, in_r, in_c, pos_consts, semi, fin);
            }
            for (int ax = last - 1; ax >= 0; --ax) {
                if (++idx[ax] < g.out.d[ax]) break;
                idx[ax] = 0;
            }
        }
    }
    // n consecutive positions of a line along the LAST axis that all have the terms T1 (res[k - 1] * m), T2 (c * res[k]), T3 (the
    // coefficient), in the finite regime — element for element horner_linear_elem's `fin` branch.  false: some operand or result
    // is outside the regime; nothing usable was stored (the caller redoes the run element by element).
    template <bool SEMI>
    static bool horner_run_sel(int sel, const double* res, size_t rp, const double* a, size_t ap, double* out, size_t op, const HornerArgs& g, size_t lin, size_t roff,
                               size_t aoff, int last, unsigned n, int flags) {
#define GFT_RUN(T1, T2, T3)                                                                                                  \
    return SEMI ? horner_run_semi<T1, T2, T3>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, flags)                   \
                : horner_run_fin<T1, T2, T3>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, flags)
        switch (sel) {
            case 0: GFT_RUN(false, false, false);
            case 1: GFT_RUN(true, false, false);
            case 2: GFT_RUN(false, true, false);
            case 3: GFT_RUN(true, true, false);
            case 4: GFT_RUN(false, false, true);
            case 5: GFT_RUN(true, false, true);
            case 6: GFT_RUN(false, true, true);
            default: GFT_RUN(true, true, true);
        }
#undef GFT_RUN
    }
    template <bool T1, bool T2, bool T3>
    static bool horner_run_fin(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap, double* __restrict out, size_t op,
                               const HornerArgs& g, size_t lin, size_t roff, size_t aoff, int last, unsigned n, int fin) {
        const size_t rs = g.rstr[last], as = g.coeff_scalar ? 0 : g.astr[last];  // (known strides: see horner_run_semi)
        if (rs == 1 && as <= 1 && host_avx512()) {  // (the same loop compiled for 8-wide / 4-wide vectors: same operations per element, same bits)
            if (as == 1) return horner_run_fin_avx512<T1, T2, T3, 1, 1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
            return horner_run_fin_avx512<T1, T2, T3, 1, 0>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
        }
        if (rs == 1 && as <= 1 && host_avx2()) {
            if (as == 1) return horner_run_fin_avx2<T1, T2, T3, 1, 1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
            return horner_run_fin_avx2<T1, T2, T3, 1, 0>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
        }
        if (rs == 1 && as == 1) return horner_run_fin_s<T1, T2, T3, 1, 1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
        if (rs == 1 && as == 0) return horner_run_fin_s<T1, T2, T3, 1, 0>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
        return horner_run_fin_s<T1, T2, T3, 0, 0>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
    }
    static bool host_avx2() {
        static const bool has = __builtin_cpu_supports("avx2");
        return has && g_host_simd != 0;
    }
    static bool host_avx512() {
        static const bool has = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
        return has && g_host_simd < 0;
    }
    template <bool T1, bool T2, bool T3, int RS1, int AS1>
    __attribute__((target("avx512f,avx512dq,avx512vl"))) static bool horner_run_fin_avx512(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap,
                                                                                           double* __restrict out, size_t op, const HornerArgs& g, size_t lin, size_t roff,
                                                                                           size_t aoff, int last, unsigned n, int fin) {
        return horner_run_fin_body<T1, T2, T3, RS1, AS1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
    }
    template <bool T1, bool T2, bool T3, int RS1, int AS1>
    __attribute__((target("avx512f,avx512dq,avx512vl"))) static bool horner_run_semi_avx512(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap,
                                                                                            double* __restrict out, size_t op, const HornerArgs& g, size_t lin, size_t roff,
                                                                                            size_t aoff, int last, unsigned n, int semi) {
        return horner_run_semi_body<T1, T2, T3, RS1, AS1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
    }
    template <bool T1, bool T2, bool T3, int RS1, int AS1>
    __attribute__((target("avx2"))) static bool horner_run_fin_avx2(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap, double* __restrict out,
                                                                    size_t op, const HornerArgs& g, size_t lin, size_t roff, size_t aoff, int last, unsigned n, int fin) {
        return horner_run_fin_body<T1, T2, T3, RS1, AS1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
    }
    template <bool T1, bool T2, bool T3, int RS1, int AS1>
    static bool horner_run_fin_s(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap, double* __restrict out, size_t op,
                                 const HornerArgs& g, size_t lin, size_t roff, size_t aoff, int last, unsigned n, int fin) {
        return horner_run_fin_body<T1, T2, T3, RS1, AS1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, fin);
    }
    template <bool T1, bool T2, bool T3, int RS1, int AS1>
    __attribute__((always_inline)) static inline bool horner_run_fin_body(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap, double* __restrict out, size_t op,
                                 const HornerArgs& g, size_t lin, size_t roff, size_t aoff, int last, unsigned n, int fin) {
        if constexpr (E::HAS_POS) {
            const V cv = E::from(g.c), mv = E::from(g.m);
            const size_t rs = RS1 ? (size_t)1 : g.rstr[last], as = RS1 ? (size_t)AS1 : (g.coeff_scalar ? (size_t)0 : g.astr[last]), wback = g.rstr[g.w];
            const bool m_one = (fin & 32) != 0;
            const double* const xl = res + roff, * const xh = res + rp + roff;
            const double* const ml = xl - wback, * const mh = xh - wback;
            const double* const al = a + (g.coeff_scalar ? g.a_base : aoff), * const ah = al + ap;
            double* const ol = out + lin, * const oh = out + op + lin;
            unsigned bad = 0;  // one verdict per run, no exit inside the loop (the caller redoes a failed run element by element)
            for (unsigned i = 0; i < n; ++i) {
                V p = E::zero();
                unsigned good = 1;
                if constexpr (T1) {
                    const V xm1 = Iv{ml[(size_t)i * rs], mh[(size_t)i * rs]};
                    good &= (unsigned)E::fin_ok(xm1);
                    p = m_one ? xm1 : E::mul_fin(xm1, mv);
                }
                if constexpr (T2) {
                    const V x = Iv{xl[(size_t)i * rs], xh[(size_t)i * rs]};
                    good &= (unsigned)E::fin_ok(x);
                    const V p2 = E::mul_fin(cv, x);
                    if constexpr (T1) p = E::widen_fin(p.lo + p2.lo, p.hi + p2.hi);
                    else p = p2;
                }
                V v = p;
                if constexpr (T3) {
                    const V cf = Iv{al[(size_t)i * as], ah[(size_t)i * as]};
                    good &= (unsigned)E::fin_ok(cf);
                    if constexpr (T1 || T2) v = E::widen_fin(p.lo + cf.lo, p.hi + cf.hi);
                    else v = cf;
                }
                good &= (unsigned)!E::is_nan(v);
                ol[i] = v.lo;
                oh[i] = v.hi;
                bad |= good ^ 1u;
            }
            return bad == 0;
        } else {
            return false;
        }
    }
    // ... and the same run in the `semi` regime (positive data, c with known signs): element for element horner_linear_elem's
    // `semi` branch — no flags, no exits inside the loop (one verdict per run: a run with an element outside the regime is redone
    // element by element by the caller, every regime tried), the planes read as planes.
    template <bool T1, bool T2, bool T3>
    static bool horner_run_semi(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap, double* __restrict out, size_t op,
                                const HornerArgs& g, size_t lin, size_t roff, size_t aoff, int last, unsigned n, int semi) {
        // (the accumulator is compact: unit stride along the line; the coefficient's stride is 1, or 0 where the line runs along
        // the substituted axis — known strides are what lets the compiler use vector loads)
        const size_t rs = g.rstr[last], as = g.coeff_scalar ? 0 : g.astr[last];
        if (rs == 1 && as <= 1 && host_avx512()) {
            if (as == 1) return horner_run_semi_avx512<T1, T2, T3, 1, 1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
            return horner_run_semi_avx512<T1, T2, T3, 1, 0>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
        }
        if (rs == 1 && as <= 1 && host_avx2()) {
            if (as == 1) return horner_run_semi_avx2<T1, T2, T3, 1, 1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
            return horner_run_semi_avx2<T1, T2, T3, 1, 0>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
        }
        if (rs == 1 && as == 1) return horner_run_semi_s<T1, T2, T3, 1, 1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
        if (rs == 1 && as == 0) return horner_run_semi_s<T1, T2, T3, 1, 0>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
        return horner_run_semi_s<T1, T2, T3, 0, 0>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
    }
    template <bool T1, bool T2, bool T3, int RS1, int AS1>
    __attribute__((target("avx2"))) static bool horner_run_semi_avx2(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap, double* __restrict out,
                                                                     size_t op, const HornerArgs& g, size_t lin, size_t roff, size_t aoff, int last, unsigned n, int semi) {
        return horner_run_semi_body<T1, T2, T3, RS1, AS1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
    }
    template <bool T1, bool T2, bool T3, int RS1, int AS1>
    static bool horner_run_semi_s(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap, double* __restrict out, size_t op,
                                  const HornerArgs& g, size_t lin, size_t roff, size_t aoff, int last, unsigned n, int semi) {
        return horner_run_semi_body<T1, T2, T3, RS1, AS1>(res, rp, a, ap, out, op, g, lin, roff, aoff, last, n, semi);
    }
    template <bool T1, bool T2, bool T3, int RS1, int AS1>  // RS1: the accumulator's stride along the line is 1 (else g's); AS1 likewise (with RS1: 1 or 0)
    __attribute__((always_inline)) static inline bool horner_run_semi_body(const double* __restrict res, size_t rp, const double* __restrict a, size_t ap, double* __restrict out, size_t op,
                                  const HornerArgs& g, size_t lin, size_t roff, size_t aoff, int last, unsigned n, int semi) {
        if constexpr (E::HAS_POS) {
            const V cv = E::from(g.c), mv = E::from(g.m);
            const size_t rs = RS1 ? (size_t)1 : g.rstr[last], as = RS1 ? (size_t)AS1 : (g.coeff_scalar ? (size_t)0 : g.astr[last]), wback = g.rstr[g.w];
            const bool m_one = (semi & 32) != 0;
            const double inf = bits_f64(0x7ff0000000000000LL);
            const long long dlo = (semi & 8) ? 1 : -1, dhi = (semi & 16) ? 1 : -1;
            const double* const xl = res + roff, * const xh = res + rp + roff;             // x = res[k]
            const double* const x_lo_src = (semi & 2) ? xh : xl, * const x_hi_src = (semi & 4) ? xl : xh;  // the bound of x each bound of c * x takes
            const double* const ml = xl - wback, * const mh = xh - wback;                   // res[k - 1 along w]
            const double* const al = a + (g.coeff_scalar ? g.a_base : aoff), * const ah = al + ap;
            double* const ol = out + lin, * const oh = out + op + lin;
            unsigned bad = 0;  // (ONE reduction, updated once per element: what the vectoriser recognises)
            auto pos = [inf](double lo, double hi) { return (unsigned)((lo > 0.0) & (lo <= hi) & (hi < inf) & !((lo == 1.0) & (hi == 1.0))); };
            for (unsigned i = 0; i < n; ++i) {
                double plo = 0.0, phi = 0.0;
                unsigned good = 1;
                if constexpr (T1) {
                    const double l = ml[(size_t)i * rs], h = mh[(size_t)i * rs];
                    good &= pos(l, h);
                    plo = m_one ? l : bits_f64(f64_bits(l * mv.lo) - 1);
                    phi = m_one ? h : bits_f64(f64_bits(h * mv.hi) + 1);
                    good &= (unsigned)((plo > 0.0) & (phi < inf));
                }
                if constexpr (T2) {
                    good &= pos(xl[(size_t)i * rs], xh[(size_t)i * rs]);
                    const double qlo = bits_f64(f64_bits(cv.lo * x_lo_src[(size_t)i * rs]) + dlo);
                    const double qhi = bits_f64(f64_bits(cv.hi * x_hi_src[(size_t)i * rs]) + dhi);
                    good &= (unsigned)((qlo > -inf) & (qhi < inf));
                    if constexpr (T1) {
                        plo = bits_f64(f64_bits(plo + qlo) - 1);
                        phi = bits_f64(f64_bits(phi + qhi) + 1);
                        good &= (unsigned)((plo > 0.0) & (phi < inf));
                    } else {
                        plo = qlo;
                        phi = qhi;
                    }
                }
                if constexpr (T3) {
                    const double l = al[(size_t)i * as], h = ah[(size_t)i * as];
                    good &= pos(l, h);
                    if constexpr (T1 || T2) {
                        plo = bits_f64(f64_bits(plo + l) - 1);
                        phi = bits_f64(f64_bits(phi + h) + 1);
                        good &= (unsigned)((plo > 0.0) & (phi < inf));
                    } else {
                        plo = l;
                        phi = h;
                    }
                }
                ol[i] = plo;
                oh[i] = phi;
                bad |= good ^ 1u;
            }
            return bad == 0;
        } else {
            return false;
        }
    }
    static inline void horner_linear_elem(const double* res, size_t rp, const double* a, size_t ap, double* out, size_t op, const HornerArgs& g,
                                          size_t lin, size_t roff, size_t aoff, unsigned kw, bool in_p, bool in_r, bool in_c, bool pos_consts,
                                          int semi, int fin) {
        const V cv = E::from(g.c), mv = E::from(g.m);
        const bool t1 = in_p && kw >= 1 && kw - 1 < g.upper;  // res[k - 1] * m exists
        const bool t2 = in_p && !g.c_zero && in_r;            // c * res[k] exists
        const bool t3 = g.coeff_scalar ? lin == 0 : in_c;     // the coefficient slab reaches this position
        if constexpr (E::HAS_POS) {
            if (semi) {
                const V xm1 = t1 ? E::ld(res, rp, roff - g.rstr[g.w]) : E::one(), x = t2 ? E::ld(res, rp, roff) : E::one();
                const V cf = t3 ? E::ld(a, ap, g.coeff_scalar ? g.a_base : aoff) : E::one();
                if ((!t1 || E::pos_ok(xm1)) && (!t2 || E::pos_ok(x)) && (!t3 || E::pos_ok(cf))) {
                    // every SUM below is formed with add_pos, which is the reference's sum iff its lower bound comes out
                    // positive (and its upper bound finite): `lo > 0` is false for the NaN patterns the integer steps make of
                    // a zero, a negative or an infinite sum, so one test per sum validates it
                    const V p1 = (semi & 32) ? xm1 : E::mul_pos(xm1, mv);
                    V p2;
                    p2.lo = bits_f64(f64_bits(cv.lo * ((semi & 2) ? x.hi : x.lo)) + ((semi & 8) ? 1 : -1));
                    p2.hi = bits_f64(f64_bits(cv.hi * ((semi & 4) ? x.lo : x.hi)) + ((semi & 16) ? 1 : -1));
                    const double inf = bits_f64(0x7ff0000000000000LL);
                    bool bad = (t1 && !(p1.lo > 0.0 && p1.hi < inf)) || (t2 && !(p2.lo > -inf && p2.hi < inf));  // (NaN fails every compare)
                    V p = E::zero();
                    if (t1 && t2) {
                        p = E::add_pos(p1, p2);
                        bad = bad || !(p.lo > 0.0 && p.hi < inf);
                    } else if (t1) {
                        p = p1;
                    } else if (t2) {
                        p = p2;
                    }
                    V v = p;
                    if (t3) {
                        if (t1 || t2) {
                            v = E::add_pos(p, cf);
                            bad = bad || !(v.lo > 0.0 && v.hi < inf);
                        } else {
                            v = cf;
                        }
                    }
                    if (!bad) {
                        E::st(out, op, lin, v);
                        g_host_horner_stats[10]++;
                        return;
                    }
                }
                g_host_horner_stats[3]++;
            }
            if (fin) {
                const V two = Iv{2.0, 3.0};  // (a finite, unremarkable stand-in for operands a position does not have)
                const V xm1 = t1 ? E::ld(res, rp, roff - g.rstr[g.w]) : two, x = t2 ? E::ld(res, rp, roff) : two;
                const V cf = t3 ? E::ld(a, ap, g.coeff_scalar ? g.a_base : aoff) : two;
                if (E::fin_ok(xm1) && E::fin_ok(x) && E::fin_ok(cf)) {
                    const V p1 = (fin & 32) ? xm1 : E::mul_fin(xm1, mv);
                    const V p2 = E::mul_fin(cv, x);
                    V p = E::zero();
                    if (t1 && t2) p = E::widen_fin(p1.lo + p2.lo, p1.hi + p2.hi);
                    else if (t1) p = p1;
                    else if (t2) p = p2;
                    V v = p;
                    if (t3) v = (t1 || t2) ? E::widen_fin(p.lo + cf.lo, p.hi + cf.hi) : cf;
                    if (!E::is_nan(v)) {
                        E::st(out, op, lin, v);
                        g_host_horner_stats[11]++;
                        return;
                    }
                }
            }
            if (pos_consts) {
                const V xm1 = t1 ? E::ld(res, rp, roff - g.rstr[g.w]) : E::one(), x = t2 ? E::ld(res, rp, roff) : E::one();
                const V cf = t3 ? E::ld(a, ap, g.coeff_scalar ? g.a_base : aoff) : E::one();
                if ((!t1 || E::pos_ok(xm1)) && (!t2 || E::pos_ok(x)) && (!t3 || E::pos_ok(cf))) {
                    const V p1 = E::mul_pos(xm1, mv), p2 = g.c_one ? x : E::mul_pos(cv, x);
                    bool bad = (t1 && !E::pos_first_ok(p1)) || (t2 && !g.c_one && !E::pos_first_ok(p2));
                    const V p = t1 ? (t2 ? E::add_pos(p1, p2) : p1) : (t2 ? p2 : E::zero());
                    const bool has_p = t1 || t2;
                    V v = t3 ? (has_p ? E::add_pos(p, cf) : cf) : p;
                    bad = bad || ((has_p || t3) && !E::pos_result_ok(v));
                    if (!bad) {
                        E::st(out, op, lin, v);
                        g_host_horner_stats[12]++;
                        return;
                    }
                }
            }
        }
        g_host_horner_stats[13]++;
        V p = E::zero();
        if (in_p) {
            if (t1) p = E::mul(E::ld(res, rp, roff - g.rstr[g.w]), mv);
            if (!g.c_zero) {
                p = E::add0(p);
                if (in_r) {
                    const V x = E::ld(res, rp, roff);
                    p = E::add(p, g.c_one ? x : E::mul(cv, x));
                }
            }
        }
        V v;
        if (g.coeff_scalar) {
            v = p;
            if (lin == 0) v = E::add(p, E::ld(a, ap, g.a_base));
        } else {
            v = E::zero();
            if (in_p) v = E::add0(p);
            if (in_c) v = E::add(v, E::ld(a, ap, aoff));
        }
        E::st(out, op, lin, v);
    }

    static size_t count_neq(const double* a, size_t ap, const double* b, size_t bp, size_t n) {
        size_t c = 0;
        for (size_t i = 0; i < n; ++i)
            if (!E::eq(E::ld(a, ap, i), E::ld(b, bp, i))) c++;
        return c;
    }

    // k_conv_naive: the reference's loop nest (mt:971-1012) per output element
    static void conv_axis(const ConvArgs& a, int ax, const unsigned* k, const double* x, size_t xp, const double* y, size_t yp,
                          size_t xoff, size_t yoff, V& acc) {
        const unsigned kk = k[ax];
        unsigned lo = (kk + 1 > a.ys[ax]) ? (kk + 1 - a.ys[ax]) : 0;
        unsigned hi = (kk + 1 < a.xs[ax]) ? (kk + 1) : a.xs[ax];
        bool desc = false;
        if (ax == 0) {
            if (lo < (unsigned)a.j0_min) lo = (unsigned)a.j0_min;
            if (a.j0_excl && hi > kk) hi = kk;
            desc = a.j0_desc != 0;
        }
        if (hi <= lo) return;
        const unsigned cnt = hi - lo;
        if (ax == a.nd - 1) {
            if (a.inner_from_zero) {
                V inner = E::zero();
                for (unsigned j = lo; j < hi; ++j)
                    inner = E::add(inner, E::mul(E::ld(x, xp, xoff + (size_t)j * a.xstr[ax]),
                                                 E::ld(y, yp, yoff + (size_t)(kk - j) * a.ystr[ax])));
                acc = E::add(acc, inner);
            } else {
                for (unsigned t = 0; t < cnt; ++t) {
                    const unsigned j = desc ? (hi - 1 - t) : (lo + t);
                    acc = E::add(acc, E::mul(E::ld(x, xp, xoff + (size_t)j * a.xstr[ax]),
                                             E::ld(y, yp, yoff + (size_t)(kk - j) * a.ystr[ax])));
                }
            }
        } else {
            for (unsigned t = 0; t < cnt; ++t) {
                const unsigned j = desc ? (hi - 1 - t) : (lo + t);
                conv_axis(a, ax + 1, k, x, xp, y, yp, xoff + (size_t)j * a.xstr[ax], yoff + (size_t)(kk - j) * a.ystr[ax], acc);
            }
        }
    }
    static void conv_naive(const double* x, size_t xp, const double* y, size_t yp, double* z, size_t zp, const ConvArgs& a) {
        size_t slab = 1;
        for (int i = 1; i < a.nd; ++i) slab *= a.zs[i];
        const size_t total = (a.nd == 0) ? 1 : (size_t)(a.slab_hi - a.slab_lo) * slab;
        for (size_t lin = 0; lin < total; ++lin) {
            const size_t zlin = lin + (size_t)a.slab_lo * slab;
            unsigned k[MAXD > 0 ? MAXD : 1];
            size_t r = zlin;
            for (int ax = a.nd - 1; ax >= 0; --ax) {
                const unsigned d = a.zs[ax];
                k[ax] = (unsigned)(r % d);
                r /= d;
            }
            V acc = a.accumulate ? E::ld(z, zp, zlin) : E::zero();
            if (a.nd == 0) acc = E::add(acc, E::mul(E::ld(x, xp, 0), E::ld(y, yp, 0)));
            else conv_axis(a, 0, k, x, xp, y, yp, 0, 0, acc);
            E::st(z, zp, zlin, acc);
        }
    }
};

}  // namespace gft
