// Blocked power-series division: the whole (last two axes) level of the division recurrence (mt:1162-1192) in ONE
// launch.  The reference divides slab by slab: for every leading index k0 it forms `cur = xs[k0] - sum_{j<k0} res[j] (*)
// ys[k0-j]` and then divides `cur` by ys[0] — a (d-1)-dimensional division that recurses the same way down to 1-d rows.
// Driving that recursion from the host costs one launch per row and level (4096 k_div_1d launches + 4 x 4160 helper
// launches for a 64^3 quotient, 0.08 % of the FP64 roof, profiles/r01/recurrences.txt).  Here one workgroup owns a 2-d
// slab: divisor and quotient rows live in LDS, the inner 1-d products of a row are formed by 16 waves in parallel and
// ADDED IN THE REFERENCE'S ORDER (mt:971-982: every row's partial sum from zero, sums added by ascending row), the 1-d
// division (mt:1162-1185 with 0-dim base) runs in lock step over j exactly like k_div_1d.  Same operations per element
// in the same order as the host-driven recursion => bit-identical quotients; only the launch count changes
// (64^3: 20 000 -> 128).
//
// The dividend of the slab is either a tensor (plain 2-d division) or, fused, `(-acc) + x` where `acc` is the slab of
// partial sums the leading-axis step just wrote into the quotient's own memory (the reference's neg + add + copy,
// mt:1186-1189) — the quotient row overwrites it in place.
#include <algorithm>
#include <cstdlib>

#include "gft_kernels.hpp"
#include <cstring>
#include <type_traits>
#include <map>
#include <vector>

namespace gft {

struct Div2dArgs {
    unsigned n1, n2;        // quotient slab shape (rows, row length)
    unsigned ny1, ny2;      // divisor shape (compact: <= n1, n2)
    unsigned nx1, nx2;      // box of the dividend tensor x (0, 0: none)
    size_t x_rstride;       // row stride of x
    int fused;              // 1: dividend row = (-res_in_place[k1][k2]) (+ x[k1][k2] inside x's box)
                            // 2: log's slab step (mt:1366-1384): dividend = (-res_in_place) (+ from(log_k) * x inside x's box), and
                            //    the slab stored is quotient / from(log_k), with (that) * from(log_k) stored to res2 as well
    unsigned log_k;
    double* res2;
    size_t r2p;
    int diag;               // timing diagnostics (GFT_DIV2D_DIAG, wrong results): 1 = no updater work, 2 = no division loop,
                            // 4 = no dividend prefetch
    unsigned n2p, ny2p;     // LDS row pitches
};

constexpr int D2_NW = 16;  // waves per workgroup

// what a finished quotient coefficient becomes in memory: itself, or — log's slab step — res[k] = q / k and rs[k] = res[k] * k
// (mt:1384 and mt:1362-1365; the same element operations MAP_DIV_U32 / MAP_MUL_U32 perform when the step is not fused)
template <class E>
__device__ inline void store_quotient(const Div2dArgs& g, double* res, size_t rp, size_t at, typename E::V q) {
    if (g.fused == 2) {
        const typename E::V kk = E::from_u32(g.log_k);
        const typename E::V r = E::div(q, kk);
        E::st(res, rp, at, r);
        E::st(g.res2, g.r2p, at, E::mul(r, kk));
    } else {
        E::st(res, rp, at, q);
    }
}

// Division by a divisor that stays the same for a whole slab (y[0, 0]): the f64 quotient n / d as the compiler expands
// it on gfx950 (v_div_scale, v_rcp, two Newton steps on the reciprocal, q = n*r, one residual step, v_div_fmas,
// v_div_fixup — a ~30-instruction dependent chain) spends most of its length on the reciprocal of d, which does not
// change.  With |d| and |n| in [2^-300, 2^300] the scale factors are 1 and the fix-up passes through, so the quotient is
// exactly  q = n*r; e = fma(-d, q, n); fma(e, r, q)  with the refined reciprocal r formed once: the same operations on
// the same values as the full expansion, hence the same (correctly rounded) bits, on a 3-instruction chain.  Outside
// the window the full division runs.  (The 1-d recurrence is one dependent chain of n2 divisions per row: this is the
// critical path of the whole quotient.)
struct FastDiv {
    double d, r;
    bool ok;
};
__device__ inline bool fd_mid(double v) {
    const unsigned e = (unsigned)((f64_bits(v) >> 52) & 0x7ff);
    return e > 1023 - 300 && e < 1023 + 300;
}
__device__ inline FastDiv fd_make(double d) {
    FastDiv f;
    f.d = d;
    f.ok = fd_mid(d);
    const double r0 = __builtin_amdgcn_rcp(d);
    const double r1 = __builtin_fma(r0, __builtin_fma(-d, r0, 1.0), r0);
    f.r = __builtin_fma(r1, __builtin_fma(-d, r1, 1.0), r1);
    return f;
}
__device__ inline double fd_div(double n, const FastDiv& f) {
    if (f.ok && fd_mid(n)) {
        const double q = n * f.r;
        return __builtin_fma(__builtin_fma(-f.d, q, n), f.r, q);
    }
    return n / f.d;
}
// value of lane j (wave-uniform j) in every lane
__device__ inline double bcast_f64(double x, unsigned j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), (int)j), hi = __builtin_amdgcn_readlane(__double2hiint(x), (int)j);
    return __hiloint2double(hi, lo);
}
template <class E>
__device__ inline typename E::V bcast_lane(typename E::V v, unsigned j);
template <>
__device__ inline double bcast_lane<EF64>(double v, unsigned j) { return bcast_f64(v, j); }
template <>
__device__ inline Iv bcast_lane<EIv>(Iv v, unsigned j) { return Iv{bcast_f64(v.lo, j), bcast_f64(v.hi, j)}; }

template <class E>
struct SlabDiv {  // generic element type: the functor's own division
    typename E::V d;
    static constexpr bool SPECULATIVE = false;
    __device__ explicit SlabDiv(typename E::V y) : d(y) {}
    __device__ typename E::V operator()(typename E::V n) const { return E::div(n, d); }
    __device__ typename E::V fast(typename E::V n) const { return E::div(n, d); }
    __device__ bool fast_ok(typename E::V) const { return true; }
};
template <>
struct SlabDiv<EF64> {
    FastDiv f;
    static constexpr bool SPECULATIVE = true;
    __device__ explicit SlabDiv(double y) : f(fd_make(y)) {}
    __device__ double operator()(double n) const { return fd_div(n, f); }
    // the three-instruction quotient WITHOUT the window test (the caller checks fast_ok(n) off the critical path and
    // redoes the row with operator() if some numerator was outside); an exact zero is fine: 0 * r = 0, both residual
    // steps keep it, and the sign is IEEE's
    __device__ double fast(double n) const {
        const double q = n * f.r;
        return __builtin_fma(__builtin_fma(-f.d, q, n), f.r, q);
    }
    __device__ bool fast_ok(double n) const { return f.ok && (n == 0.0 || fd_mid(n)); }
};

template <class E>
__global__ void __launch_bounds__(1024) k_div_2d(const double* __restrict__ x, size_t xp, const double* __restrict__ y, size_t yp,
                                                 double* res, size_t rp, Div2dArgs g) {
    typedef typename E::V V;
    extern __shared__ double d2_lds[];
    // layout: [y: W x ny1 x ny2p][r: W x n1 x n2p][part: W x NW x n2p][row: W x n2p]
    const size_t ysz = (size_t)g.ny1 * g.ny2p, rsz = (size_t)g.n1 * g.n2p, psz = (size_t)D2_NW * g.n2p;
    double* yl = d2_lds;
    double* rl = yl + E::W * ysz;
    double* pl = rl + E::W * rsz;
    double* tl = pl + E::W * psz;
    const unsigned tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
    for (size_t i = tid; i < (size_t)g.ny1 * g.ny2; i += nthr) {
        const unsigned a = (unsigned)(i / g.ny2), b = (unsigned)(i - (size_t)a * g.ny2);
        E::st(yl, ysz, (size_t)a * g.ny2p + b, E::ld(y, yp, i));
    }
    __syncthreads();
    const SlabDiv<E> div_y00(E::ld(yl, ysz, 0));
    const bool owner = tid < g.n2;
    const unsigned k2 = tid;
    const unsigned lo2 = (k2 + 1 > g.ny2) ? (k2 + 1 - g.ny2) : 0;  // 1-d level: first j whose y[k2 - j] exists
    for (unsigned k1 = 0; k1 < g.n1; ++k1) {
        // ---- sum_{j1 < k1} (res[j1, :] (*) y[k1 - j1, :])[k2]: each term's row product from zero, terms added by ascending j1
        const unsigned lo1 = (k1 + 1 > g.ny1) ? (k1 + 1 - g.ny1) : 0;
        V acc = E::zero();
        for (unsigned base = lo1; base < k1; base += nwaves) {
            const unsigned j1 = base + wave;
            if (j1 < k1) {
                const size_t rrow = (size_t)j1 * g.n2p, yrow = (size_t)(k1 - j1) * g.ny2p;
                for (unsigned c = lane; c < g.n2; c += 64) {
                    const unsigned jl = (c + 1 > g.ny2) ? (c + 1 - g.ny2) : 0;
                    V inner = E::zero();
#pragma unroll 4
                    for (unsigned j2 = jl; j2 <= c; ++j2)
                        inner = E::add(inner, E::mul(E::ld(rl, rsz, rrow + j2), E::ld(yl, ysz, yrow + (c - j2))));
                    E::st(pl, psz, (size_t)wave * g.n2p + c, inner);
                }
            }
            __syncthreads();
            if (owner) {
                const unsigned cnt = (k1 - base < nwaves) ? (k1 - base) : nwaves;
                for (unsigned w = 0; w < cnt; ++w) acc = E::add(acc, E::ld(pl, psz, (size_t)w * g.n2p + k2));
            }
            __syncthreads();
        }
        // ---- dividend row: cur = -acc; cur += x[k1]  (mt:1186-1188 at this level), x itself possibly the fused outer step
        V cur1 = E::zero();  // the 1-d level's running sum of res[j] * y[0][k2 - j]
        V t = E::zero();
        if (owner) {
            t = E::neg(acc);
            if (g.fused) {
                V xv = E::neg(E::ld(res, rp, (size_t)k1 * g.n2 + k2));
                if (k1 < g.nx1 && k2 < g.nx2) {
                    const V xin = E::ld(x, xp, (size_t)k1 * g.x_rstride + k2);
                    xv = E::add(xv, g.fused == 2 ? E::mul(E::from_u32(g.log_k), xin) : xin);
                }
                t = E::add(t, xv);
            } else if (k1 < g.nx1 && k2 < g.nx2) {
                t = E::add(t, E::ld(x, xp, (size_t)k1 * g.x_rstride + k2));
            }
        }
        // ---- 1-d division of the row by y[0, :], lock step over j (k_div_1d): lane j finalises res[j], then every lane
        // k2 > j adds res[j] * y[0][k2 - j].  Rows of at most 64 coefficients live in ONE wave: res[j] travels by
        // v_readlane (no LDS round trip, no barrier) and the finished row is written out once, coalesced.
        V mine = E::zero();
        if (g.n2 <= 64) {
            if (wave == 0) {
                for (unsigned j = 0; j < g.n2; ++j) {
                    const V r = div_y00(bcast_lane<E>(E::add(E::neg(cur1), t), j));
                    if (k2 == j) mine = r;
                    if (owner && k2 > j && j >= lo2) cur1 = E::add(cur1, E::mul(r, E::ld(yl, ysz, k2 - j)));
                }
            }
        } else {
            for (unsigned j = 0; j < g.n2; ++j) {
                if (owner && k2 == j) {
                    mine = div_y00(E::add(E::neg(cur1), t));
                    E::st(tl, g.n2p, j, mine);
                }
                __syncthreads();
                if (owner && k2 > j && j >= lo2) cur1 = E::add(cur1, E::mul(E::ld(tl, g.n2p, j), E::ld(yl, ysz, k2 - j)));
            }
        }
        if (owner) {
            E::st(rl, rsz, (size_t)k1 * g.n2p + k2, mine);
            store_quotient<E>(g, res, rp, (size_t)k1 * g.n2 + k2, mine);
        }
        __syncthreads();  // row k1 of the quotient is in LDS for the next rows
    }
}

// ---- rows of at most 64 coefficients: right-looking, operands in registers ---------------------------------------------
// The kernel above keeps every operand of a row product in LDS (two 8-byte reads per multiply-add: LDS-bound, and a
// 64 x 64 slab took 1.6 ms).  Here a row lives in ONE wave, one coefficient per lane:
//   * x operand (a finished quotient row): lane j's value reaches all lanes as an SGPR pair (v_readlane);
//   * y operand (a divisor row): the lanes hold y[c], and one DPP wave shift per step turns that into y[c - j2]
//     (zeros enter at lane 0, which is exactly the truncation);
// so a row product  inner[c] = sum_{j2 <= c} r[j2] * y[c - j2]  is 64 steps of {2 readlane, mul, add, 2 dpp-mov} with no
// memory traffic at all.  The accumulation is RIGHT-LOOKING: when row k1 is final, every wave adds the term
// res[k1] (*) y[k1' - k1] to the accumulator rows k1' it owns (k1' = wave mod NW); a row is owned by one wave, which
// applies its terms in ascending k1 — the reference's order (mt:971-982), each term's product formed from zero — and
// then divides the row in lock step (mt:1162-1185) with the same shift trick for y[0, :].  The owner of row k1 + 1
// applies the last missing term first, so the division (the critical path) overlaps the other waves' updates.
__device__ inline double wave_shr1_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);  // wave_shr:1, zeros shifted in
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <class E>
__device__ inline bool elem_finite(typename E::V v);
template <>
__device__ inline bool elem_finite<EF64>(double v) { return finite_d(v); }
template <>
__device__ inline bool elem_finite<EIv>(Iv v) { return EIv::is_finite(v); }
__device__ inline double row_shr1_f64(double x) {  // lane l of every 16-lane row takes lane l-1's value, lane 0 takes 0
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);  // row_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <class E>
__device__ inline typename E::V row_shr1(typename E::V v);
template <>
__device__ inline double row_shr1<EF64>(double v) { return row_shr1_f64(v); }
template <>
__device__ inline Iv row_shr1<EIv>(Iv v) { return Iv{row_shr1_f64(v.lo), row_shr1_f64(v.hi)}; }
template <class E>
__device__ inline typename E::V wave_shr1(typename E::V v);
template <>
__device__ inline double wave_shr1<EF64>(double v) { return wave_shr1_f64(v); }
template <>
__device__ inline Iv wave_shr1<EIv>(Iv v) { return Iv{wave_shr1_f64(v.lo), wave_shr1_f64(v.hi)}; }

template <class E>
__global__ void __launch_bounds__(1024) k_div_2d_rows64(const double* __restrict__ x, size_t xp, const double* __restrict__ y, size_t yp,
                                                        double* res, size_t rp, Div2dArgs g) {
    typedef typename E::V V;
    extern __shared__ double d2_lds[];
    // layout: [y: W x ny1 x ny2p][acc: W x n1 x n2p][final rows: W x 2 x 64]
    const size_t ysz = (size_t)g.ny1 * g.ny2p, asz = (size_t)g.n1 * g.n2p;
    double* yl = d2_lds;
    double* al = yl + E::W * ysz;
    double* fl = al + E::W * asz;
    const unsigned tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, c = tid & 63, nw = nthr >> 6;
    for (size_t i = tid; i < (size_t)g.ny1 * g.ny2; i += nthr) {
        const unsigned a = (unsigned)(i / g.ny2), b = (unsigned)(i - (size_t)a * g.ny2);
        E::st(yl, ysz, (size_t)a * g.ny2p + b, E::ld(y, yp, i));
    }
    for (size_t i = tid; i < asz; i += nthr) E::st(al, asz, i, E::zero());
    __syncthreads();
    const bool col = c < g.n2;
    const V y0v = (c < g.ny2) ? E::ld(yl, ysz, c) : E::zero();  // y[0, c]
    const SlabDiv<E> div_y00(E::ld(yl, ysz, 0));
    // dividend of the row this wave finalises next (prefetched: the global load is off the critical path)
    auto dividend = [&](unsigned k1) -> V {
        V d = E::zero();
        if (!col || k1 >= g.n1) return d;
        if (g.fused) {
            d = E::neg(E::ld(res, rp, (size_t)k1 * g.n2 + c));
            if (k1 < g.nx1 && c < g.nx2) {
                const V xin = E::ld(x, xp, (size_t)k1 * g.x_rstride + c);
                d = E::add(d, g.fused == 2 ? E::mul(E::from_u32(g.log_k), xin) : xin);
            }
        } else if (k1 < g.nx1 && c < g.nx2) {
            d = E::ld(x, xp, (size_t)k1 * g.x_rstride + c);
        }
        return d;
    };
    // UPDATER layout: a wave is four 16-lane DPP rows, each row of lanes serves ONE accumulator row r, and lane l of it
    // owns the four coefficients c = 4l .. 4l+3 — a register-tiled sliding window: at step j2 the lane needs
    // y[d, c - j2] for its four c; stepping j2 shifts that window by one, three of the four values stay in the lane (a
    // register rotation, free under 4x unrolling) and one arrives from the left neighbour by a DPP row_shr:1 — the
    // cheap intra-row kind (a wave-wide shift costs ~25 cycles of issue per 64-bit value, tools/microbench_rowconv.hip).
    // The x coefficient of the step is one v_readlane shared by all 256 multiply-adds of the wave.  Per output still
    // ascending j2 from zero, so still the reference's bits; four accumulator rows per wave and step, 60 per pass of
    // the 15 updaters.
    const unsigned nup = nw > 1 ? nw - 1 : 1;
    // index among the updaters: the waves of SIMDs 1..3 first, round-robin over the SIMDs (wave w runs on SIMD w % 4), the
    // three that share SIMD 0 with the divider last — when a step has fewer rows than slots, the waves left idle are the
    // divider's neighbours first, and the busy ones are spread evenly over the other SIMDs
    const unsigned n_other = (nw - 1) - (nw - 1) / 4;
    const unsigned uidx = nw <= 1 ? 0 : (wave & 3u) ? (wave >> 2) * 3 + (wave & 3u) - 1 : n_other + (wave >> 2) - 1;
    const unsigned drow = c >> 4, dl = c & 15u;
    auto apply_terms4 = [&](unsigned r_first, unsigned n_rows, unsigned j1) {
        // this DPP row's accumulator row (rows r_first + 4 * uidx + drow, then + 4 * nup per pass)
        for (unsigned base = r_first + 4 * uidx; base < r_first + n_rows; base += 4 * nup) {
            const unsigned r = base + drow;
            const unsigned d = r - j1;
            const bool act = r < r_first + n_rows && d < g.ny1;  // no such y row: the reference's bound lo1 excludes the term
            V xl[4], w[4], inner[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned cc = 4 * dl + e;
                xl[e] = cc < g.n2 ? E::ld(fl, 128, (size_t)(j1 & 1) * 64 + cc) : E::zero();
                w[e] = (act && cc < g.ny2) ? E::ld(yl, ysz, (size_t)d * g.ny2p + cc) : E::zero();
                inner[e] = E::zero();
            }
            bool fin = true;
#pragma unroll
            for (int e = 0; e < 4; ++e) fin = fin && elem_finite<E>(xl[e]);
            const unsigned steps = (g.n2 + 3) / 4;  // groups of four j2
            if (!any_lane(!fin)) {
                // finite x row: excluded positions multiply an exact zero (shifted in at lane 0 of the row / beyond y's
                // row), inner + x * 0 == inner — no masks (see the divider's loop)
                for (unsigned jg = 0; jg < steps; ++jg) {
                    const V x0 = bcast_lane<E>(xl[0], jg), x1 = bcast_lane<E>(xl[1], jg), x2 = bcast_lane<E>(xl[2], jg), x3 = bcast_lane<E>(xl[3], jg);
                    const V t1 = row_shr1<E>(w[3]), t2 = row_shr1<E>(w[2]), t3 = row_shr1<E>(w[1]), t4 = row_shr1<E>(w[0]);
                    // j2 = 4 jg: window (w0, w1, w2, w3)
                    inner[0] = E::add(inner[0], E::mul(x0, w[0])); inner[1] = E::add(inner[1], E::mul(x0, w[1]));
                    inner[2] = E::add(inner[2], E::mul(x0, w[2])); inner[3] = E::add(inner[3], E::mul(x0, w[3]));
                    // j2 + 1: (t1, w0, w1, w2)
                    inner[0] = E::add(inner[0], E::mul(x1, t1)); inner[1] = E::add(inner[1], E::mul(x1, w[0]));
                    inner[2] = E::add(inner[2], E::mul(x1, w[1])); inner[3] = E::add(inner[3], E::mul(x1, w[2]));
                    // j2 + 2: (t2, t1, w0, w1)
                    inner[0] = E::add(inner[0], E::mul(x2, t2)); inner[1] = E::add(inner[1], E::mul(x2, t1));
                    inner[2] = E::add(inner[2], E::mul(x2, w[0])); inner[3] = E::add(inner[3], E::mul(x2, w[1]));
                    // j2 + 3: (t3, t2, t1, w0)
                    inner[0] = E::add(inner[0], E::mul(x3, t3)); inner[1] = E::add(inner[1], E::mul(x3, t2));
                    inner[2] = E::add(inner[2], E::mul(x3, t1)); inner[3] = E::add(inner[3], E::mul(x3, w[0]));
                    w[0] = t4; w[1] = t3; w[2] = t2; w[3] = t1;
                }
            } else {
                // a non-finite coefficient in the x row: the reference's bounds as selects (c >= j2, c - j2 < ny2)
                for (unsigned jg = 0; jg < steps; ++jg) {
                    const V xq[4] = {bcast_lane<E>(xl[0], jg), bcast_lane<E>(xl[1], jg), bcast_lane<E>(xl[2], jg), bcast_lane<E>(xl[3], jg)};
                    const V t1 = row_shr1<E>(w[3]), t2 = row_shr1<E>(w[2]), t3 = row_shr1<E>(w[1]), t4 = row_shr1<E>(w[0]);
                    const V win[4][4] = {{w[0], w[1], w[2], w[3]}, {t1, w[0], w[1], w[2]}, {t2, t1, w[0], w[1]}, {t3, t2, t1, w[0]}};
#pragma unroll
                    for (int sgm = 0; sgm < 4; ++sgm) {
                        const unsigned j2 = 4 * jg + sgm;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned cc = 4 * dl + e;
                            if (j2 < g.n2 && cc < g.n2 && cc >= j2 && cc - j2 < g.ny2) inner[e] = E::add(inner[e], E::mul(xq[sgm], win[sgm][e]));
                        }
                    }
                    w[0] = t4; w[1] = t3; w[2] = t2; w[3] = t1;
                }
            }
            if (act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned cc = 4 * dl + e;
                    if (cc < g.n2) {
                        const size_t at = (size_t)r * g.n2p + cc;
                        E::st(al, asz, at, E::add(E::ld(al, asz, at), inner[e]));
                    }
                }
            }
        }
    };
    // Wave 0 is the DIVIDER: it owns the critical path — the lock-step division of every row — and, while it divides row
    // k1, it already forms the term res[k1] (*) y[1] of row k1 + 1 from the quotient coefficients as they appear (the same
    // broadcast serves both), so row k1 + 1's dividend is complete the moment row k1 is.  The other waves are UPDATERS:
    // during step k1 they add the term res[k1 - 1] (*) y[r - k1 + 1] to the rows r >= k1 + 1 they own.  Every row
    // receives its terms in ascending order (the updaters' by construction, the divider's last), each term's product
    // formed from zero: the reference's operations in the reference's order.
    const bool divider = wave == 0, updater = nw == 1 || wave >= 1;
    const bool next_term = g.ny1 >= 2;
    const V y1v = (next_term && c < g.ny2) ? E::ld(yl, ysz, (size_t)g.ny2p + c) : E::zero();  // y[1, c]
    V next_dividend = divider ? dividend(0) : E::zero();
    V inner_next = E::zero();  // divider: res[k1 - 1] (*) y[1], the last term of row k1
    for (unsigned k1 = 0; k1 < g.n1; ++k1) {
        if (divider) {
            // dividend row: acc += last term; cur = -acc; cur += x[k1]  (mt:1186-1188 at this level)
            V t = E::zero();
            if (col) {
                V a0 = E::ld(al, asz, (size_t)k1 * g.n2p + c);
                if (k1 > 0 && next_term) a0 = E::add(a0, inner_next);
                t = E::neg(a0);
            }
            const bool has_div = g.fused || (k1 < g.nx1);
            if (col && has_div && (g.fused || c < g.nx2)) t = E::add(t, next_dividend);
            next_dividend = (g.diag & 4) ? E::zero() : dividend(k1 + 1);  // in flight during the division below
            // 1-d division by y[0, :], lock step over j, and the next row's last term on the same broadcast
            V cur1 = E::zero(), ys = y0v, y1s = y1v, mine = E::zero();
            inner_next = E::zero();
            // ys = y[0, c - j], y1s = y[1, c - j] after j shifts.  Branch-free body (every branch in this loop is paid 4096
            // times a slab on the critical path): lanes c <= j are past their own step (their sum no longer matters), so
            // cur1's lower bound needs no mask; the other positions the reference's bounds exclude multiply by a shifted-in
            // zero, which only a NON-FINITE quotient coefficient could turn into something — the loop notes whether one
            // appeared and the row is then redone with the bounds as selects.
            // The quotient of the step: speculatively the 3-instruction form for every step — whether lane j's numerator
            // was inside the exponent window is noted per lane (vector code beside the chain, not in it) and settled by
            // one ballot after the row; a numerator in the window and a divisor in the window give a finite quotient,
            // so the same verdict covers the masks dropped above.
            bool lane_bad = false;
            if constexpr (SlabDiv<E>::SPECULATIVE) {
                // as few instructions as the step allows (one wave issues roughly one instruction per 8 cycles on this
                // chain, so the instruction COUNT is the row's latency): numerator, its copy for the check after the
                // row (same compare as `mine`), broadcast, 3-instruction quotient, two multiply-adds, two shifts
                V mynum = E::zero();
                for (unsigned j = 0; j < ((g.diag & 2) ? 1u : g.n2); ++j) {
                    const V num = E::add(E::neg(cur1), t);
                    const bool is_j = c == j;
                    mynum = is_j ? num : mynum;
                    const V q = div_y00.fast(bcast_lane<E>(num, j));
                    mine = is_j ? q : mine;
                    cur1 = E::add(cur1, E::mul(q, ys));
                    inner_next = E::add(inner_next, E::mul(q, y1s));
                    ys = wave_shr1<E>(ys);  // (reading the shifted rows from LDS instead costs the same: measured)
                    y1s = wave_shr1<E>(y1s);
                }
                lane_bad = col && !div_y00.fast_ok(mynum);
            } else {
                for (unsigned j = 0; j < ((g.diag & 2) ? 1u : g.n2); ++j) {
                    const V num = E::add(E::neg(cur1), t);
                    if (c == j && !elem_finite<E>(num)) lane_bad = true;
                    const V q = div_y00.fast(bcast_lane<E>(num, j));
                    if (c == j) mine = q;
                    cur1 = E::add(cur1, E::mul(q, ys));
                    inner_next = E::add(inner_next, E::mul(q, y1s));
                    ys = wave_shr1<E>(ys);
                    y1s = wave_shr1<E>(y1s);
                }
            }
            bool all_finite = !any_lane(lane_bad);
            if (!SlabDiv<E>::SPECULATIVE && all_finite) all_finite = !any_lane(col && !elem_finite<E>(mine));
            if (!all_finite) {
                cur1 = E::zero(), ys = y0v, y1s = y1v, mine = E::zero();
                inner_next = E::zero();
                for (unsigned j = 0; j < g.n2; ++j) {
                    const V q = div_y00(bcast_lane<E>(E::add(E::neg(cur1), t), j));
                    if (c == j) mine = q;
                    if (col && c > j && c - j < g.ny2) cur1 = E::add(cur1, E::mul(q, ys));
                    if (col && c >= j && c - j < g.ny2) inner_next = E::add(inner_next, E::mul(q, y1s));
                    ys = wave_shr1<E>(ys);
                    y1s = wave_shr1<E>(y1s);
                }
            }
            if (col) {
                E::st(fl, 128, (size_t)(k1 & 1) * 64 + c, mine);
                store_quotient<E>(g, res, rp, (size_t)k1 * g.n2 + c, mine);
            }
        }
        if (updater && k1 > 0 && k1 + 1 < g.n1 && !(g.diag & 1))
            apply_terms4(k1 + 1, g.n1 - (k1 + 1), k1 - 1);  // term j1 = k1 - 1 into the rows r >= k1 + 1
        __syncthreads();  // row k1 is final (fl); every accumulator row >= k1 + 1 holds the terms up to k1 - 1
    }
}

// ---- 1-d division (mt:1162-1185 base case) -----------------------------------------------------------------------------
// res[k] = (x[k] - sum_{j<k} res[j] * y[k-j]) / y[0], the sum accumulated for j ascending.  Serial in k, but lane k's
// chain only needs res[j] at its j-th step and res[j] is final once lane j has done its steps 0..j-1 — so all lanes
// advance in lock step over j: step j = (the owner of j finalises and publishes res[j]) then (every lane k > j adds
// res[j] * y[k-j]).  Depth n instead of n^2/2, per-lane operation order unchanged.  A step is an LDS round trip and a
// handful of instructions, so nothing slow may sit in it: the divisor row lives in LDS (was: a global load per step and
// lane), every lane's dividend is loaded up front, the quotient uses the slab division's refined reciprocal (3
// instructions inside the exponent window instead of the ~30 of a full f64 division), the barrier orders LDS only
// (the result's global store stays in flight), and the fused form takes the dividend as (-res) (+ x) in place —
// the three element-wise launches the row recursion of div_rec spent per row.
constexpr int DIV1D_EPT = 4;  // outputs per thread: n <= 4096 in one workgroup
template <class E>
__global__ void __launch_bounds__(1024) k_div_1d(const double* __restrict__ xs, size_t xp, unsigned nx,
                                                 const double* __restrict__ ys, size_t yp, unsigned ny, double* res, size_t rp,
                                                 unsigned n, int fused) {
    typedef typename E::V V;
    extern __shared__ double d1_lds[];  // [plane][n] quotient mirror, [plane][ny] divisor row
    double* rl = d1_lds;
    double* yl = d1_lds + (size_t)E::W * n;
    const unsigned nys = ny < n ? ny : n;  // y[k - j] with k - j < n
    for (unsigned i = threadIdx.x; i < nys; i += blockDim.x) E::st(yl, nys, i, E::ld(ys, yp, i));
    V cur[DIV1D_EPT], dvd[DIV1D_EPT];
    unsigned kk[DIV1D_EPT], lo[DIV1D_EPT];
#pragma unroll
    for (int e = 0; e < DIV1D_EPT; ++e) {
        kk[e] = threadIdx.x + e * blockDim.x;
        lo[e] = (kk[e] + 1 > ny) ? (kk[e] + 1 - ny) : 0;
        cur[e] = E::zero();
        dvd[e] = E::zero();
        if (kk[e] < n) {
            if (fused) {  // dividend = (-acc) (+ x): mt:1186-1188 at the level above
                dvd[e] = E::neg(E::ld(res, rp, kk[e]));
                if (kk[e] < nx) dvd[e] = E::add(dvd[e], E::ld(xs, xp, kk[e]));
            } else if (kk[e] < nx) {
                dvd[e] = E::ld(xs, xp, kk[e]);
            }
        }
    }
    const SlabDiv<E> div_y0(E::ld(ys, yp, 0));
    __syncthreads();
    for (unsigned j = 0; j < n; ++j) {
        // the owner of output j finalises it
        const unsigned oe = j / blockDim.x;
        if (threadIdx.x == j - oe * blockDim.x) {
#pragma unroll
            for (int e = 0; e < DIV1D_EPT; ++e) {
                if (e == (int)oe) {
                    V c = E::neg(cur[e]);
                    if (fused || j < nx) c = E::add(c, dvd[e]);
                    const V r = div_y0(c);
                    E::st(res, rp, j, r);
                    E::st(rl, n, j, r);
                }
            }
        }
        lds_barrier();
        const V rj = E::ld(rl, n, j);
#pragma unroll
        for (int e = 0; e < DIV1D_EPT; ++e) {
            const unsigned k = kk[e];
            if (k < n && k > j && j >= lo[e]) cur[e] = E::add(cur[e], E::mul(rj, E::ld(yl, nys, k - j)));
        }
    }
}
// Rows of at most 1024 coefficients: ONE wave, DIV1D_WSEG = 4, 8 or 16 coefficients per lane (k = 64 e + lane).  The quotient coefficient of
// the step reaches all lanes by v_readlane instead of an LDS round trip plus a workgroup barrier — 0.06 us a step on
// the slab kernel's divider against 0.44 us here before (a lone workgroup runs at the idle clock; every instruction and
// every barrier of the chain is paid in full) — and the divisor row comes from LDS at a per-lane sliding address.
template <class E, int DIV1D_WSEG>
__global__ void __launch_bounds__(64) k_div_1d_wave(const double* __restrict__ xs, size_t xp, unsigned nx,
                                                    const double* __restrict__ ys, size_t yp, unsigned ny, double* res, size_t rp,
                                                    unsigned n, int fused) {
    typedef typename E::V V;
    extern __shared__ double d1_lds[];  // [plane][64 zeros | 64 * DIV1D_WSEG of the divisor row, zero beyond ny]
    constexpr unsigned CAP = 64 * (DIV1D_WSEG + 1);
    const unsigned lane = threadIdx.x;
    E::st(d1_lds, CAP, lane, E::zero());
#pragma unroll
    for (int e = 0; e < DIV1D_WSEG; ++e) {
        const unsigned i = 64 * e + lane;
        E::st(d1_lds, CAP, 64 + i, i < ny ? E::ld(ys, yp, i) : E::zero());
    }
    V cur[DIV1D_WSEG], dvd[DIV1D_WSEG], mine[DIV1D_WSEG];
    unsigned lo[DIV1D_WSEG];
#pragma unroll
    for (int e = 0; e < DIV1D_WSEG; ++e) {
        const unsigned k = 64 * e + lane;
        lo[e] = (k + 1 > ny) ? (k + 1 - ny) : 0;
        cur[e] = mine[e] = dvd[e] = E::zero();
        if (k < n) {
            if (fused) {
                dvd[e] = E::neg(E::ld(res, rp, k));
                if (k < nx) dvd[e] = E::add(dvd[e], E::ld(xs, xp, k));
            } else if (k < nx) {
                dvd[e] = E::ld(xs, xp, k);
            }
        }
    }
    const SlabDiv<E> div_y0(E::ld(ys, yp, 0));
    __syncthreads();
    // Fast pass: no per-lane bounds.  A lane past its own step keeps accumulating a sum nobody reads; a position beyond
    // the divisor's length multiplies a staged zero, and cur + q * 0 == cur for every FINITE q (cur is never -0: it
    // starts at +0 and only adds) — so the pass is exact unless a non-finite quotient coefficient appears, which is
    // noted and settled by redoing the row with the bounds as selects (the reference's own case analysis).
    // The divisor values of step j + 1 are requested before the quotient of step j is formed: no LDS wait in the chain.
    bool bad = false;
#pragma unroll
    for (int oe = 0; oe < DIV1D_WSEG; ++oe) {
        const unsigned j0 = 64u * oe, j1 = n < j0 + 64 ? n : j0 + 64;
        if (j0 >= j1) break;
        V yv[DIV1D_WSEG];
#pragma unroll
        for (int e = oe; e < DIV1D_WSEG; ++e) yv[e] = E::ld(d1_lds, CAP, 64 + 64 * e + lane - j0);
        for (unsigned j = j0; j < j1; ++j) {
            const unsigned ol = j - j0;
            V num = E::neg(cur[oe]);
            if (fused || j < nx) num = E::add(num, dvd[oe]);
            V yn[DIV1D_WSEG];
#pragma unroll
            for (int e = oe; e < DIV1D_WSEG; ++e) yn[e] = E::ld(d1_lds, CAP, 64 + 64 * e + lane - (j + 1));
            const V q = div_y0(bcast_lane<E>(num, ol));
            bad = bad || !elem_finite<E>(q);
            if (lane == ol) mine[oe] = q;
#pragma unroll
            for (int e = oe; e < DIV1D_WSEG; ++e) {
                cur[e] = E::add(cur[e], E::mul(q, yv[e]));
                yv[e] = yn[e];
            }
        }
    }
    if (bad) {  // (wave-uniform: every lane saw the same quotient coefficients)
#pragma unroll
        for (int e = 0; e < DIV1D_WSEG; ++e) cur[e] = E::zero();
        for (unsigned j = 0; j < n; ++j) {
            const unsigned oe = j >> 6, ol = j & 63u;
            V num = E::zero();
#pragma unroll
            for (int e = 0; e < DIV1D_WSEG; ++e)
                if ((unsigned)e == oe) {
                    num = E::neg(cur[e]);
                    if (fused || j < nx) num = E::add(num, dvd[e]);
                }
            const V q = div_y0(bcast_lane<E>(num, ol));
#pragma unroll
            for (int e = 0; e < DIV1D_WSEG; ++e) {
                const unsigned k = 64 * e + lane;
                if ((unsigned)e == oe && lane == ol) mine[e] = q;
                if (k < n && k > j && j >= lo[e]) cur[e] = E::add(cur[e], E::mul(q, E::ld(d1_lds, CAP, 64 + k - j)));
            }
        }
    }
#pragma unroll
    for (int e = 0; e < DIV1D_WSEG; ++e) {
        const unsigned k = 64 * e + lane;
        if (k < n) E::st(res, rp, k, mine[e]);
    }
}
// serial fallback for n > 4096
template <class E>
__global__ void k_div_1d_serial(const double* xs, size_t xp, unsigned nx, const double* ys, size_t yp, unsigned ny,
                                double* res, size_t rp, unsigned n) {
    typedef typename E::V V;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    V y0 = E::ld(ys, yp, 0);
    for (unsigned k = 0; k < n; ++k) {
        V cur = E::zero();
        unsigned lo = (k + 1 > ny) ? (k + 1 - ny) : 0;
        for (unsigned j = lo; j < k; ++j) cur = E::add(cur, E::mul(E::ld(res, rp, j), E::ld(ys, yp, k - j)));
        cur = E::neg(cur);
        if (k < nx) cur = E::add(cur, E::ld(xs, xp, k));
        E::st(res, rp, k, E::div(cur, y0));
    }
}
template <class E>
bool K<E>::div_1d(hipStream_t st, const double* xs, size_t x_plane, unsigned nx, const double* ys, size_t y_plane,
                  unsigned ny, double* res, size_t r_plane, unsigned n, int fused) {
    if (n == 0) return true;
    if (n > 1024 * DIV1D_EPT) {
        if (fused) return false;  // the caller prepares the dividend itself
        GFT_LAUNCH(k_div_1d_serial<E>, dim3(1), dim3(64), 0, st, xs, x_plane, nx, ys, y_plane, ny, res, r_plane, n);
        return true;
    }
    static const bool wave_on = true;
    if (wave_on && n <= 1024) {
#define GFT_D1W(SEG)                                                                                                        \
    GFT_LAUNCH((k_div_1d_wave<E, SEG>), dim3(1), dim3(64), (size_t)E::W * 64 * (SEG + 1) * sizeof(double), st, xs, x_plane, \
                       nx, ys, y_plane, ny, res, r_plane, n, fused)
        if (n <= 256) GFT_D1W(4);
        else if (n <= 512) GFT_D1W(8);
        else GFT_D1W(16);
#undef GFT_D1W
        return true;
    }
    unsigned threads = std::min<unsigned>(1024, (n + 63) / 64 * 64);
    GFT_LAUNCH(k_div_1d<E>, dim3(1), dim3(threads), (size_t)E::W * ((size_t)n + std::min(ny, n)) * sizeof(double), st, xs, x_plane, nx, ys,
                       y_plane, ny, res, r_plane, n, fused);
    return true;
}
template bool K<EF64>::div_1d(hipStream_t, const double*, size_t, unsigned, const double*, size_t, unsigned, double*, size_t, unsigned, int);
template bool K<EIv>::div_1d(hipStream_t, const double*, size_t, unsigned, const double*, size_t, unsigned, double*, size_t, unsigned, int);

// ------------------------------------------------------------------------------------------
// The whole division as a ROW WAVEFRONT (round 3)
// ------------------------------------------------------------------------------------------
// The blocked recurrence above finishes slab k0 before slab k0 + 1 starts: n0 slab divisions in a row, each a
// single-workgroup chain of n1 lock-step row divisions — 4096 dependent row divisions for a 64^3 quotient, 22 of its 26
// ms, on one CU of 256 (0.9 % of the FP64 roof).  But the recurrence's dependency graph is only n0 + n1 rows deep:
// quotient row K = (k0, .., k_{L-1}) needs the rows J <= K (componentwise, J != K) and nothing else.  Here every row is a
// TASK of one wave; tasks are claimed in lexicographic order from a global counter (so a claimed task only ever waits
// for tasks claimed before it by running waves — no deadlock whatever the dispatch order), and a task consumes its
// source rows in EXACTLY the reference's order (mt:1162-1192 with mt:984-1012 inside):
//   for level l = 0 .. L-1:   S = 0
//       for (j_l < k_l, j_{l+1} <= k_{l+1}, .., j_{L-1} <= k_{L-1}) in lexicographic order, each within the divisor's box:
//           S += row_product(res[k_0 .. k_{l-1}, j_l .., j_{L-1}],  ys[0 .. 0, k_l - j_l, .., k_{L-1} - j_{L-1}])     (from zero, ascending j)
//       r = (-S) + (l == 0 ? xs[K] inside its box : r of the level above)
//   res[K] = r / ys[0 .. 0, :]   (the lock-step 1-d division)
// waiting on a per-row flag (release / acquire at device scope) before it reads a row another wave produced.  Rows are
// consumed as they become available, so by the time a row's LAST source arrives everything else is already summed: the
// critical path is n0 + n1 row divisions plus one row product each, and the multiply-adds of all rows run side by side
// on all CUs.  A row product keeps one coefficient per lane: the source row's coefficient j reaches all lanes by
// v_readlane, the divisor row slides by one DPP wave shift per step (zeros enter at lane 0: the truncation).  Same
// operations in the same order per coefficient as the host-driven recursion => the same bits (GFT_DIV_WAVEFRONT=0 A/B).
struct DivWfArgs {
    int L;                    // leading axes (tasks); the last axis is the row
    unsigned n[3], m[3], xn[3];   // extents of res / ys / xs on the leading axes
    unsigned nr, mr, xnr;     // row lengths
    size_t rstr[3], ystr[3], xstr[3];  // strides of the leading axes (rows are contiguous)
    unsigned ntasks;
    unsigned* flags;          // [rows] row done; zeroed before the launch
    unsigned* counter;        // next task; zeroed before the launch
    // log_mode (mt:1335-1386): res = log(xs) for the slabs k0 >= 1 (slab 0, a log one dimension down, is the caller's).
    //   level 0:  S = sum_{j0 = max(k0 + 1 - xn0, 1)}^{k0 - 1} sum_{j' lexicographic} rowproduct(xs[k0 - j0, j'], j0 * res[j0, k - j'])
    //             r = (-S) + k0 * xs[K]
    //   levels >= 1 and the row division: the division of the slab by xs[0] — as above with ys = xs[0], on the rows q of
    //             the slab's own quotient (kept in `qb`); finally res[K] = q / k0.
    // log_mode == 2: res = exp(xs) for the slabs k0 >= 1 (mt:1271-1300; slab 0, an exp one dimension down, is the caller's):
    //   res[K] = ( sum_{j0 = 1}^{min(k0, xn0 - 1)} sum_{j' lexicographic} rowproduct(j0 * xs[j0, j'], res[k0 - j0, k - j']) ) / k0
    int log_mode;
    int rev;                  // log_mode 2: the source SLABS in descending j0 — the order in which they become available (1e-10 contract,
                              // see k_rows_wavefront); set by the caller where it would otherwise take the right-looking tiled form
    const unsigned* order;    // task t works on row order[t] of the task rows (anti-diagonal order, see dwf_order); null: t
    int pack;                 // rows <= 32: two source rows per wave (GFT_DWF_PACK=0: one, for A/B)
    double* qb;               // log_mode: the quotient rows before the division by k0 (same layout as res)
    size_t qbp;
};

template <class E>
__device__ inline typename E::V ld_coherent(const double* p, size_t plane, size_t i);
template <>
__device__ inline double ld_coherent<EF64>(const double* p, size_t, size_t i) {
    return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <>
__device__ inline Iv ld_coherent<EIv>(const double* p, size_t plane, size_t i) {
    return Iv{__hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(p + plane + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
}
__device__ inline void st_coherent(double* p, size_t, size_t i, double v) { __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_coherent(double* p, size_t plane, size_t i, Iv v) {
    __hip_atomic_store(p + i, v.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + plane + i, v.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// inner[c] = sum_{j <= c, c - j < mr} a[j] * b[c - j], from zero, ascending j (mul_1d, mt:971-982): the lane holds a[c], b[c]
// (zero beyond the rows' lengths).  Finite rows: positions outside the sum multiply an exact zero (inner + x * 0 == inner:
// a partial sum formed from +0 is never -0) — no masks; otherwise the bounds are applied as predicates.
// `stage`: the LDS region of this lane's row pair, {a: R}{zeros: R}{b: R} doubles per plane (planes DWF_STG apart): a[j] is
// a broadcast read, b[c - j] a read at a per-lane address that walks down one coefficient per step into the zeros staged
// in front of the row (the truncation) — two DS instructions beside two VALU instructions per step; a DPP wave shift of
// the sliding row costs ~6 VALU slots per step instead (tools/microbench_rowconv.hip).  R = 64: one row pair per wave;
// R = 32 (rows <= 32): the two halves of the wave work on two row pairs, each in its own region.
constexpr unsigned DWF_STG = 200;   // doubles per wave and plane: 3 x 64, or 2 regions of 100 (3 x 32, pitched off the other half's banks)
template <class E>
__device__ inline typename E::V row_product(typename E::V a, typename E::V b, unsigned c, unsigned nr, unsigned mr, double* stage, unsigned R) {
    typedef typename E::V V;
    V inner = E::zero();
    E::st(stage, DWF_STG, c, a);
    E::st(stage, DWF_STG, R + c, E::zero());
    E::st(stage, DWF_STG, 2 * R + c, b);
    const double* bl = stage + 2 * R + c;
    if (!any_lane(!elem_finite<E>(a) || !elem_finite<E>(b))) {
#pragma unroll 8
        for (unsigned j = 0; j < nr; ++j) inner = E::add(inner, E::mul(E::ld(stage, DWF_STG, j), E::ld(bl - j, DWF_STG, 0)));
    } else {
        for (unsigned j = 0; j < nr; ++j) {
            const V t = E::add(inner, E::mul(E::ld(stage, DWF_STG, j), E::ld(bl - j, DWF_STG, 0)));
            if (c >= j && c - j < mr) inner = t;
        }
    }
    return inner;
}

template <class E>
struct DwfCfg {
    static constexpr unsigned NW = E::W == 1 ? 16 : 8;  // waves per task (the LDS stages of an interval task are twice as large)
};
// Quotient rows start out as this bit pattern (a NaN payload no arithmetic produces): a consumer loads a source row with
// ONE coherent load — requested a batch ahead — and sees from the row itself whether its producer has stored it (every
// coefficient is an 8-byte store); the per-row flags (release / acquire) remain the authority when a row keeps looking
// unwritten, so a genuine coefficient of that pattern only costs time.
constexpr unsigned long long DWF_EMPTY = 0x7ff8dead0badf00dull;
static const int dwf_pack = 1;
__global__ void __launch_bounds__(256) k_fill_bits(double* p, size_t n, unsigned long long bits) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<unsigned long long*>(p)[i] = bits;
}
__device__ inline bool is_empty_bits(double v) { return (unsigned long long)f64_bits(v) == DWF_EMPTY; }
__device__ inline bool is_empty_bits(Iv v) { return (unsigned long long)f64_bits(v.lo) == DWF_EMPTY || (unsigned long long)f64_bits(v.hi) == DWF_EMPTY; }

// (L, the number of leading axes, is a template parameter: the odometers over those axes are then straight-line code on
// registers — with a run-time L they were loops over dynamically indexed arrays, ~400 instructions per source row, more than
// the row product itself)
template <class E, int L>
__global__ void __launch_bounds__(64 * DwfCfg<E>::NW) k_div_wavefront(const double* __restrict__ xs, size_t xp, const double* __restrict__ ys, size_t yp,
                                                               double* res, size_t rp, DivWfArgs g) {
    typedef typename E::V V;
    // A task (one quotient row) belongs to a WORKGROUP: its source rows are consumed in batches of DWF_NW — every wave waits
    // for, loads and multiplies one source row (the row products of a batch run side by side on the CU's SIMDs, their
    // global latencies overlap), then wave 0 adds the batch's products in the reference's order.  Same operations per
    // coefficient in the same order as one wave doing everything, 1 / DWF_NW of the chain.
    constexpr unsigned DWF_NW = DwfCfg<E>::NW;
    __shared__ double part[2][E::W][DWF_NW][64];  // [buffer][plane][wave][c]
    __shared__ double stage[DWF_NW][E::W][DWF_STG];   // per wave and plane: row_product's {broadcast row, zeros, sliding row}
    __shared__ unsigned s_task;
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // rows of at most 32 coefficients: the two halves of a wave take two source rows of a batch (PK = 2); `c` is the lane's
    // coefficient in the row products and sums, `lane` its coefficient in the row's 1-d division (wave 0, lower half)
    const bool packed = g.pack && g.nr <= 32;
    const unsigned PK = packed ? 2u : 1u, R = packed ? 32u : 64u;
    const unsigned c = packed ? (lane & 31u) : lane, half = packed ? (lane >> 5) : 0u;
    double* const my_stage = &stage[wave][0][0] + (packed ? half * 100u : 0u);
    const bool lg = g.log_mode == 1, ex = g.log_mode == 2;
    unsigned slab_rows = 1;  // rows of one slab (exp: the rows of slab 0 are the caller's and complete)
#pragma unroll
    for (int a = 1; a < L; ++a) slab_rows *= g.n[a];
    const V y0row = lane < g.mr ? E::ld(ys, yp, lane) : E::zero();
    const SlabDiv<E> div_y00(E::ld(ys, yp, 0));
    for (;;) {
        if (threadIdx.x == 0) s_task = atomicAdd(g.counter, 1u);
        __syncthreads();
        const unsigned t = s_task;
        __syncthreads();  // everyone has read s_task before the next claim overwrites it
        if (t >= g.ntasks) break;
        unsigned k[3] = {0, 0, 0};
        unsigned row_id = 0;  // the row's index among ALL rows of res (flags)
        {
            unsigned r = g.order ? g.order[t] : t;
#pragma unroll
            for (int a = L - 1; a >= 1; --a) {
                k[a] = r % g.n[a];
                r /= g.n[a];
            }
            k[0] = r + ((lg || ex) ? 1u : 0u);
#pragma unroll
            for (int a = 0; a < L; ++a) row_id = row_id * g.n[a] + k[a];
        }
        V r_prev = E::zero();  // (meaningful in wave 0)
        unsigned buf = 0;
        auto level = [&](auto lev_c) {
            constexpr int lev = decltype(lev_c)::value;
            V S = E::zero();
            const bool lg0 = (lg && lev == 0) || ex;  // the other operand's rows come from xs, the odometer runs over ITS rows
            // the level's source rows, in the reference's order (the last axis of the odometer fastest):
            //   division (and log's levels >= 1): result row (k_0 .. k_{lev-1}, j_lev .. j_{L-1}) with j_lev in [lo_lev, k_lev),
            //     j_t in [lo_t, k_t]; the other operand's row is (0 .., k_lev - j_lev, ..)
            //   log's level 0: j0 in [max(lo_0, 1), k_0), then j'_t in [0, min(k_t, xn_t - 1)] = the row index in xs[k0 - j0];
            //     the result row is (j0, k_1 - j'_1, ..)
            unsigned lo[3] = {0, 0, 0}, cnt[3] = {1, 1, 1};
            unsigned total = 1;
#pragma unroll
            for (int a = lev; a < L; ++a) {
                if (lg0 && a > 0) {
                    lo[a] = 0;
                    cnt[a] = (k[a] < g.xn[a] ? k[a] : g.xn[a] - 1) + 1;
                } else {
                    const unsigned mm = lg0 ? g.xn[a] : g.m[a];
                    lo[a] = k[a] + 1 > mm ? k[a] + 1 - mm : 0;
                    if (lg0 && lo[a] < 1) lo[a] = 1;
                    unsigned hi = a == lev ? k[a] : k[a] + 1;  // exclusive
                    if (ex) {  // j0 = the index of the xs slab: 1 .. min(k0, xn0 - 1)
                        lo[a] = 1;
                        hi = (k[0] < g.xn[0] ? k[0] : g.xn[0] - 1) + 1;
                    }
                    cnt[a] = hi > lo[a] ? hi - lo[a] : 0;
                }
                total *= cnt[a];
            }
            const double* const coh_base = (lg && lev > 0) ? g.qb : res;  // where the result-side rows live
            const size_t coh_plane = (lg && lev > 0) ? g.qbp : rp;
            // this wave's source row i of the level: where it lies, and its operands REQUESTED (consumed one batch later)
            size_t roff_n = 0;
            unsigned src_n = 0, j0_n = 0;
            V coh_n = E::zero(), oth_n = E::zero();
            auto request = [&](unsigned i0) {  // i0: the wave's first source row of the batch (wave-uniform)
                if (i0 + half >= total) return;
                unsigned rem = i0, j[3] = {0, 0, 0};
#pragma unroll
                for (int a = L - 1; a >= lev; --a) {
                    j[a] = rem % cnt[a];
                    rem /= cnt[a];
                }
                if (half) {  // the upper half's row is the next one of the odometer
                    bool carry = true;
#pragma unroll
                    for (int a = L - 1; a >= lev; --a)
                        if (carry) {
                            if (++j[a] == cnt[a]) j[a] = 0;
                            else carry = false;
                        }
                }
                if (ex && g.rev) j[0] = cnt[0] - 1u - j[0];
#pragma unroll
                for (int a = lev; a < L; ++a) j[a] += lo[a];
                size_t roff = 0, ooff = 0;
                unsigned src = 0;
#pragma unroll
                for (int a = 0; a < L; ++a) {
                    unsigned ra;  // the result row's index on this axis
                    if (a < lev) ra = k[a];
                    else if (ex || (lg0 && a > 0)) ra = k[a] - j[a];
                    else ra = j[a];
                    roff += (size_t)ra * g.rstr[a];
                    src = src * g.n[a] + ra;
                    if (a >= lev) {
                        if (ex) ooff += (size_t)j[a] * g.xstr[a];
                        else if (lg0) ooff += (size_t)(a == 0 ? k[0] - j[0] : j[a]) * g.xstr[a];
                        else ooff += (size_t)(k[a] - j[a]) * g.ystr[a];
                    }
                }
                roff_n = roff;
                src_n = src;
                j0_n = j[0];
                if (lg0) oth_n = c < g.xnr ? E::ld(xs, xp, ooff + c) : E::zero();
                else oth_n = c < g.mr ? E::ld(ys, yp, ooff + c) : E::zero();
                coh_n = c < g.nr ? ld_coherent<E>(coh_base, coh_plane, roff + c) : E::zero();
            };
            request(wave * PK);
            for (unsigned base = 0; base < total; base += DWF_NW * PK) {
                const unsigned i0 = base + wave * PK;
                const bool live = i0 + half < total;
                V coh = live ? coh_n : E::zero();
                const V oth = live ? oth_n : E::zero();
                const size_t roff = roff_n;
                const unsigned src = src_n, j0 = j0_n;
                request(i0 + DWF_NW * PK);
                if (i0 < total) {
                    // a row that still shows the EMPTY pattern has not been stored by its producer (or, once in a blue moon,
                    // holds that pattern for real: then the producer's flag says so)
                    bool confirmed = ex && src < slab_rows;
                    for (unsigned spins = 1; any_lane(live && !confirmed && c < g.nr && is_empty_bits(coh)); ++spins) {
                        __builtin_amdgcn_s_sleep(2);
                        if (live && (spins & 31u) == 0u && __hip_atomic_load(g.flags + src, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                            confirmed = true;  // (what the reload below returns is the row)
                        if (live) coh = c < g.nr ? ld_coherent<E>(coh_base, coh_plane, roff + c) : E::zero();
                    }
                    V prod;
                    if (ex) {  // mul_1d(j0 * xs row, res row)
                        const V scaled = c < g.xnr ? E::mul(oth, E::from_u32(j0)) : E::zero();
                        prod = row_product<E>(scaled, coh, c, g.xnr, g.nr, my_stage, R);
                    } else if (lg0) {  // mul_1d(xs row, j0 * res row): the input's coefficient is the broadcast one
                        const V scaled = c < g.nr ? E::mul(coh, E::from_u32(j0)) : E::zero();
                        prod = row_product<E>(oth, scaled, c, g.xnr, g.nr, my_stage, R);
                    } else {
                        prod = row_product<E>(coh, oth, c, g.nr, g.mr, my_stage, R);
                    }
                    E::st(&part[buf][0][wave][0], (size_t)DWF_NW * 64, lane, prod);  // = [source row of the batch][c]
                }
                __syncthreads();
                if (wave == 0) {
                    // (full batches: constant trip counts, so that the LDS reads are issued together ahead of the chain of
                    // adds instead of one read latency per add)
                    const unsigned nb = total - base < DWF_NW * PK ? total - base : DWF_NW * PK;
                    const double* pb = &part[buf][0][0][0];
                    if (nb == DWF_NW * 2) {
                        V pv[DWF_NW * 2];
#pragma unroll
                        for (unsigned w = 0; w < DWF_NW * 2; ++w) pv[w] = E::ld(pb, (size_t)DWF_NW * 64, (size_t)w * 32 + c);
#pragma unroll
                        for (unsigned w = 0; w < DWF_NW * 2; ++w) S = E::add(S, pv[w]);
                    } else if (nb == DWF_NW && !packed) {
                        V pv[DWF_NW];
#pragma unroll
                        for (unsigned w = 0; w < DWF_NW; ++w) pv[w] = E::ld(pb, (size_t)DWF_NW * 64, (size_t)w * 64 + c);
#pragma unroll
                        for (unsigned w = 0; w < DWF_NW; ++w) S = E::add(S, pv[w]);
                    } else {
                        for (unsigned w = 0; w < nb; ++w) S = E::add(S, E::ld(pb, (size_t)DWF_NW * 64, (size_t)w * R + c));
                    }
                }
                buf ^= 1u;  // the next batch writes the other buffer while wave 0 still reads this one
            }
            if (ex) {
                if (wave == 0) r_prev = S;
                __syncthreads();
                return;
            }
            if (wave == 0) {
                V r = E::neg(S);
                if (lev == 0) {
                    bool in_x = c < g.xnr;
                    size_t xoff = 0;
#pragma unroll
                    for (int a = 0; a < L; ++a) {
                        if (k[a] >= g.xn[a]) in_x = false;
                        xoff += (size_t)k[a] * g.xstr[a];
                    }
                    if (in_x) {
                        const V xin = E::ld(xs, xp, xoff + c);
                        r = E::add(r, lg ? E::mul(E::from_u32(k[0]), xin) : xin);
                    }
                } else {
                    r = E::add(r, r_prev);
                }
                r_prev = r;
            }
            __syncthreads();  // a level's last batch buffer is free again before the next level reuses it
        };
        level(std::integral_constant<int, 0>{});
        if (!ex) {
            if constexpr (L > 1) level(std::integral_constant<int, 1>{});
            if constexpr (L > 2) level(std::integral_constant<int, 2>{});
        }
        if (ex) {
            if (wave == 0) {
                size_t qoff = 0;
#pragma unroll
                for (int a = 0; a < L; ++a) qoff += (size_t)k[a] * g.rstr[a];
                if (lane < g.nr) st_coherent(res, rp, qoff + lane, E::div(r_prev, E::from_u32(k[0])));  // mt:1298
                __threadfence();
                if (lane == 0) __hip_atomic_store(g.flags + row_id, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else
        if (wave == 0) {
            // ---- the row's 1-d division by ys[0 .. 0, :] in lock step (mt:1162-1185): lane j's coefficient becomes final
            // at step j and reaches the lanes above it by v_readlane; the divisor row slides by one DPP shift per step
            if (lane >= g.nr) r_prev = E::zero();  // (packed rows: the upper half only mirrored the lower half's sums)
            V cur1 = E::zero(), mine = E::zero(), ysl = y0row;  // ysl[c] = y0[c - j] at step j (zero for c < j and beyond the divisor's row)
            const bool fin = !any_lane(!elem_finite<E>(r_prev)) && !any_lane(!elem_finite<E>(y0row));
            for (unsigned jj = 0; jj < g.nr; ++jj) {
                const V q = div_y00(bcast_lane<E>(E::add(E::neg(cur1), r_prev), jj));
                if (lane == jj) mine = q;
                const V tnew = E::add(cur1, E::mul(q, ysl));
                if (fin && elem_finite<E>(q)) {
                    cur1 = tnew;  // positions outside the sum multiply a shifted-in zero
                } else if (lane > jj && lane - jj < g.mr) {
                    cur1 = tnew;
                }
                ysl = wave_shr1<E>(ysl);
            }
            size_t qoff = 0;
#pragma unroll
            for (int a = 0; a < L; ++a) qoff += (size_t)k[a] * g.rstr[a];
            if (lane < g.nr) {
                if (lg) {  // res[K] = q / k0 (mt:1384); the slab's own later rows read q itself
                    st_coherent(g.qb, g.qbp, qoff + lane, mine);
                    st_coherent(res, rp, qoff + lane, E::div(mine, E::from_u32(k[0])));
                } else {
                    st_coherent(res, rp, qoff + lane, mine);
                }
            }
            __threadfence();  // the row is visible device-wide before its flag is
            if (lane == 0) __hip_atomic_store(g.flags + row_id, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// ------------------------------------------------------------------------------------------
// k_div_wavefront, four source rows per wave (round 4; f64, rows of 33 .. 64 coefficients)
// ------------------------------------------------------------------------------------------
// The row product above costs two 8-byte LDS reads per multiply-add (the broadcast coefficient and the sliding one): with
// four SIMDs sharing one LDS pipe the kernel is LDS-bound four times over (64^3 div: 1.9 TMAC/s).  Here a wave works on FOUR
// source rows of a batch at once, one per 16-lane group, and a lane owns FOUR consecutive coefficients c = 4 l .. 4 l + 3 of
// its group's row product: the broadcast coefficient a[j] is read once per step for four multiply-adds, and the sliding
// operand is a register window — step j + 1 needs b[4 l - j - 1], ONE new value, the other three move up — so a step is two
// LDS reads for four multiply-adds instead of eight.  Each output still receives its terms from zero in ascending j
// (mul_1d, mt:971-982), the batch's products are still added in source order by wave 0: the same bits.  Everything else —
// tasks, claim order, levels, publication, the lock-step division — is k_div_wavefront's.
constexpr unsigned QNW = 8;             // waves per task
// GL lanes per group (16: rows of 33 .. 64 coefficients, four groups = four source rows per wave; 8: rows <= 32, eight per wave)
template <int GL>
struct QCfg {
    static constexpr unsigned NG = 64 / GL;          // groups (source rows) per wave
    static constexpr unsigned RC = 4 * GL;           // coefficients a group holds
    static constexpr unsigned QS = NG * QNW;         // source rows per batch
    static constexpr unsigned QSTG = GL == 16 ? 198 : 101;  // doubles per group: {a: RC}{zeros + b: 2 RC + pads}, see qphys
};

// Bank layout of a group's sliding operand.  Lane l reads b[4 l - 1 - j]: neighbouring lanes are 4 doubles = 8 banks apart, so
// lanes l and l + 8 (32 doubles apart) would meet in one bank — one pad double per 32 moves them two banks apart — and the
// group pitch (198 doubles = 12 banks mod 64, i.e. 4 mod 8, for 16-lane groups; 101 = 10 banks, i.e. 2 mod 8, for 8-lane groups)
// puts a half-wave's groups on disjoint banks: its 32 lanes read conflict-free, and the groups' broadcast reads of a[j] hit
// different banks.
__device__ inline unsigned qphys(unsigned i) { return i + (i >> 5); }

// inner[r] = sum_{j < nr} a[j] * b[4 l + r - j]  (b = 0 below index 0), ascending j, for r = 0 .. 3.  `stage`: this group's region.
template <int GL>
__device__ inline void row_product4(double (&inner)[4], const double (&a4)[4], const double (&b4)[4], unsigned l, unsigned nr, unsigned mr,
                                    double* stage) {
    constexpr unsigned RC = QCfg<GL>::RC;
    double* const zb = stage + RC;  // [zeros: RC][b: RC] through qphys
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        stage[4 * l + r] = a4[r];
        zb[qphys(4 * l + r)] = 0.0;
        zb[qphys(RC + 4 * l + r)] = b4[r];
        inner[r] = 0.0;
    }
    // w[r] = b[4 l + r - j] at step j; the value that enters at r = 0 in step j + 1 is b[4 l - j - 1] = zb[qphys(RC + 4 l - 1 - j)]
    double w0 = b4[0], w1 = b4[1], w2 = b4[2], w3 = b4[3];
    const unsigned nbi = RC + 4 * l - 1;
    auto nb = [&](unsigned j) { return zb[qphys(nbi - j)]; };
    bool fin = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) fin = fin && finite_d(a4[r]) && finite_d(b4[r]);
    if (!any_lane(!fin)) {
        // four steps per iteration: the window rotates through its registers without moves
        unsigned j = 0;
        for (; j + 4 <= nr; j += 4) {
            const double a0 = stage[j], a1 = stage[j + 1], a2 = stage[j + 2], a3 = stage[j + 3];
            const double n0 = nb(j), n1 = nb(j + 1), n2 = nb(j + 2), n3 = nb(j + 3);
            inner[0] = inner[0] + a0 * w0; inner[1] = inner[1] + a0 * w1; inner[2] = inner[2] + a0 * w2; inner[3] = inner[3] + a0 * w3;
            inner[0] = inner[0] + a1 * n0; inner[1] = inner[1] + a1 * w0; inner[2] = inner[2] + a1 * w1; inner[3] = inner[3] + a1 * w2;
            inner[0] = inner[0] + a2 * n1; inner[1] = inner[1] + a2 * n0; inner[2] = inner[2] + a2 * w0; inner[3] = inner[3] + a2 * w1;
            inner[0] = inner[0] + a3 * n2; inner[1] = inner[1] + a3 * n1; inner[2] = inner[2] + a3 * n0; inner[3] = inner[3] + a3 * w0;
            w3 = n0; w2 = n1; w1 = n2; w0 = n3;
        }
        for (; j < nr; ++j) {
            const double a0 = stage[j], n0 = nb(j);
            inner[0] = inner[0] + a0 * w0; inner[1] = inner[1] + a0 * w1; inner[2] = inner[2] + a0 * w2; inner[3] = inner[3] + a0 * w3;
            w3 = w2; w2 = w1; w1 = w0; w0 = n0;
        }
    } else {
        for (unsigned j = 0; j < nr; ++j) {
            const double a0 = stage[j], n0 = nb(j);
            const double wv[4] = {w0, w1, w2, w3};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned c = 4 * l + r;
                const double t = inner[r] + a0 * wv[r];
                if (c >= j && c - j < mr) inner[r] = t;
            }
            w3 = w2; w2 = w1; w1 = w0; w0 = n0;
        }
    }
}

template <int L, int GL>
__global__ void __launch_bounds__(64 * QNW) k_div_wavefront_q(const double* __restrict__ xs, size_t xp, const double* __restrict__ ys, size_t yp,
                                                         double* res, size_t rp, DivWfArgs g) {
    typedef EF64 E;
    typedef double V;
    extern __shared__ double q_lds[];
    constexpr unsigned NG = QCfg<GL>::NG, QS = QCfg<GL>::QS, QSTG = QCfg<GL>::QSTG;
    double* const part = q_lds;                                  // [2][QS][64]
    double* const stage_all = q_lds + 2 * QS * 64;               // [QNW][NG][QSTG]
    __shared__ unsigned s_task;
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned g4 = lane / GL, l = lane % GL, c0 = 4u * l;
    double* const my_stage = stage_all + ((size_t)wave * NG + g4) * QSTG;
    const bool lg = g.log_mode == 1, ex = g.log_mode == 2;
    unsigned slab_rows = 1;
#pragma unroll
    for (int a = 1; a < L; ++a) slab_rows *= g.n[a];
    const V y0row = lane < g.mr ? E::ld(ys, yp, lane) : E::zero();
    const SlabDiv<E> div_y00(E::ld(ys, yp, 0));
    for (;;) {
        if (threadIdx.x == 0) s_task = atomicAdd(g.counter, 1u);
        __syncthreads();
        const unsigned t = s_task;
        __syncthreads();
        if (t >= g.ntasks) break;
        unsigned k[3] = {0, 0, 0};
        unsigned row_id = 0;
        {
            unsigned r = g.order ? g.order[t] : t;
#pragma unroll
            for (int a = L - 1; a >= 1; --a) {
                k[a] = r % g.n[a];
                r /= g.n[a];
            }
            k[0] = r + ((lg || ex) ? 1u : 0u);
#pragma unroll
            for (int a = 0; a < L; ++a) row_id = row_id * g.n[a] + k[a];
        }
        V r_prev = E::zero();
        unsigned buf = 0;
        auto level = [&](auto lev_c) {
            constexpr int lev = decltype(lev_c)::value;
            V S = E::zero();
            const bool lg0 = (lg && lev == 0) || ex;
            unsigned lo[3] = {0, 0, 0}, cnt[3] = {1, 1, 1};
            unsigned total = 1;
#pragma unroll
            for (int a = lev; a < L; ++a) {
                if (lg0 && a > 0) {
                    lo[a] = 0;
                    cnt[a] = (k[a] < g.xn[a] ? k[a] : g.xn[a] - 1) + 1;
                } else {
                    const unsigned mm = lg0 ? g.xn[a] : g.m[a];
                    lo[a] = k[a] + 1 > mm ? k[a] + 1 - mm : 0;
                    if (lg0 && lo[a] < 1) lo[a] = 1;
                    unsigned hi = a == lev ? k[a] : k[a] + 1;
                    if (ex) {
                        lo[a] = 1;
                        hi = (k[0] < g.xn[0] ? k[0] : g.xn[0] - 1) + 1;
                    }
                    cnt[a] = hi > lo[a] ? hi - lo[a] : 0;
                }
                total *= cnt[a];
            }
            const double* const coh_base = (lg && lev > 0) ? g.qb : res;
            size_t roff_n = 0;
            unsigned src_n = 0, j0_n = 0;
            V coh_n[4] = {0, 0, 0, 0}, oth_n[4] = {0, 0, 0, 0};
            auto request = [&](unsigned i0) {  // i0: the wave's first source row of the batch; this lane's group takes row i0 + g4
                if (i0 + g4 >= total) return;
                unsigned rem = i0 + g4, j[3] = {0, 0, 0};
#pragma unroll
                for (int a = L - 1; a >= lev; --a) {
                    j[a] = rem % cnt[a];
                    rem /= cnt[a];
                }
                if (ex && g.rev) j[0] = cnt[0] - 1u - j[0];
#pragma unroll
                for (int a = lev; a < L; ++a) j[a] += lo[a];
                size_t roff = 0, ooff = 0;
                unsigned src = 0;
#pragma unroll
                for (int a = 0; a < L; ++a) {
                    unsigned ra;
                    if (a < lev) ra = k[a];
                    else if (ex || (lg0 && a > 0)) ra = k[a] - j[a];
                    else ra = j[a];
                    roff += (size_t)ra * g.rstr[a];
                    src = src * g.n[a] + ra;
                    if (a >= lev) {
                        if (ex) ooff += (size_t)j[a] * g.xstr[a];
                        else if (lg0) ooff += (size_t)(a == 0 ? k[0] - j[0] : j[a]) * g.xstr[a];
                        else ooff += (size_t)(k[a] - j[a]) * g.ystr[a];
                    }
                }
                roff_n = roff;
                src_n = src;
                j0_n = j[0];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned c = c0 + r;
                    if (lg0) oth_n[r] = c < g.xnr ? xs[ooff + c] : 0.0;
                    else oth_n[r] = c < g.mr ? ys[ooff + c] : 0.0;
                    coh_n[r] = c < g.nr ? ld_coherent<E>(coh_base, 0, roff + c) : 0.0;
                }
            };
            request(wave * NG);
            for (unsigned base = 0; base < total; base += QS) {
                const unsigned i0 = base + wave * NG;
                const bool live = i0 + g4 < total;
                V coh[4], oth[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    coh[r] = live ? coh_n[r] : 0.0;
                    oth[r] = live ? oth_n[r] : 0.0;
                }
                const size_t roff = roff_n;
                const unsigned src = src_n, j0 = j0_n;
                request(i0 + QS);
                if (i0 < total) {
                    bool confirmed = ex && src < slab_rows;
                    auto empty_any = [&]() {
                        bool e = false;
#pragma unroll
                        for (int r = 0; r < 4; ++r) e = e || (c0 + r < g.nr && is_empty_bits(coh[r]));
                        return e;
                    };
                    for (unsigned spins = 1; any_lane(live && !confirmed && empty_any()); ++spins) {
                        __builtin_amdgcn_s_sleep(2);
                        if (live && (spins & 31u) == 0u && __hip_atomic_load(g.flags + src, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                            confirmed = true;
                        if (live) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) coh[r] = c0 + r < g.nr ? ld_coherent<E>(coh_base, 0, roff + c0 + r) : 0.0;
                        }
                    }
                    V prod[4];
                    if (ex) {  // mul_1d(j0 * xs row, res row)
                        V sc[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[r] = c0 + r < g.xnr ? oth[r] * (double)j0 : 0.0;
                        row_product4<GL>(prod, sc, coh, l, g.xnr, g.nr, my_stage);
                    } else if (lg0) {  // mul_1d(xs row, j0 * res row)
                        V sc[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[r] = c0 + r < g.nr ? coh[r] * (double)j0 : 0.0;
                        row_product4<GL>(prod, oth, sc, l, g.xnr, g.nr, my_stage);
                    } else {
                        row_product4<GL>(prod, coh, oth, l, g.nr, g.mr, my_stage);
                    }
                    double* pd = part + ((size_t)buf * QS + (wave * NG + g4)) * 64 + c0;  // = [source row of the batch][c]
#pragma unroll
                    for (int r = 0; r < 4; ++r) pd[r] = prod[r];
                }
                __syncthreads();
                if (wave == 0) {
                    const unsigned nbt = total - base < QS ? total - base : QS;
                    const double* pb = part + (size_t)buf * QS * 64 + lane;
                    if (nbt == QS) {
                        // (32 products' reads issued together ahead of their chain of adds)
#pragma unroll
                        for (unsigned w0 = 0; w0 < QS; w0 += 32) {
                            V pv[32];
#pragma unroll
                            for (unsigned w = 0; w < 32; ++w) pv[w] = pb[(size_t)(w0 + w) * 64];
#pragma unroll
                            for (unsigned w = 0; w < 32; ++w) S = S + pv[w];
                        }
                    } else {
                        for (unsigned w = 0; w < nbt; ++w) S = S + pb[(size_t)w * 64];
                    }
                }
                buf ^= 1u;
            }
            if (ex) {
                if (wave == 0) r_prev = S;
                __syncthreads();
                return;
            }
            if (wave == 0) {
                V r = E::neg(S);
                if (lev == 0) {
                    bool in_x = lane < g.xnr;
                    size_t xoff = 0;
#pragma unroll
                    for (int a = 0; a < L; ++a) {
                        if (k[a] >= g.xn[a]) in_x = false;
                        xoff += (size_t)k[a] * g.xstr[a];
                    }
                    if (in_x) {
                        const V xin = xs[xoff + lane];
                        r = E::add(r, lg ? E::mul(E::from_u32(k[0]), xin) : xin);
                    }
                } else {
                    r = E::add(r, r_prev);
                }
                r_prev = r;
            }
            __syncthreads();
        };
        level(std::integral_constant<int, 0>{});
        if (!ex) {
            if constexpr (L > 1) level(std::integral_constant<int, 1>{});
            if constexpr (L > 2) level(std::integral_constant<int, 2>{});
        }
        if (ex) {
            if (wave == 0) {
                size_t qoff = 0;
#pragma unroll
                for (int a = 0; a < L; ++a) qoff += (size_t)k[a] * g.rstr[a];
                if (lane < g.nr) st_coherent(res, rp, qoff + lane, E::div(r_prev, E::from_u32(k[0])));
                __threadfence();
                if (lane == 0) __hip_atomic_store(g.flags + row_id, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (wave == 0) {
            if (lane >= g.nr) r_prev = E::zero();
            V cur1 = E::zero(), mine = E::zero(), ysl = y0row;
            const bool fin = !any_lane(!elem_finite<E>(r_prev)) && !any_lane(!elem_finite<E>(y0row));
            for (unsigned jj = 0; jj < g.nr; ++jj) {
                const V q = div_y00(bcast_lane<E>(E::add(E::neg(cur1), r_prev), jj));
                if (lane == jj) mine = q;
                const V tnew = E::add(cur1, E::mul(q, ysl));
                if (fin && elem_finite<E>(q)) {
                    cur1 = tnew;
                } else if (lane > jj && lane - jj < g.mr) {
                    cur1 = tnew;
                }
                ysl = wave_shr1<E>(ysl);
            }
            size_t qoff = 0;
#pragma unroll
            for (int a = 0; a < L; ++a) qoff += (size_t)k[a] * g.rstr[a];
            if (lane < g.nr) {
                if (lg) {
                    st_coherent(g.qb, g.qbp, qoff + lane, mine);
                    st_coherent(res, rp, qoff + lane, E::div(mine, E::from_u32(k[0])));
                } else {
                    st_coherent(res, rp, qoff + lane, mine);
                }
            }
            __threadfence();
            if (lane == 0) __hip_atomic_store(g.flags + row_id, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <class E>
static void launch_dwf(hipStream_t st, unsigned blocks, const double* xs, size_t xp, const double* ys, size_t yp, double* res, size_t rp, const DivWfArgs& g) {
    // f64 rows of 33 .. 64 coefficients: four source rows per wave, four coefficients per lane (k_div_wavefront_q)
    static const int quad_on = 1;
    // (where a row has thousands of source rows — 64^3 div 4.8 -> 4.0 ms, 24^4 7.7 -> 6.2; thin or small quotients, whose time is the
    // chain of rows, lose to its larger batches: 1000 x 32 6.2 -> 8.0 ms, 32^3 0.53 -> 0.62 — they keep one or two rows per wave)
    size_t max_sources = 1;
    for (int a = 0; a < g.L; ++a) max_sources *= g.n[a];
    if constexpr (E::W == 1) {
        if (quad_on && g.nr >= 8 && g.nr <= 64 && (max_sources >= 2048 || quad_on == 2)) {
            const bool wide = g.nr > 32;
            const size_t lds = wide ? sizeof(double) * (2 * QCfg<16>::QS * 64 + (size_t)QNW * QCfg<16>::NG * QCfg<16>::QSTG)
                                    : sizeof(double) * (2 * QCfg<8>::QS * 64 + (size_t)QNW * QCfg<8>::NG * QCfg<8>::QSTG);
            static bool attr_set = false;
            bool ok = true;
            if (!attr_set) {
                auto set = [](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess; };
                ok = set((const void*)k_div_wavefront_q<1, 16>) && set((const void*)k_div_wavefront_q<2, 16>) && set((const void*)k_div_wavefront_q<3, 16>) &&
                     set((const void*)k_div_wavefront_q<1, 8>) && set((const void*)k_div_wavefront_q<2, 8>) && set((const void*)k_div_wavefront_q<3, 8>);
                if (!ok) (void)hipGetLastError();
                attr_set = ok;
            }
            if (ok) {
                const dim3 qgrid(blocks), qblock(64 * QNW);
                if (wide) {
                    if (g.L == 1) GFT_LAUNCH((k_div_wavefront_q<1, 16>), qgrid, qblock, lds, st, xs, xp, ys, yp, res, rp, g);
                    else if (g.L == 2) GFT_LAUNCH((k_div_wavefront_q<2, 16>), qgrid, qblock, lds, st, xs, xp, ys, yp, res, rp, g);
                    else GFT_LAUNCH((k_div_wavefront_q<3, 16>), qgrid, qblock, lds, st, xs, xp, ys, yp, res, rp, g);
                } else {
                    if (g.L == 1) GFT_LAUNCH((k_div_wavefront_q<1, 8>), qgrid, qblock, lds, st, xs, xp, ys, yp, res, rp, g);
                    else if (g.L == 2) GFT_LAUNCH((k_div_wavefront_q<2, 8>), qgrid, qblock, lds, st, xs, xp, ys, yp, res, rp, g);
                    else GFT_LAUNCH((k_div_wavefront_q<3, 8>), qgrid, qblock, lds, st, xs, xp, ys, yp, res, rp, g);
                }
                return;
            }
        }
    }
    const dim3 grid(blocks), block(64 * DwfCfg<E>::NW);
    if (g.L == 1) GFT_LAUNCH((k_div_wavefront<E, 1>), grid, block, 0, st, xs, xp, ys, yp, res, rp, g);
    else if (g.L == 2) GFT_LAUNCH((k_div_wavefront<E, 2>), grid, block, 0, st, xs, xp, ys, yp, res, rp, g);
    else GFT_LAUNCH((k_div_wavefront<E, 3>), grid, block, 0, st, xs, xp, ys, yp, res, rp, g);
}

// Claim order of the tasks.  Any order in which a row comes after the rows it reads is deadlock-free (a claimed task only
// waits for tasks claimed before it, and those belong to running workgroups).  Lexicographic order has every row wait for
// the row claimed just before it — (k0, k1 - 1) is the LAST source of (k0, k1), so the rows of a slab run as a chain of
// row divisions; in order of the ANTI-DIAGONAL k0 + .. + k_{L-1} (lexicographic within one) a row's sources are all at
// least one diagonal — dozens of tasks — older, and the rows of a diagonal are independent.  The table (one word per
// task row) is built on the host once per shape and kept on the device.
struct DwfOrderKey {
    unsigned L, n[3], first;
    bool operator<(const DwfOrderKey& o) const { return std::memcmp(this, &o, sizeof(*this)) < 0; }
};
static std::map<DwfOrderKey, unsigned*>& dwf_orders() {
    static std::map<DwfOrderKey, unsigned*> m;
    return m;
}
static const int dwf_diag = 1;
// rows (k0 >= first, k1, ..) of an n[0] x .. x n[L-1] grid, task-relative index (row index - first * rows per slab)
static const unsigned* dwf_order(int L, const unsigned* n, unsigned first) {
    if (!dwf_diag || L < 2) return nullptr;
    DwfOrderKey key;
    std::memset(&key, 0, sizeof(key));
    key.L = (unsigned)L;
    for (int a = 0; a < L; ++a) key.n[a] = n[a];
    key.first = first;
    auto it = dwf_orders().find(key);
    if (it != dwf_orders().end()) return it->second;
    size_t slab = 1, maxd = 0;
    for (int a = 1; a < L; ++a) slab *= n[a];
    for (int a = 0; a < L; ++a) maxd += n[a] - 1;
    const size_t ntasks = (size_t)(n[0] - first) * slab;
    std::vector<unsigned> start(maxd + 2, 0), tab(ntasks);
    auto diag_of = [&](size_t t) {
        size_t r = t, d = 0;
        for (int a = L - 1; a >= 1; --a) {
            d += r % n[a];
            r /= n[a];
        }
        return d + r + first;
    };
    for (size_t t = 0; t < ntasks; ++t) start[diag_of(t) + 1]++;
    for (size_t d = 0; d + 1 < start.size(); ++d) start[d + 1] += start[d];
    for (size_t t = 0; t < ntasks; ++t) tab[start[diag_of(t)]++] = (unsigned)t;   // (ascending t within a diagonal)
    unsigned* dev = nullptr;
    if (hipMalloc(&dev, sizeof(unsigned) * ntasks) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipMemcpy(dev, tab.data(), sizeof(unsigned) * ntasks, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(dev);
        return nullptr;
    }
    if (dwf_orders().size() >= 64) {  // (shapes of a run are few; bound the cache anyway)
        for (auto& kv : dwf_orders()) (void)hipFree(kv.second);
        dwf_orders().clear();
    }
    dwf_orders()[key] = dev;
    return dev;
}
void dwf_release_orders() {
    for (auto& kv : dwf_orders()) (void)hipFree(kv.second);
    dwf_orders().clear();
}

template <class E>
bool K<E>::div_wavefront(hipStream_t st, const double* xs, size_t x_plane, const unsigned* xshape, const double* ys, size_t y_plane,
                         const unsigned* yshape, double* res, size_t r_plane, const unsigned* rshape, int nd, unsigned* flags_and_counter) {
    if (nd < 2 || nd > 4) return false;
    DivWfArgs g;
    std::memset(&g, 0, sizeof(g));
    g.L = nd - 1;
    g.pack = dwf_pack;
    g.nr = rshape[nd - 1];
    g.mr = yshape[nd - 1];
    g.xnr = xshape[nd - 1];
    if (g.nr < 2 || g.nr > 64 || g.mr > g.nr || g.xnr > g.nr) return false;
    size_t rs = g.nr, ysd = g.mr, xsd = g.xnr, ntasks = 1;
    for (int a = g.L - 1; a >= 0; --a) {
        g.n[a] = rshape[a];
        g.m[a] = yshape[a];
        g.xn[a] = xshape[a];
        if (g.m[a] > g.n[a] || g.xn[a] > g.n[a] || g.n[a] == 0) return false;
        g.rstr[a] = rs;
        g.ystr[a] = ysd;
        g.xstr[a] = xsd;
        rs *= rshape[a];
        ysd *= yshape[a];
        xsd *= xshape[a];
        ntasks *= rshape[a];
    }
    if (ntasks > 0x7fffffffu) return false;
    g.ntasks = (unsigned)ntasks;
    g.flags = flags_and_counter;
    g.counter = flags_and_counter + ntasks;
    g.order = dwf_order(g.L, g.n, 0);
    // enough waves to keep every SIMD busy with several tasks; all of them persistent (they claim tasks until none is left)
    // persistent workgroups (they claim tasks until none is left): a few per CU so that the SIMDs stay busy while some wait
    const unsigned blocks = (unsigned)std::min<size_t>(ntasks, (size_t)256 * 2);
    for (int pl = 0; pl < E::W; ++pl) {
        const size_t nel = ntasks * g.nr;
        GFT_LAUNCH(k_fill_bits, dim3((unsigned)std::min<size_t>((nel + 255) / 256, 2048)), dim3(256), 0, st, res + (size_t)pl * r_plane, nel, DWF_EMPTY);
    }
    launch_dwf<E>(st, blocks, xs, x_plane, ys, y_plane, res, r_plane, g);
    return true;
}
// res[1..] = log(xs)[1..] (slabs k0 >= 1; mt:1335-1386) as the same row wavefront.  `qbuf`: a tensor like res for the slab
// quotients before the division by k0; `flags_and_counter`: (rows of res + 1) zeroed words.
template <class E>
bool K<E>::log_wavefront(hipStream_t st, const double* xs, size_t x_plane, const unsigned* xshape, double* res, size_t r_plane,
                         const unsigned* rshape, int nd, double* qbuf, size_t q_plane, unsigned* flags_and_counter) {
    if (nd < 2 || nd > 4) return false;
    DivWfArgs g;
    std::memset(&g, 0, sizeof(g));
    g.L = nd - 1;
    g.pack = dwf_pack;
    g.log_mode = 1;
    g.nr = rshape[nd - 1];
    g.xnr = xshape[nd - 1];
    g.mr = g.xnr;  // the divisor of the slab divisions is xs[0]
    if (g.nr < 2 || g.nr > 64 || g.xnr > g.nr || rshape[0] < 2) return false;
    size_t rs = g.nr, xsd = g.xnr, rows = 1;
    for (int a = g.L - 1; a >= 0; --a) {
        g.n[a] = rshape[a];
        g.xn[a] = xshape[a];
        g.m[a] = xshape[a];
        if (g.xn[a] > g.n[a] || g.n[a] == 0 || g.xn[a] == 0) return false;
        g.rstr[a] = rs;
        g.xstr[a] = xsd;
        g.ystr[a] = xsd;
        rs *= rshape[a];
        xsd *= xshape[a];
        rows *= rshape[a];
    }
    const size_t slab_rows = rows / rshape[0], ntasks = rows - slab_rows;
    if (rows > 0x7fffffffu) return false;
    g.ntasks = (unsigned)ntasks;
    g.flags = flags_and_counter;
    g.counter = flags_and_counter + rows;
    g.order = dwf_order(g.L, g.n, 1);
    g.qb = qbuf;
    g.qbp = q_plane;
    const size_t slab_el = slab_rows * g.nr, nel = ntasks * g.nr;
    for (int pl = 0; pl < E::W; ++pl) {
        const unsigned fb = (unsigned)std::min<size_t>((nel + 255) / 256, 2048);
        GFT_LAUNCH(k_fill_bits, dim3(fb), dim3(256), 0, st, res + (size_t)pl * r_plane + slab_el, nel, DWF_EMPTY);
        GFT_LAUNCH(k_fill_bits, dim3(fb), dim3(256), 0, st, qbuf + (size_t)pl * q_plane + slab_el, nel, DWF_EMPTY);
    }
    const unsigned blocks = (unsigned)std::min<size_t>(ntasks, (size_t)256 * 2);
    launch_dwf<E>(st, blocks, xs, x_plane, xs, x_plane, res, r_plane, g);
    return true;
}
// res[1..] = exp(xs)[1..] (slabs k0 >= 1; mt:1271-1300) as the same row wavefront: no division, the row sum / k0.
template <class E>
bool K<E>::exp_wavefront(hipStream_t st, const double* xs, size_t x_plane, const unsigned* xshape, double* res, size_t r_plane,
                         const unsigned* rshape, int nd, unsigned* flags_and_counter, int arrival_order) {
    if (nd < 2 || nd > 4) return false;
    DivWfArgs g;
    std::memset(&g, 0, sizeof(g));
    g.rev = arrival_order;
    g.L = nd - 1;
    g.pack = dwf_pack;
    g.log_mode = 2;
    g.nr = rshape[nd - 1];
    g.xnr = xshape[nd - 1];
    g.mr = g.xnr;
    if (g.nr < 2 || g.nr > 64 || g.xnr > g.nr || rshape[0] < 2) return false;
    size_t rs = g.nr, xsd = g.xnr, rows = 1;
    for (int a = g.L - 1; a >= 0; --a) {
        g.n[a] = rshape[a];
        g.xn[a] = xshape[a];
        g.m[a] = xshape[a];
        if (g.xn[a] > g.n[a] || g.n[a] == 0 || g.xn[a] == 0) return false;
        g.rstr[a] = rs;
        g.xstr[a] = xsd;
        g.ystr[a] = xsd;
        rs *= rshape[a];
        xsd *= xshape[a];
        rows *= rshape[a];
    }
    const size_t slab_rows = rows / rshape[0], ntasks = rows - slab_rows;
    if (rows > 0x7fffffffu) return false;
    g.ntasks = (unsigned)ntasks;
    g.flags = flags_and_counter;
    g.counter = flags_and_counter + rows;
    g.order = dwf_order(g.L, g.n, 1);
    const size_t slab_el = slab_rows * g.nr, nel = ntasks * g.nr;
    for (int pl = 0; pl < E::W; ++pl)
        GFT_LAUNCH(k_fill_bits, dim3((unsigned)std::min<size_t>((nel + 255) / 256, 2048)), dim3(256), 0, st, res + (size_t)pl * r_plane + slab_el, nel, DWF_EMPTY);
    const unsigned blocks = (unsigned)std::min<size_t>(ntasks, (size_t)256 * 2);
    launch_dwf<E>(st, blocks, xs, x_plane, xs, x_plane, res, r_plane, g);
    return true;
}
template bool K<EF64>::exp_wavefront(hipStream_t, const double*, size_t, const unsigned*, double*, size_t, const unsigned*, int, unsigned*, int);
template bool K<EIv>::exp_wavefront(hipStream_t, const double*, size_t, const unsigned*, double*, size_t, const unsigned*, int, unsigned*, int);
template bool K<EF64>::log_wavefront(hipStream_t, const double*, size_t, const unsigned*, double*, size_t, const unsigned*, int, double*, size_t, unsigned*);
template bool K<EIv>::log_wavefront(hipStream_t, const double*, size_t, const unsigned*, double*, size_t, const unsigned*, int, double*, size_t, unsigned*);

template bool K<EF64>::div_wavefront(hipStream_t, const double*, size_t, const unsigned*, const double*, size_t, const unsigned*, double*, size_t,
                                     const unsigned*, int, unsigned*);
template bool K<EIv>::div_wavefront(hipStream_t, const double*, size_t, const unsigned*, const double*, size_t, const unsigned*, double*, size_t,
                                    const unsigned*, int, unsigned*);

template <class E>
bool K<E>::div_2d(hipStream_t st, const double* x, size_t x_plane, unsigned nx1, unsigned nx2, size_t x_rstride, const double* y,
                  size_t y_plane, unsigned ny1, unsigned ny2, double* res, size_t r_plane, unsigned n1, unsigned n2, int fused, unsigned log_k, double* res2,
                  size_t r2_plane) {
    if (n1 == 0 || n2 < 2 || n2 > 1024 || ny1 > n1 || ny2 > n2) return false;
    Div2dArgs g;
    g.n1 = n1; g.n2 = n2; g.ny1 = ny1; g.ny2 = ny2; g.nx1 = nx1; g.nx2 = nx2;
    g.x_rstride = x_rstride;
    g.fused = fused;
    g.log_k = log_k;
    g.res2 = res2;
    g.r2p = r2_plane;
    if (fused == 2 && (!res2 || log_k == 0)) return false;
    g.diag = 0;
    g.n2p = n2 | 1;   // odd pitches: rows of one column do not share a bank
    g.ny2p = ny2 | 1;
    if (n2 <= 64) {
        const size_t lds = sizeof(double) * E::W * ((size_t)ny1 * g.ny2p + (size_t)n1 * g.n2p + 128);
        if (lds <= 150 * 1024) {
            static bool attr_set64 = false;
            if (!attr_set64) {
                if (hipFuncSetAttribute((const void*)k_div_2d_rows64<E>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
                    (void)hipGetLastError();
                    return false;
                }
                attr_set64 = true;
            }
            const unsigned waves = std::max(1u, std::min<unsigned>(D2_NW, n1));
            GFT_LAUNCH(k_div_2d_rows64<E>, dim3(1), dim3(64 * waves), lds, st, x, x_plane, y, y_plane, res, r_plane, g);
            return true;
        }
    }
    const size_t lds = sizeof(double) * E::W * ((size_t)ny1 * g.ny2p + (size_t)n1 * g.n2p + (size_t)D2_NW * g.n2p + g.n2p);
    if (lds > 150 * 1024) return false;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_div_2d<E>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    // threads: one per row element, at least enough waves to form the row products in parallel
    unsigned threads = std::max<unsigned>((n2 + 63) / 64 * 64, std::min<unsigned>(1024, 64 * (unsigned)std::min<unsigned>(D2_NW, std::max(1u, n1))));
    GFT_LAUNCH(k_div_2d<E>, dim3(1), dim3(threads), lds, st, x, x_plane, y, y_plane, res, r_plane, g);
    return true;
}

template bool K<EF64>::div_2d(hipStream_t, const double*, size_t, unsigned, unsigned, size_t, const double*, size_t, unsigned, unsigned,
                              double*, size_t, unsigned, unsigned, int, unsigned, double*, size_t);
template bool K<EIv>::div_2d(hipStream_t, const double*, size_t, unsigned, unsigned, size_t, const double*, size_t, unsigned, unsigned,
                             double*, size_t, unsigned, unsigned, int, unsigned, double*, size_t);


// ------------------------------------------------------------------------------------------
// Rank-2 quotients / logarithms / exponentials with LONG rows (round 4): the coefficient-level wavefront
// ------------------------------------------------------------------------------------------
// k_div_wavefront keeps a row (<= 64 coefficients) in one wave.  Longer rows — 200^2, 400^2: what two-variable programs
// with a few hundred observations produce — ran as a chain of per-row launches (right-looking update + 1-d division),
// 400^2 div 38 ms = 0.4 % of the FP64 roof.  The dependency graph is finer than rows: coefficient c of row k0 needs the
// coefficients <= c of the rows above it (mt:1162-1192 over mt:971-982) and the coefficients < c of its own row (the 1-d
// division).  So a TASK is one 64-coefficient SEGMENT (k0, s) of a row, claimed in row-major order — every task comes
// after the tasks it reads, so a claimed task only ever waits for tasks claimed earlier by running workgroups — and
// row k0 + 1 divides its segment s while row k0 is still busy with segment s + 1: the critical path is n0 + nseg
// segment steps instead of n0 whole-row divisions.
//   * sum over the rows above: source row j0 contributes  sum_{j <= c} A[j] * B[c - j]  formed from zero in ascending j
//     (mul_1d), the sums added in ascending j0: the waves of the workgroup take the source rows of a batch side by side, wave 0
//     adds the batch's sums in order (k_div_wavefront's scheme).  A segment's sum runs over the 64-coefficient chunks
//     t = 0 .. s of A against the two chunks of B that slide past it ({A_t}{B_{s-t-1}}{B_{s-t}} in the wave's LDS stage, the
//     layout of row_product with the previous chunk where that one stages zeros): same additions in the same order;
//   * the row's own division: cur[c] = sum_{j < c} q[j] * y0[c - j] — first over the chunks of q that earlier segments of
//     this row have published (wave 0, same chunk product), then in lock step inside the segment (lane jj's coefficient is
//     final at step jj and reaches the others by v_readlane; y0[l - jj] does not depend on the segment);
//   * publication as in k_div_wavefront: result rows start out as the EMPTY pattern, every coefficient is one coherent
//     8-byte store, per-segment release / acquire flags are the authority for a chunk that keeps looking unwritten.
// mode 0: res = xs / ys.  mode 1: rows k0 >= 1 of log(xs) — S = sum_{j0 >= 1} xs[k0 - j0] (*) (j0 res[j0]),
// r = (-S) + k0 xs[k0], q = r / xs[0] (1-d), res[k0] = q / k0 (q kept in `qb`: later segments of the row divide with it).
// mode 2: rows k0 >= 1 of exp(xs) — res[k0] = (sum_{j0 >= 1} (j0 xs[j0]) (*) res[k0 - j0]) / k0.  Row 0 of log / exp is
// the caller's (a 1-d log / exp, complete in stream order).  Same operations per coefficient in the same order as the
// host-driven recursion => the same bits.
// exp's sum starts with j0 = 1, i.e. with the row finished LAST (res[k0 - 1]): in the reference's order every row waits for
// its predecessor before it can add anything, and the wavefront degenerates into a chain of whole rows (400^2: 116 ms).
// `rev` takes the source rows in descending j0 — the oldest result row first, the order in which they become available,
// the right-looking tiled form's order of arrival — under that form's contract (1e-10, measured ~1e-15: all terms of an
// exponential's recurrence carry the same sign pattern as the series itself); the caller sets it exactly where it would
// otherwise take the right-looking tiled form (f64, `exp_right`).
struct RowsWfArgs {
    unsigned n0, nr, m0, mr, xn0, xnr;   // rows / row lengths of res, ys (mode 0), xs
    unsigned nseg, ntasks, first_row;
    int mode;
    unsigned* flags;                     // [n0 * nseg] segment stored; zeroed before the launch
    unsigned* counter;                   // next task; zeroed before the launch
    double* qb;                          // mode 1: the quotient rows before the division by k0
    size_t qbp;
    int rev;                             // mode 2: take the source rows in DESCENDING j0 (see k_rows_wavefront)
};

// inner += sum_{i < 64} a[i] * bwin[64 + l - i]   (bwin = {bprev[64], bcur[64]}), ascending i — one chunk of a row product.
// jbase = 64 t (the chunk's first j), c = the lane's coefficient, alen / blen = the rows' lengths: positions outside the
// sum hold zeros (finite rows: inner + x * 0 == inner, a sum formed from +0 is never -0); non-finite rows apply the
// bounds as predicates.
template <class E>
__device__ inline typename E::V chunk_mac(typename E::V inner, typename E::V a_l, typename E::V bp_l, typename E::V bc_l, unsigned l,
                                          double* stage, unsigned jbase, unsigned c, unsigned alen, unsigned blen) {
    typedef typename E::V V;
    // (A's coefficient as a broadcast LDS read: v_readlane pairs instead — a scalar operand of the multiply — were measured
    // slower, 400^2 div 7.3 -> 8.4 ms)
    E::st(stage, 192, l, a_l);
    E::st(stage, 192, 64 + l, bp_l);
    E::st(stage, 192, 128 + l, bc_l);
    const double* bl = stage + 128 + l;
    if (!any_lane(!elem_finite<E>(a_l) || !elem_finite<E>(bp_l) || !elem_finite<E>(bc_l))) {
#pragma unroll 8
        for (unsigned i = 0; i < 64; ++i) inner = E::add(inner, E::mul(E::ld(stage, 192, i), E::ld(bl - i, 192, 0)));
    } else {
        for (unsigned i = 0; i < 64; ++i) {
            const V t = E::add(inner, E::mul(E::ld(stage, 192, i), E::ld(bl - i, 192, 0)));
            const unsigned j = jbase + i;
            if (j <= c && j < alen && c - j < blen) inner = t;
        }
    }
    return inner;
}

template <class E>
__global__ void __launch_bounds__(64 * DwfCfg<E>::NW) k_rows_wavefront(const double* __restrict__ xs, size_t xp, const double* __restrict__ ys, size_t yp,
                                                                 double* res, size_t rp, RowsWfArgs g) {
    typedef typename E::V V;
    constexpr unsigned NW = DwfCfg<E>::NW;
    __shared__ double part[2][E::W][NW][64];
    __shared__ double stage[NW][E::W][192];
    __shared__ double cur_l[E::W][64];
    __shared__ unsigned s_task;
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* const my_stage = &stage[wave][0][0];
    const bool lg = g.mode == 1, ex = g.mode == 2;
    // div / log: the last wave takes no source rows — during the LAST batch (the one that holds the row finished last) it forms
    // the part of the 1-d division's running sum that comes from this row's earlier segments, so that after the last source
    // only one chunk product, the batch's sum and the 64 lock steps remain on the critical path
    const unsigned NS = ex ? NW : NW - 1u;
    // the divisor's first row (the 1-d divisions divide by it): ys[0] or, for log, xs[0]
    const double* const d0 = lg ? xs : ys;
    const size_t d0p = lg ? xp : yp;
    const unsigned d0len = lg ? g.xnr : g.mr;
    const V y0row = (!ex && lane < d0len) ? E::ld(d0, d0p, lane) : E::zero();
    const SlabDiv<E> div_y00(ex ? E::one() : E::ld(d0, d0p, 0));
    // a chunk of a row that other workgroups of this launch produce: one coherent load, polled while it shows the EMPTY
    // pattern (the segment's flag is the authority after a while); `confirmed`: the row was complete before the launch
    auto coherent_raw = [&](const double* base, size_t plane, unsigned row, unsigned chunk, unsigned len) -> V {
        const unsigned idx = 64u * chunk + lane;
        return idx < len ? ld_coherent<E>(base, plane, (size_t)row * g.nr + idx) : E::zero();
    };
    auto coherent_confirm = [&](V v, const double* base, size_t plane, unsigned row, unsigned chunk, unsigned len, bool confirmed) -> V {
        const unsigned idx = 64u * chunk + lane;
        const bool in = idx < len;
        const size_t off = (size_t)row * g.nr + idx;
        for (unsigned spins = 1; !confirmed && any_lane(in && is_empty_bits(v)); ++spins) {
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 31u) == 0u && __hip_atomic_load(g.flags + (size_t)row * g.nseg + chunk, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                confirmed = true;  // (what the reload below returns is the chunk)
            v = in ? ld_coherent<E>(base, plane, off) : E::zero();
        }
        return v;
    };
    auto plain_chunk = [&](const double* base, size_t plane, size_t rowoff, int chunk, unsigned len) -> V {
        if (chunk < 0) return E::zero();
        const unsigned idx = 64u * (unsigned)chunk + lane;
        return idx < len ? E::ld(base, plane, rowoff + idx) : E::zero();
    };
    // cur[c] = sum_{j < 64 s} q[j] * y0[c - j], ascending j: the chunks of this row's quotient that its earlier segments have
    // published against the divisor's first row (chunk t meets y0 at 64 (s - t) - 63 .. 64 (s - t) + 63)
    auto division_prefix = [&](unsigned k0, unsigned s, unsigned c) -> V {
        const double* const qbase = lg ? g.qb : res;
        const size_t qplane = lg ? g.qbp : rp;
        V cur1 = E::zero();
        const unsigned t_lo = s > (d0len + 62u) / 64u ? s - (d0len + 62u) / 64u : 0u;
        if (t_lo < s) {
            V bc = plain_chunk(d0, d0p, 0, (int)s - (int)t_lo, d0len);
            V a_n = coherent_raw(qbase, qplane, k0, t_lo, g.nr);
            for (unsigned tt = t_lo; tt < s; ++tt) {
                const V a = coherent_confirm(a_n, qbase, qplane, k0, tt, g.nr, false);
                const V bp = plain_chunk(d0, d0p, 0, (int)s - (int)tt - 1, d0len);
                if (tt + 1 < s) a_n = coherent_raw(qbase, qplane, k0, tt + 1, g.nr);
                cur1 = chunk_mac<E>(cur1, a, bp, bc, lane, my_stage, 64u * tt, c, g.nr, d0len);
                bc = bp;
            }
        }
        return cur1;
    };
    for (;;) {
        if (threadIdx.x == 0) s_task = atomicAdd(g.counter, 1u);
        __syncthreads();
        const unsigned t = s_task;
        __syncthreads();  // everyone has read s_task before the next claim overwrites it
        if (t >= g.ntasks) break;
        const unsigned k0 = g.first_row + t / g.nseg, s = t % g.nseg;
        const unsigned c = 64u * s + lane;
        // the source rows j0 of this row's sum, in the reference's order (ascending)
        unsigned jlo, jhi;
        if (ex) {
            jlo = 1;
            jhi = (k0 < g.xn0 ? k0 : g.xn0 - 1) + 1;
        } else if (lg) {
            jlo = k0 + 1 > g.xn0 ? k0 + 1 - g.xn0 : 0;
            if (jlo < 1) jlo = 1;
            jhi = k0;
        } else {
            jlo = k0 + 1 > g.m0 ? k0 + 1 - g.m0 : 0;
            jhi = k0;
        }
        const unsigned cnt = jhi > jlo ? jhi - jlo : 0;
        V S = E::zero();
        unsigned buf = 0;
        for (unsigned base = 0; base < cnt; base += NS) {
            const unsigned i = base + wave;
            if (!ex && wave == NW - 1u) {
                if (base + NS >= cnt) E::st(&cur_l[0][0], 64, lane, division_prefix(k0, s, c));
            } else if (i < cnt) {
                const unsigned j0 = g.rev ? jhi - 1u - i : jlo + i;
                // A = the operand whose coefficient j is broadcast, B = the one read at c - j
                //   div: A = res[j0] (produced here), B = ys[k0 - j0]
                //   log: A = xs[k0 - j0],              B = j0 * res[j0] (produced here)
                //   exp: A = j0 * xs[j0],              B = res[k0 - j0] (produced here; row 0 is the caller's)
                const unsigned alen = (lg || ex) ? g.xnr : g.nr, blen = (lg || ex) ? g.nr : g.mr;
                const unsigned coh_row = ex ? k0 - j0 : j0;
                const size_t plain_off = ex ? (size_t)j0 * g.xnr : (lg ? (size_t)(k0 - j0) * g.xnr : (size_t)(k0 - j0) * g.mr);
                const V scale = E::from_u32(j0);
                const bool confirmed = ex && coh_row == 0;
                // raw_*: the loads; fix_*: wait for the producer where the chunk is one of this launch's results, then scale
                auto raw_a = [&](unsigned tt) -> V {
                    if (!lg && !ex) return coherent_raw(res, rp, coh_row, tt, g.nr);
                    return plain_chunk(xs, xp, plain_off, (int)tt, g.xnr);
                };
                auto fix_a = [&](V v, unsigned tt) -> V {
                    if (!lg && !ex) return coherent_confirm(v, res, rp, coh_row, tt, g.nr, false);
                    if (ex && 64u * tt + lane < g.xnr) v = E::mul(v, scale);
                    return v;
                };
                auto raw_b = [&](int ch) -> V {
                    if (ch < 0) return E::zero();
                    if (!lg && !ex) return plain_chunk(ys, yp, plain_off, ch, g.mr);
                    return coherent_raw(res, rp, coh_row, (unsigned)ch, g.nr);
                };
                auto fix_b = [&](V v, int ch) -> V {
                    if (ch < 0 || (!lg && !ex)) return v;
                    v = coherent_confirm(v, res, rp, coh_row, (unsigned)ch, g.nr, confirmed);
                    if (lg && 64u * (unsigned)ch + lane < g.nr) v = E::mul(v, scale);
                    return v;
                };
                // chunks t of A with any coefficient (j < alen) whose partner B[c - j] can exist (c - j < blen for some c of the segment)
                // (chunk t meets B at the indices 64 (s - t) - 63 .. 64 (s - t) + 63: below t_lo every one of them is >= blen — terms
                // the reference's sum does not have)
                const unsigned t_hi = (alen + 63u) / 64u - 1u < s ? (alen + 63u) / 64u - 1u : s;
                const unsigned t_lo = s > (blen + 62u) / 64u ? s - (blen + 62u) / 64u : 0u;
                V inner = E::zero();
                if (t_lo <= t_hi) {
                    // the next chunk's loads are requested before this chunk's 64 multiply-add steps (confirmed — polled if
                    // their producer has not stored them yet — when they are needed)
                    V bc = fix_b(raw_b((int)s - (int)t_lo), (int)s - (int)t_lo);
                    V a_n = raw_a(t_lo), bp_n = raw_b((int)s - (int)t_lo - 1);
                    for (unsigned tt = t_lo; tt <= t_hi; ++tt) {
                        const V a = fix_a(a_n, tt);
                        const V bp = fix_b(bp_n, (int)s - (int)tt - 1);
                        if (tt < t_hi) {
                            a_n = raw_a(tt + 1);
                            bp_n = raw_b((int)s - (int)tt - 2);
                        }
                        inner = chunk_mac<E>(inner, a, bp, bc, lane, my_stage, 64u * tt, c, alen, blen);
                        bc = bp;  // the next chunk of A meets this one as its upper window
                    }
                }
                E::st(&part[buf][0][wave][0], (size_t)NW * 64, lane, inner);
            }
            __syncthreads();
            if (wave == 0) {
                const unsigned nb = cnt - base < NS ? cnt - base : NS;
                const double* pb = &part[buf][0][0][0];
                if (nb == NW - 1u) {
                    V pv[NW - 1];
#pragma unroll
                    for (unsigned w = 0; w < NW - 1; ++w) pv[w] = E::ld(pb, (size_t)NW * 64, (size_t)w * 64 + lane);
#pragma unroll
                    for (unsigned w = 0; w < NW - 1; ++w) S = E::add(S, pv[w]);
                } else if (nb == NW) {
                    V pv[NW];
#pragma unroll
                    for (unsigned w = 0; w < NW; ++w) pv[w] = E::ld(pb, (size_t)NW * 64, (size_t)w * 64 + lane);
#pragma unroll
                    for (unsigned w = 0; w < NW; ++w) S = E::add(S, pv[w]);
                } else {
                    for (unsigned w = 0; w < nb; ++w) S = E::add(S, E::ld(pb, (size_t)NW * 64, (size_t)w * 64 + lane));
                }
            }
            buf ^= 1u;  // the next batch writes the other buffer while wave 0 still reads this one
        }
        if (wave == 0) {
            const size_t qoff = (size_t)k0 * g.nr + c;
            if (ex) {
                if (c < g.nr) st_coherent(res, rp, qoff, E::div(S, E::from_u32(k0)));  // mt:1298
            } else {
                V r = E::neg(S);
                if (k0 < g.xn0 && c < g.xnr) {
                    const V xin = E::ld(xs, xp, (size_t)k0 * g.xnr + c);
                    r = E::add(r, lg ? E::mul(E::from_u32(k0), xin) : xin);
                }
                if (c >= g.nr) r = E::zero();
                // the 1-d division (mt:1162-1185): cur[c] = sum_{j < c} q[j] * y0[c - j], ascending j — the chunks of q the
                // earlier segments of this row have published, then lock step inside the segment
                // (its part from this row's earlier segments: formed by the last wave during the last batch, or here if the row
                // has no source rows at all)
                V cur1 = cnt ? E::ld(&cur_l[0][0], 64, lane) : division_prefix(k0, s, c);
                V mine = E::zero(), ysl = y0row;  // ysl[l] = y0[l - jj] at step jj (zero for l < jj and beyond the divisor's row)
                const bool fin = !any_lane(!elem_finite<E>(r)) && !any_lane(!elem_finite<E>(y0row)) && !any_lane(!elem_finite<E>(cur1));
                const unsigned nsteps = g.nr - 64u * s < 64u ? g.nr - 64u * s : 64u;
                for (unsigned jj = 0; jj < nsteps; ++jj) {
                    const V q = div_y00(bcast_lane<E>(E::add(E::neg(cur1), r), jj));
                    if (lane == jj) mine = q;
                    const V tnew = E::add(cur1, E::mul(q, ysl));
                    if (fin && elem_finite<E>(q)) {
                        cur1 = tnew;  // positions outside the sum multiply a shifted-in zero
                    } else if (lane > jj && lane - jj < d0len) {
                        cur1 = tnew;
                    }
                    ysl = wave_shr1<E>(ysl);
                }
                if (c < g.nr) {
                    if (lg) {  // res[k0] = q / k0 (mt:1384); the row's later segments divide with q itself
                        st_coherent(g.qb, g.qbp, qoff, mine);
                        st_coherent(res, rp, qoff, E::div(mine, E::from_u32(k0)));
                    } else {
                        st_coherent(res, rp, qoff, mine);
                    }
                }
            }
            __threadfence();  // the segment is visible device-wide before its flag is
            if (lane == 0) __hip_atomic_store(g.flags + (size_t)k0 * g.nseg + s, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// mode 0: res = xs / ys;  1: rows >= 1 of log(xs) (qbuf: a tensor like res);  2: rows >= 1 of exp(xs).  Rank 2, rows of
// 65 .. 4096 coefficients.  `flags_and_counter`: n0 * ceil(nr / 64) + 1 zeroed words.  false: outside the kernel's domain,
// nothing launched.
template <class E>
bool K<E>::rows_wavefront(hipStream_t st, int mode, const double* xs, size_t x_plane, const unsigned* xshape, const double* ys, size_t y_plane,
                          const unsigned* yshape, double* res, size_t r_plane, const unsigned* rshape, double* qbuf, size_t q_plane,
                          unsigned* flags_and_counter) {
    RowsWfArgs g;
    std::memset(&g, 0, sizeof(g));
    g.rev = (mode & 4) ? 1 : 0;  // mode 2 | 4: exp with the source rows in descending order (1e-10 contract)
    mode &= 3;
    g.mode = mode;
    g.n0 = rshape[0];
    g.nr = rshape[1];
    g.xn0 = xshape[0];
    g.xnr = xshape[1];
    g.m0 = mode == 0 ? yshape[0] : xshape[0];
    g.mr = mode == 0 ? yshape[1] : xshape[1];
    if (g.nr <= 64 || g.nr > 4096 || g.n0 == 0 || g.xn0 == 0 || g.xnr == 0 || g.m0 == 0 || g.mr == 0) return false;
    if (g.xn0 > g.n0 || g.xnr > g.nr || g.m0 > g.n0 || g.mr > g.nr) return false;
    g.first_row = mode == 0 ? 0u : 1u;
    if (g.n0 <= g.first_row) return false;
    g.nseg = (g.nr + 63u) / 64u;
    const size_t ntasks = (size_t)(g.n0 - g.first_row) * g.nseg;
    if (ntasks > 0x7fffffffu) return false;
    g.ntasks = (unsigned)ntasks;
    g.flags = flags_and_counter;
    g.counter = flags_and_counter + (size_t)g.n0 * g.nseg;
    g.qb = qbuf;
    g.qbp = q_plane;
    if (mode == 1 && !qbuf) return false;
    const size_t skip = (size_t)g.first_row * g.nr, nel = (size_t)(g.n0 - g.first_row) * g.nr;
    for (int pl = 0; pl < E::W; ++pl) {
        const unsigned fb = (unsigned)std::min<size_t>((nel + 255) / 256, 2048);
        GFT_LAUNCH(k_fill_bits, dim3(fb), dim3(256), 0, st, res + (size_t)pl * r_plane + skip, nel, DWF_EMPTY);
        if (mode == 1) GFT_LAUNCH(k_fill_bits, dim3(fb), dim3(256), 0, st, qbuf + (size_t)pl * q_plane + skip, nel, DWF_EMPTY);
    }
    // persistent workgroups (they claim tasks until none is left): two per CU so that the SIMDs stay busy while some wait
    const unsigned blocks = (unsigned)std::min<size_t>(ntasks, (size_t)256 * 2);
    GFT_LAUNCH((k_rows_wavefront<E>), dim3(blocks), dim3(64 * DwfCfg<E>::NW), 0, st, xs, x_plane, mode == 0 ? ys : xs, mode == 0 ? y_plane : x_plane,
               res, r_plane, g);
    return true;
}
template bool K<EF64>::rows_wavefront(hipStream_t, int, const double*, size_t, const unsigned*, const double*, size_t, const unsigned*, double*, size_t,
                                      const unsigned*, double*, size_t, unsigned*);
template bool K<EIv>::rows_wavefront(hipStream_t, int, const double*, size_t, const unsigned*, const double*, size_t, const unsigned*, double*, size_t,
                                     const unsigned*, double*, size_t, unsigned*);

// ------------------------------------------------------------------------------------------
// Quotients / logarithms of rank 3 and 4 with LONG rows (round 6): the segment wavefront with leading axes
// ------------------------------------------------------------------------------------------
// k_div_wavefront takes rows of at most 64 coefficients, k_rows_wavefront rank 2 only: a rank-3 quotient with rows of 65 or
// more fell back to the slab-by-slab blocked form of round 2 — 96^3 div 404 ms against 4.0 ms at 64^3, a 30x cliff at row 65.
// Here the two are put together: a TASK is one 64-coefficient segment (K, s) of the quotient row K = (k_0 .. k_{L-1}); it
// consumes its source rows level by level in the reference's order exactly as k_div_wavefront does (mt:1162-1192 over
// mt:984-1012; see the comment there), but every source row contributes only the chunks of its row product that reach the
// segment — chunk_mac over t = 0 .. s, k_rows_wavefront's scheme: the same additions in the same order — and the row's 1-d
// division first takes the chunks its own earlier segments have published, then runs in lock step inside the segment.
// Tasks are claimed in (row order, s) order: a task's sources — segments <= s of rows earlier in the row order, segments
// < s of its own row — are all claimed before it.  Publication: EMPTY pattern + coherent 8-byte stores + per-segment flags.
// log_mode 1: slabs k0 >= 1 of log(xs) (level 0: xs[k0 - j0] rows against j0 * res[j0] rows; levels >= 1 and the division:
// by xs[0], on the slab's own quotient rows kept in `qb`; res = q / k0) — slab 0 is the caller's.
// chunk_mac for segments of SL <= 64 coefficients: inner += sum_{i < SL} a[i] * bwin[SL + l - i]  (bwin = {bprev[SL], bcur[SL]}),
// ascending i; lanes >= SL idle (they stage nothing and their sums are discarded).
template <class E>
__device__ inline typename E::V seg_chunk_mac(typename E::V inner, typename E::V a_l, typename E::V bp_l, typename E::V bc_l, unsigned l,
                                              double* stage, unsigned SL, unsigned jbase, unsigned c, unsigned alen, unsigned blen) {
    typedef typename E::V V;
    if (l < SL) {
        E::st(stage, 192, l, a_l);
        E::st(stage, 192, SL + l, bp_l);
        E::st(stage, 192, 2u * SL + l, bc_l);
    }
    const double* bl = stage + 2u * SL + (l < SL ? l : 0u);
    if (!any_lane(!elem_finite<E>(a_l) || !elem_finite<E>(bp_l) || !elem_finite<E>(bc_l))) {
#pragma unroll 4
        for (unsigned i = 0; i < SL; ++i) inner = E::add(inner, E::mul(E::ld(stage, 192, i), E::ld(bl - i, 192, 0)));
    } else {
        for (unsigned i = 0; i < SL; ++i) {
            const V t = E::add(inner, E::mul(E::ld(stage, 192, i), E::ld(bl - i, 192, 0)));
            const unsigned j = jbase + i;
            if (j <= c && j < alen && c - j < blen) inner = t;
        }
    }
    return inner;
}
// The same chunk product with the BROADCAST operand read by scalar loads instead of from LDS (round 6: the segment wavefront is
// LDS-bound — two 8-byte reads per multiply-add; 96^3 div 50 -> 38 ms in an experiment without the broadcast one).  A scalar load
// needs memory nobody writes during the launch, i.e. a PLAIN operand — and the broadcast operand is A, the one whose index
// ascends: that is xs's row at log's level 0 (where most of log's multiply-adds are: 96^3 log 51 -> 43 ms), but a result row other
// workgroups publish everywhere else (div's every level).  Measured and not kept for those: broadcasting the divisor's chunks in
// DESCENDING order against a window over the result row (the same terms in the same order, bit-exact) — 50 -> 57 ms at 96^3, with
// the broadcast from SGPRs or from LDS alike: the loop over descending blocks costs more than the LDS read it saves.
// `n` = elements of the broadcast chunk that exist (beyond them the staged form multiplies by zero: no term).  The staged copy of
// the broadcast chunk (stage[0 .. SL)) serves the finiteness test and the masked, non-finite path.
typedef const double __attribute__((address_space(4))) * seg_cptr_t;
// a wave-uniform 64-bit value the compiler cannot see to be one (it came through LDS: the claimed task): into SGPRs
__device__ __forceinline__ size_t seg_uniform(size_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((size_t)hi << 32) | lo;
}
template <class E>
__device__ __forceinline__ typename E::V seg_sload(seg_cptr_t p, size_t plane, unsigned i) {
    if constexpr (E::W == 2) return Iv{p[i], p[plane + i]};
    else return p[i];
}
template <class E>
__device__ inline typename E::V seg_chunk_mac_sa(typename E::V inner, typename E::V a_l, typename E::V bp_l, typename E::V bc_l, unsigned l, double* stage,
                                                 unsigned SL, unsigned jbase, unsigned c, unsigned alen, unsigned blen, seg_cptr_t ap, size_t aplane, unsigned n) {
    typedef typename E::V V;
    if (l < SL) {
        E::st(stage, 192, l, a_l);
        E::st(stage, 192, SL + l, bp_l);
        E::st(stage, 192, 2u * SL + l, bc_l);
    }
    const double* bl = stage + 2u * SL + (l < SL ? l : 0u);
    if (!any_lane(!elem_finite<E>(a_l) || !elem_finite<E>(bp_l) || !elem_finite<E>(bc_l))) {
#pragma unroll 8
        for (unsigned i = 0; i < n; ++i) inner = E::add(inner, E::mul(seg_sload<E>(ap, aplane, i), E::ld(bl - i, 192, 0)));
    } else {
        for (unsigned i = 0; i < SL; ++i) {
            const V t = E::add(inner, E::mul(E::ld(stage, 192, i), E::ld(bl - i, 192, 0)));
            const unsigned j = jbase + i;
            if (j <= c && j < alen && c - j < blen) inner = t;
        }
    }
    return inner;
}
struct SegWfArgs {
    int L;
    unsigned n[3], m[3], xn[3];
    unsigned nr, mr, xnr;
    size_t rstr[3], ystr[3], xstr[3];
    unsigned nseg, sl, ntasks;    // segments per row, coefficients per segment (<= 64: rows are cut EVENLY — a 65-long row is 33 + 32, not 64 + 1)
    unsigned* flags;              // [all rows of res][nseg], zeroed before the launch
    unsigned* counter;
    int log_mode;
    const unsigned* order;        // task row t works on row order[t] of the task rows (dwf_order); null: t
    double* qb;
    size_t qbp;
};
template <class E, int L>
__global__ void __launch_bounds__(64 * DwfCfg<E>::NW) k_seg_wavefront(const double* __restrict__ xs, size_t xp, const double* __restrict__ ys, size_t yp,
                                                               double* res, size_t rp, SegWfArgs g) {
    typedef typename E::V V;
    constexpr unsigned NW = DwfCfg<E>::NW;
    __shared__ double part[2][E::W][NW][64];
    __shared__ double stage[NW][E::W][192];
    __shared__ unsigned s_task;
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* const my_stage = &stage[wave][0][0];
    const bool lg = g.log_mode == 1;
    const unsigned SL = g.sl;
    // the divisor's first row: ys[0 .. 0, :] (log: xs[0 .. 0, :] — the caller passes xs as ys)
    const V y0row = lane < g.mr ? E::ld(ys, yp, lane) : E::zero();
    const SlabDiv<E> div_y00(E::ld(ys, yp, 0));
    auto plain_chunk = [&](const double* base, size_t plane, size_t rowoff, int chunk, unsigned len) -> V {
        if (chunk < 0 || lane >= SL) return E::zero();
        const unsigned idx = SL * (unsigned)chunk + lane;
        return idx < len ? E::ld(base, plane, rowoff + idx) : E::zero();
    };
    auto coherent_raw = [&](const double* base, size_t plane, size_t rowoff, int chunk) -> V {
        if (chunk < 0 || lane >= SL) return E::zero();
        const unsigned idx = SL * (unsigned)chunk + lane;
        return idx < g.nr ? ld_coherent<E>(base, plane, rowoff + idx) : E::zero();
    };
    auto coherent_confirm = [&](V v, const double* base, size_t plane, size_t rowoff, unsigned row_id, int chunk) -> V {
        if (chunk < 0) return v;
        const unsigned idx = SL * (unsigned)chunk + lane;
        const bool in = lane < SL && idx < g.nr;
        bool confirmed = false;
        for (unsigned spins = 1; !confirmed && any_lane(in && is_empty_bits(v)); ++spins) {
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 31u) == 0u && __hip_atomic_load(g.flags + (size_t)row_id * g.nseg + (unsigned)chunk, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                confirmed = true;  // (what the reload below returns is the chunk)
            v = in ? ld_coherent<E>(base, plane, rowoff + idx) : E::zero();
        }
        return v;
    };
    for (;;) {
        if (threadIdx.x == 0) s_task = atomicAdd(g.counter, 1u);
        __syncthreads();
        const unsigned t = s_task;
        __syncthreads();
        if (t >= g.ntasks) break;
        const unsigned s = t % g.nseg;
        const unsigned c = SL * s + lane;  // (lanes >= SL idle: their values are zero and never stored)
        unsigned k[3] = {0, 0, 0};
        unsigned row_id = 0;
        size_t qoff_row = 0;
        {
            unsigned r = g.order ? g.order[t / g.nseg] : t / g.nseg;
#pragma unroll
            for (int a = L - 1; a >= 1; --a) {
                k[a] = r % g.n[a];
                r /= g.n[a];
            }
            k[0] = r + (lg ? 1u : 0u);
#pragma unroll
            for (int a = 0; a < L; ++a) {
                row_id = row_id * g.n[a] + k[a];
                qoff_row += (size_t)k[a] * g.rstr[a];
            }
        }
        V r_prev = E::zero();  // (meaningful in wave 0)
        unsigned buf = 0;
        auto level = [&](auto lev_c) {
            constexpr int lev = decltype(lev_c)::value;
            V S = E::zero();
            const bool lg0 = lg && lev == 0;  // the other operand's rows come from xs, the odometer runs over ITS rows
            unsigned lo[3] = {0, 0, 0}, cnt[3] = {1, 1, 1};
            unsigned total = 1;
#pragma unroll
            for (int a = lev; a < L; ++a) {
                if (lg0 && a > 0) {
                    lo[a] = 0;
                    cnt[a] = (k[a] < g.xn[a] ? k[a] : g.xn[a] - 1) + 1;
                } else {
                    const unsigned mm = lg0 ? g.xn[a] : g.m[a];
                    lo[a] = k[a] + 1 > mm ? k[a] + 1 - mm : 0;
                    if (lg0 && lo[a] < 1) lo[a] = 1;
                    const unsigned hi = a == lev ? k[a] : k[a] + 1;  // exclusive
                    cnt[a] = hi > lo[a] ? hi - lo[a] : 0;
                }
                total *= cnt[a];
            }
            const double* const coh_base = (lg && lev > 0) ? g.qb : res;  // where the result-side rows live
            const size_t coh_plane = (lg && lev > 0) ? g.qbp : rp;
            for (unsigned base = 0; base < total; base += NW) {
                const unsigned i = base + wave;
                if (i < total) {
                    unsigned rem = i, j[3] = {0, 0, 0};
#pragma unroll
                    for (int a = L - 1; a >= lev; --a) {
                        j[a] = rem % cnt[a];
                        rem /= cnt[a];
                    }
#pragma unroll
                    for (int a = lev; a < L; ++a) j[a] += lo[a];
                    size_t roff = 0, ooff = 0;
                    unsigned src = 0;
#pragma unroll
                    for (int a = 0; a < L; ++a) {
                        unsigned ra;  // the result row's index on this axis
                        if (a < lev) ra = k[a];
                        else if (lg0 && a > 0) ra = k[a] - j[a];
                        else ra = j[a];
                        roff += (size_t)ra * g.rstr[a];
                        src = src * g.n[a] + ra;
                        if (a >= lev) {
                            if (lg0) ooff += (size_t)(a == 0 ? k[0] - j[0] : j[a]) * g.xstr[a];
                            else ooff += (size_t)(k[a] - j[a]) * g.ystr[a];
                        }
                    }
                    const V scale = E::from_u32(j[0]);
                    // A = the operand whose coefficient j is broadcast, B = the one read at c - j
                    //   div, log levels >= 1: A = result row (coherent), B = divisor row (plain)
                    //   log level 0:          A = xs row (plain),        B = j0 * result row (coherent)
                    const unsigned alen = lg0 ? g.xnr : g.nr, blen = lg0 ? g.nr : g.mr;
                    auto raw_a = [&](unsigned tt) -> V { return lg0 ? plain_chunk(xs, xp, ooff, (int)tt, g.xnr) : coherent_raw(coh_base, coh_plane, roff, (int)tt); };
                    auto fix_a = [&](V v, unsigned tt) -> V { return lg0 ? v : coherent_confirm(v, coh_base, coh_plane, roff, src, (int)tt); };
                    auto raw_b = [&](int ch) -> V {
                        if (ch < 0) return E::zero();
                        return lg0 ? coherent_raw(coh_base, coh_plane, roff, ch) : plain_chunk(ys, yp, ooff, ch, g.mr);
                    };
                    auto fix_b = [&](V v, int ch) -> V {
                        if (ch < 0 || !lg0) return v;
                        v = coherent_confirm(v, coh_base, coh_plane, roff, src, ch);
                        if (lane < SL && SL * (unsigned)ch + lane < g.nr) v = E::mul(v, scale);
                        return v;
                    };
                    const unsigned t_hi = (alen + SL - 1u) / SL - 1u < s ? (alen + SL - 1u) / SL - 1u : s;
                    const unsigned t_lo = s > (blen + SL - 2u) / SL ? s - (blen + SL - 2u) / SL : 0u;
                    V inner = E::zero();
                    if (lg0) {
                        // log level 0: A (xs's row) is the plain operand: broadcast by scalar loads, ascending as before
                        if (t_lo <= t_hi) {
                            const seg_cptr_t xrow = (seg_cptr_t)(xs + seg_uniform(ooff));
                            V bc = fix_b(raw_b((int)s - (int)t_lo), (int)s - (int)t_lo);
                            V a_n = raw_a(t_lo), bp_n = raw_b((int)s - (int)t_lo - 1);
                            for (unsigned tt = t_lo; tt <= t_hi; ++tt) {
                                const V a = fix_a(a_n, tt);
                                const V bp = fix_b(bp_n, (int)s - (int)tt - 1);
                                if (tt < t_hi) {
                                    a_n = raw_a(tt + 1);
                                    bp_n = raw_b((int)s - (int)tt - 2);
                                }
                                const unsigned na = alen - SL * tt < SL ? alen - SL * tt : SL;
                                inner = seg_chunk_mac_sa<E>(inner, a, bp, bc, lane, my_stage, SL, SL * tt, c, alen, blen, xrow + SL * tt, xp, na);
                                bc = bp;
                            }
                        }
                    } else if (t_lo <= t_hi) {
                        V bc = fix_b(raw_b((int)s - (int)t_lo), (int)s - (int)t_lo);
                        V a_n = raw_a(t_lo), bp_n = raw_b((int)s - (int)t_lo - 1);
                        for (unsigned tt = t_lo; tt <= t_hi; ++tt) {
                            const V a = fix_a(a_n, tt);
                            const V bp = fix_b(bp_n, (int)s - (int)tt - 1);
                            if (tt < t_hi) {
                                a_n = raw_a(tt + 1);
                                bp_n = raw_b((int)s - (int)tt - 2);
                            }
                            inner = seg_chunk_mac<E>(inner, a, bp, bc, lane, my_stage, SL, SL * tt, c, alen, blen);
                            bc = bp;
                        }
                    }
                    E::st(&part[buf][0][wave][0], (size_t)NW * 64, lane, inner);
                }
                __syncthreads();
                if (wave == 0) {
                    const unsigned nb = total - base < NW ? total - base : NW;
                    const double* pb = &part[buf][0][0][0];
                    if (nb == NW) {
                        V pv[NW];
#pragma unroll
                        for (unsigned w = 0; w < NW; ++w) pv[w] = E::ld(pb, (size_t)NW * 64, (size_t)w * 64 + lane);
#pragma unroll
                        for (unsigned w = 0; w < NW; ++w) S = E::add(S, pv[w]);
                    } else {
                        for (unsigned w = 0; w < nb; ++w) S = E::add(S, E::ld(pb, (size_t)NW * 64, (size_t)w * 64 + lane));
                    }
                }
                buf ^= 1u;
            }
            if (wave == 0) {
                if (lane >= SL) S = E::zero();
                V r = E::neg(S);
                if (lev == 0) {
                    bool in_x = lane < SL && c < g.xnr;
                    size_t xoff = 0;
#pragma unroll
                    for (int a = 0; a < L; ++a) {
                        if (k[a] >= g.xn[a]) in_x = false;
                        xoff += (size_t)k[a] * g.xstr[a];
                    }
                    if (in_x) {
                        const V xin = E::ld(xs, xp, xoff + c);
                        r = E::add(r, lg ? E::mul(E::from_u32(k[0]), xin) : xin);
                    }
                } else {
                    r = E::add(r, r_prev);
                }
                r_prev = r;
            }
            __syncthreads();  // a level's last batch buffer is free again before the next level reuses it
        };
        level(std::integral_constant<int, 0>{});
        if constexpr (L > 1) level(std::integral_constant<int, 1>{});
        if constexpr (L > 2) level(std::integral_constant<int, 2>{});
        if (wave == 0) {
            const double* const qbase = lg ? g.qb : res;
            const size_t qplane = lg ? g.qbp : rp;
            V r = r_prev;
            if (lane >= SL || c >= g.nr) r = E::zero();
            // the row's 1-d division (mt:1162-1185): cur[c] = sum_{j < c} q[j] * y0[c - j], ascending j — first the chunks of q the
            // earlier segments of this row have published, then lock step inside the segment
            V cur1 = E::zero();
            {
                const unsigned t_lo = s > (g.mr + SL - 2u) / SL ? s - (g.mr + SL - 2u) / SL : 0u;
                if (t_lo < s) {
                    V bc = plain_chunk(ys, yp, 0, (int)s - (int)t_lo, g.mr);
                    V a_n = coherent_raw(qbase, qplane, qoff_row, (int)t_lo);
                    for (unsigned tt = t_lo; tt < s; ++tt) {
                        const V a = coherent_confirm(a_n, qbase, qplane, qoff_row, row_id, (int)tt);
                        const V bp = plain_chunk(ys, yp, 0, (int)s - (int)tt - 1, g.mr);
                        if (tt + 1 < s) a_n = coherent_raw(qbase, qplane, qoff_row, (int)tt + 1);
                        cur1 = seg_chunk_mac<E>(cur1, a, bp, bc, lane, my_stage, SL, SL * tt, c, g.nr, g.mr);
                        bc = bp;
                    }
                }
            }
            V mine = E::zero(), ysl = y0row;  // ysl[l] = y0[l - jj] at step jj (zero for l < jj and beyond the divisor's row)
            const bool fin = !any_lane(!elem_finite<E>(r)) && !any_lane(!elem_finite<E>(y0row)) && !any_lane(!elem_finite<E>(cur1));
            if (lane >= SL) cur1 = E::zero();
            const unsigned nsteps = g.nr - SL * s < SL ? g.nr - SL * s : SL;
            for (unsigned jj = 0; jj < nsteps; ++jj) {
                const V q = div_y00(bcast_lane<E>(E::add(E::neg(cur1), r), jj));
                if (lane == jj) mine = q;
                const V tnew = E::add(cur1, E::mul(q, ysl));
                if (fin && elem_finite<E>(q)) {
                    cur1 = tnew;  // positions outside the sum multiply a shifted-in zero
                } else if (lane > jj && lane - jj < g.mr) {
                    cur1 = tnew;
                }
                ysl = wave_shr1<E>(ysl);
            }
            if (lane < SL && c < g.nr) {
                if (lg) {  // res[K] = q / k0 (mt:1384); the slab's own later rows and segments read q itself
                    st_coherent(g.qb, g.qbp, qoff_row + c, mine);
                    st_coherent(res, rp, qoff_row + c, E::div(mine, E::from_u32(k[0])));
                } else {
                    st_coherent(res, rp, qoff_row + c, mine);
                }
            }
            __threadfence();  // the segment is visible device-wide before its flag is
            if (lane == 0) __hip_atomic_store(g.flags + (size_t)row_id * g.nseg + s, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// mode 0: res = xs / ys; mode 1: slabs k0 >= 1 of res = log(xs) (slab 0 is the caller's; qbuf: a tensor like res).  Ranks 3 and 4,
// rows of 65 .. 4096 coefficients.  `flags_and_counter`: (rows of res) * ceil(row length / 64) + 1 zeroed words.  false: outside
// the kernel's domain, nothing launched.
template <class E>
bool K<E>::seg_wavefront(hipStream_t st, int mode, const double* xs, size_t x_plane, const unsigned* xshape, const double* ys, size_t y_plane,
                         const unsigned* yshape, double* res, size_t r_plane, const unsigned* rshape, int nd, double* qbuf, size_t q_plane,
                         unsigned* flags_and_counter) {
    if (nd < 3 || nd > 4 || (mode != 0 && mode != 1)) return false;
    SegWfArgs g;
    std::memset(&g, 0, sizeof(g));
    g.L = nd - 1;
    g.log_mode = mode;
    g.nr = rshape[nd - 1];
    g.xnr = xshape[nd - 1];
    g.mr = mode == 0 ? yshape[nd - 1] : g.xnr;
    if (g.nr <= 64 || g.nr > 4096 || g.mr == 0 || g.xnr == 0 || g.mr > g.nr || g.xnr > g.nr) return false;
    if (mode == 1 && (!qbuf || rshape[0] < 2)) return false;
    size_t rs = g.nr, ysd = g.mr, xsd = g.xnr, rows = 1;
    for (int a = g.L - 1; a >= 0; --a) {
        g.n[a] = rshape[a];
        g.xn[a] = xshape[a];
        g.m[a] = mode == 0 ? yshape[a] : xshape[a];
        if (g.m[a] > g.n[a] || g.xn[a] > g.n[a] || g.n[a] == 0 || g.m[a] == 0 || g.xn[a] == 0) return false;
        g.rstr[a] = rs;
        g.ystr[a] = ysd;
        g.xstr[a] = xsd;
        rs *= rshape[a];
        ysd *= g.m[a];
        xsd *= xshape[a];
        rows *= rshape[a];
    }
    g.nseg = (g.nr + 63u) / 64u;
    g.sl = (g.nr + g.nseg - 1u) / g.nseg;  // even cut: the work of a row is nseg (nseg + 1) / 2 chunk products of sl steps each
    const size_t slab_rows = rows / rshape[0];
    const size_t task_rows = mode == 1 ? rows - slab_rows : rows;
    const size_t ntasks = task_rows * g.nseg;
    if (rows * g.nseg > 0x7fffffffu || task_rows == 0) return false;
    g.ntasks = (unsigned)ntasks;
    g.flags = flags_and_counter;
    g.counter = flags_and_counter + rows * g.nseg;
    g.order = dwf_order(g.L, g.n, mode == 1 ? 1u : 0u);
    g.qb = qbuf;
    g.qbp = q_plane;
    const size_t skip = mode == 1 ? slab_rows * g.nr : 0, nel = task_rows * g.nr;
    for (int pl = 0; pl < E::W; ++pl) {
        const unsigned fb = (unsigned)std::min<size_t>((nel + 255) / 256, 2048);
        GFT_LAUNCH(k_fill_bits, dim3(fb), dim3(256), 0, st, res + (size_t)pl * r_plane + skip, nel, DWF_EMPTY);
        if (mode == 1) GFT_LAUNCH(k_fill_bits, dim3(fb), dim3(256), 0, st, qbuf + (size_t)pl * q_plane + skip, nel, DWF_EMPTY);
    }
    const dim3 grid((unsigned)std::min<size_t>(ntasks, (size_t)256 * 2)), block(64 * DwfCfg<E>::NW);
    const double* yarg = mode == 0 ? ys : xs;
    const size_t yplane = mode == 0 ? y_plane : x_plane;
    if (g.L == 2) GFT_LAUNCH((k_seg_wavefront<E, 2>), grid, block, 0, st, xs, x_plane, yarg, yplane, res, r_plane, g);
    else GFT_LAUNCH((k_seg_wavefront<E, 3>), grid, block, 0, st, xs, x_plane, yarg, yplane, res, r_plane, g);
    return true;
}
template bool K<EF64>::seg_wavefront(hipStream_t, int, const double*, size_t, const unsigned*, const double*, size_t, const unsigned*, double*, size_t,
                                     const unsigned*, int, double*, size_t, unsigned*);
template bool K<EIv>::seg_wavefront(hipStream_t, int, const double*, size_t, const unsigned*, const double*, size_t, const unsigned*, double*, size_t,
                                    const unsigned*, int, double*, size_t, unsigned*);


}  // namespace gft
