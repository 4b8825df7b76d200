// Element types of the Taylor tensors on the device: F64 (src/number/f64.rs) and
// Interval<F64> (src/interval.rs) with identical static interfaces, so every structural kernel
// is written once.  Interval tensors are stored as two planes (lo, hi) `plane` doubles apart.
//
// The whole library is compiled with -ffp-contract=off: the reference never fuses a*b+c and
// Interval widening assumes separately rounded operations.  FMA is used only where a kernel
// asks for it explicitly (the tiled f64 convolution).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdint>

// Every element operation is one source for both sides of the size-threshold dispatch: the device
// kernels and the host tier of gft_host.hpp call the same functions, so a tensor computed on either side carries the
// same bits (IEEE f64 + - * / and integer bit steps; -ffp-contract=off on both passes).
#define GFT_HD __host__ __device__

namespace gft {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a full workgroup fence: it also waits for every
// outstanding GLOBAL load and store (s_waitcnt vmcnt(0)), which drains software prefetches on each step of a latency
// chain.  Use where the threads of the block communicate through LDS alone across the barrier.
__device__ inline void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}


GFT_HD inline long long f64_bits(double x) { return __builtin_bit_cast(long long, x); }
GFT_HD inline double bits_f64(long long b) { return __builtin_bit_cast(double, b); }
// true iff some lane of the wave (device) / this element (host) raises `p`
GFT_HD inline bool any_lane(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p) != 0;
#else
    return p;
#endif
}

struct Scalar2 {  // a scalar crossing the host->kernel boundary: {v,unused} or {lo,hi}
    double a, b;
};

struct EF64 {
    typedef double V;
    static constexpr int W = 1;
    GFT_HD static V ld(const double* p, size_t, size_t i) { return p[i]; }
    GFT_HD static void st(double* p, size_t, size_t i, V v) { p[i] = v; }
    GFT_HD static V from(Scalar2 s) { return s.a; }
    GFT_HD static V zero() { return 0.0; }
    GFT_HD static V one() { return 1.0; }
    GFT_HD static V from_u32(unsigned u) { return (double)u; }  // f64.rs:19-24
    GFT_HD static bool is_zero(V x) { return x == 0.0; }        // f64.rs:181-183
    GFT_HD static bool eq(V a, V b) { return a == b; }
    GFT_HD static V neg(V a) { return -a; }
    GFT_HD static V add(V a, V b) { return a + b; }
    GFT_HD static V sub(V a, V b) { return a - b; }
    GFT_HD static V mul(V a, V b) { return a * b; }
    GFT_HD static V div(V a, V b) { return a / b; }
    GFT_HD static V mac(V acc, V a, V b) { return acc + a * b; }  // separate multiply and add (-ffp-contract=off)
    GFT_HD static V mulw(V a, V b) { return a * b; }   // "wave-checked" variants: plain ops for f64
    GFT_HD static V addw(V a, V b) { return a + b; }
    GFT_HD static V add0(V b) { return 0.0 + b; }      // (0 + b): turns -0 into +0, so it is not skipped
    // positive-regime interface of the interval functor (no such regime for plain f64)
    static constexpr bool HAS_POS = false;
    GFT_HD static bool pos_ok(V) { return false; }
    GFT_HD static V mul_pos(V a, V b) { return a * b; }
    GFT_HD static V mac_pos(V acc, V a, V b, bool&) { return acc + a * b; }
    GFT_HD static bool pos_first_ok(V) { return true; }
    GFT_HD static bool pos_result_ok(V) { return true; }
    GFT_HD static bool fin_ok(V) { return false; }
    GFT_HD static V mul_fin(V a, V b) { return a * b; }
    GFT_HD static V mac_fin(V acc, V a, V b) { return acc + a * b; }
    GFT_HD static bool fin_result_ok(V) { return true; }
    GFT_HD static V exp(V a) { return ::exp(a); }  // f64.rs:54-56
    GFT_HD static V log(V a) { return ::log(a); }  // f64.rs:59-61
};

// f64.rs:127-171 — integer arithmetic on the bits.
// next_up(x): bits + (+1 | -1 by sign) is right for every input except -0.0, NaN and +inf.  Canonicalising the
// zero first (x + 0.0 == +0.0 for both zeros) removes one fix-up, and !(x < +inf) covers NaN and +inf with a single
// compare; next_down(x) == -next_up(-x) for every input (f64.rs:127-171 is symmetric), so one routine serves both.
GFT_HD inline double next_up(double x) {
    const double t = x + 0.0;  // -0 -> +0 (NaN stays NaN, everything else unchanged)
    const long long bits = f64_bits(t);
    const long long step = (bits >> 63) | 1LL;  // +1 for the positive half, -1 for the negative half
    const long long r = bits + step;
    const double inf = bits_f64(0x7ff0000000000000LL);
    return !(t < inf) ? x : bits_f64(r);  // NaN / +inf: unchanged (the reference returns x itself)
}
GFT_HD inline double next_down(double x) { return -next_up(-x); }
GFT_HD inline double fmin_ref(double a, double b) { return a < b ? a : b; }  // f64.rs:68-74
GFT_HD inline double fmax_ref(double a, double b) { return a > b ? a : b; }  // f64.rs:77-83
GFT_HD inline bool finite_d(double x) { return (x - x) == 0.0; }

struct Iv {
    double lo, hi;
};

struct EIv {
    typedef Iv V;
    static constexpr int W = 2;
    GFT_HD static V ld(const double* p, size_t plane, size_t i) { return Iv{p[i], p[plane + i]}; }
    GFT_HD static void st(double* p, size_t plane, size_t i, V v) {
        p[i] = v.lo;
        p[plane + i] = v.hi;
    }
    GFT_HD static V from(Scalar2 s) { return Iv{s.a, s.b}; }
    GFT_HD static V zero() { return Iv{0.0, 0.0}; }
    GFT_HD static V one() { return Iv{1.0, 1.0}; }
    GFT_HD static V from_u32(unsigned u) { return Iv{(double)u, (double)u}; }  // interval.rs:80-85
    GFT_HD static V widen(double lo, double hi) { return Iv{next_down(lo), next_up(hi)}; }  // :28-31
    GFT_HD static bool is_zero(V x) { return x.lo == 0.0 && x.hi == 0.0; }   // :100-103
    GFT_HD static bool is_one(V x) { return x.lo == 1.0 && x.hi == 1.0; }    // :112-115
    GFT_HD static bool is_finite(V x) { return finite_d(x.lo) && finite_d(x.hi); }
    GFT_HD static bool is_nan(V x) { return x.lo != x.lo || x.hi != x.hi; }
    GFT_HD static bool eq(V a, V b) { return a.lo == b.lo && a.hi == b.hi; }
    GFT_HD static V neg(V a) { return Iv{-a.hi, -a.lo}; }                     // :117-124
    // add / mul are written as "compute the general result, then select" (no control flow): the hot
    // convolution loops run them once per MAC and divergent early returns cost more than the selects.
    GFT_HD static V add(V a, V b) {                                           // :126-139
#if !defined(__HIP_DEVICE_COMPILE__)
        // host pass: the reference's own if-chain (a CPU predicts these branches; the select form below is for lanes)
        if (is_zero(a)) return b;
        if (is_zero(b)) return a;
        return widen(a.lo + b.lo, a.hi + b.hi);
#endif
        V g = widen(a.lo + b.lo, a.hi + b.hi);
        const bool za = is_zero(a), zb = is_zero(b);
        g.lo = zb ? a.lo : g.lo;
        g.hi = zb ? a.hi : g.hi;
        g.lo = za ? b.lo : g.lo;
        g.hi = za ? b.hi : g.hi;
        return g;
    }
    GFT_HD static V sub(V a, V b) { return add(a, neg(b)); }                  // :148-155
    GFT_HD static V mul(V a, V b) {                                           // :164-190
#if !defined(__HIP_DEVICE_COMPILE__)
        if ((is_zero(a) && is_finite(b)) || (is_finite(a) && is_zero(b))) return zero();
        if (is_one(a)) return b;
        if (is_one(b)) return a;
        if (a.lo == -1.0 && a.hi == -1.0) return neg(b);
        if (b.lo == -1.0 && b.hi == -1.0) return neg(a);
        {
            const double p = a.lo * b.lo, q = a.lo * b.hi, r = a.hi * b.lo, s = a.hi * b.hi;
            return widen(fmin_ref(fmin_ref(fmin_ref(p, q), r), s), fmax_ref(fmax_ref(fmax_ref(p, q), r), s));
        }
#endif
        const double p = a.lo * b.lo, q = a.lo * b.hi, r = a.hi * b.lo, s = a.hi * b.hi;
        V g = widen(fmin_ref(fmin_ref(fmin_ref(p, q), r), s), fmax_ref(fmax_ref(fmax_ref(p, q), r), s));
        // the reference's short-circuits, lowest priority first so that the first match of its if-chain wins
        const bool b_m1 = b.lo == -1.0 && b.hi == -1.0, a_m1 = a.lo == -1.0 && a.hi == -1.0;
        const bool b_1 = is_one(b), a_1 = is_one(a);
        const bool z = (is_zero(a) && is_finite(b)) || (is_finite(a) && is_zero(b));
        g.lo = b_m1 ? -a.hi : g.lo;
        g.hi = b_m1 ? -a.lo : g.hi;
        g.lo = a_m1 ? -b.hi : g.lo;
        g.hi = a_m1 ? -b.lo : g.hi;
        g.lo = b_1 ? a.lo : g.lo;
        g.hi = b_1 ? a.hi : g.hi;
        g.lo = a_1 ? b.lo : g.lo;
        g.hi = a_1 ? b.hi : g.hi;
        g.lo = z ? 0.0 : g.lo;
        g.hi = z ? 0.0 : g.hi;
        return g;
    }
    // General product formula for FINITE operands: no product can be NaN (only finite * finite), so the
    // reference's compare-and-select min/max (f64.rs:68-83) equals the hardware min/max up to the sign of a zero,
    // and widen() maps +0 and -0 to the same neighbour.
    GFT_HD static V mul_general_finite(V a, V b) {
        const double p = a.lo * b.lo, q = a.lo * b.hi, r = a.hi * b.lo, s = a.hi * b.hi;
        return widen(__builtin_fmin(__builtin_fmin(p, q), __builtin_fmin(r, s)),
                     __builtin_fmax(__builtin_fmax(p, q), __builtin_fmax(r, s)));
    }
    // acc + a*b as the reference computes it (mul, then add, each with its short-circuits).  When NO lane of the
    // wave holds an operand that can trigger a short-circuit (exact 0 / 1 / -1, inf/NaN, an exactly zero
    // accumulator) the general formulas are the whole story and the ~50 selects and compares of the branch-free
    // versions are skipped: 135 -> ~80 VALU instructions per interval MAC.  One wave-uniform branch (ballot).
    GFT_HD static bool maybe_special(V v) {
        const bool point = v.lo == v.hi;
        const bool unitish = v.lo == 0.0 || __builtin_fabs(v.lo) == 1.0;
        return (point && unitish) || !is_finite(v);
    }
    GFT_HD static V mac(V acc, V a, V b) {
        const bool sp = maybe_special(a) || maybe_special(b) || is_zero(acc);
        if (!any_lane(sp)) {
            const V m = mul_general_finite(a, b);
            return widen(acc.lo + m.lo, acc.hi + m.hi);
        }
        return add(acc, mul(a, b));
    }
    // mul / add with the same wave-uniform shortcut as mac(): the general formula when no lane can short-circuit
    GFT_HD static V mulw(V a, V b) {
        if (!any_lane(maybe_special(a) || maybe_special(b))) return mul_general_finite(a, b);
        return mul(a, b);
    }
    GFT_HD static V addw(V a, V b) {
        if (!any_lane(is_zero(a) || is_zero(b))) return widen(a.lo + b.lo, a.hi + b.hi);
        return add(a, b);
    }
    GFT_HD static V add0(V b) { return b; }  // [0,0] + b returns b unchanged (interval.rs:126-139)
    // ---- positive regime ------------------------------------------------------------------------------------------
    // Probability-like tensors are strictly positive.  For operands with 0 < lo <= hi < inf that are not the point
    // interval [1,1] the reference's product (interval.rs:164-190) takes no short-circuit, its min / max of the four
    // products are lo*lo and hi*hi (round-to-nearest multiplication is monotone), and every bound that follows is
    // positive, where next_down / next_up are the integer steps bits-1 / bits+1.  mac_pos is that arithmetic — the
    // same operations on the same values, so the same bits — for acc != [0,0]; the two cases it cannot represent
    // (a lower bound that rounds down to 0, an upper bound that overflows) are reported through `bad` /
    // pos_result_ok and the caller recomputes the sum with the general mac().
    static constexpr bool HAS_POS = true;
    GFT_HD static bool pos_ok(V v) { return v.lo > 0.0 && v.lo <= v.hi && v.hi < bits_f64(0x7ff0000000000000LL) && !(v.lo == 1.0 && v.hi == 1.0); }
    GFT_HD static double dec_pos(double x) { return bits_f64(f64_bits(x) - 1); }  // next_down for x > 0
    GFT_HD static double inc_pos(double x) { return bits_f64(f64_bits(x) + 1); }  // next_up for 0 <= x < inf
    GFT_HD static V mul_pos(V a, V b) { return Iv{dec_pos(a.lo * b.lo), inc_pos(a.hi * b.hi)}; }
    GFT_HD static V add_pos(V a, V b) { return Iv{dec_pos(a.lo + b.lo), inc_pos(a.hi + b.hi)}; }  // both operands pos_ok: no short-circuit, positive sums
    GFT_HD static V mac_pos(V acc, V a, V b, bool& bad) {
        const double p = a.lo * b.lo;
        const double mlo = dec_pos(p);
        bad = bad || !(mlo > 0.0);  // p underflowed to 0 or to the least subnormal: next_down leaves the positive half
        return Iv{dec_pos(acc.lo + mlo), inc_pos(acc.hi + inc_pos(a.hi * b.hi))};
    }
    // mac_pos without the per-term test: a term that leaves the regime turns the sum's bound into a NaN (dec_pos(+0) and
    // inc_pos(inf) are NaN bit patterns, and NaN survives every later add and integer step), so the caller tests the
    // FINISHED sum with pos_first_ok && pos_result_ok instead
    GFT_HD static V mac_pos_unchecked(V acc, V a, V b) {
        return Iv{dec_pos(acc.lo + dec_pos(a.lo * b.lo)), inc_pos(acc.hi + inc_pos(a.hi * b.hi))};
    }
    // ---- finite regime (round 3): mixed-sign data ----------------------------------------------------------------------
    // Operands that are finite and no exact 0 / +-1 point take no short-circuit (interval.rs:164-190), and a sum that starts
    // from a product is never [0,0] again (widened intervals are not points), so a row sum over such operands is the
    // general formulas all the way: four products, min / max, outward step, add, outward step.  What the general
    // next_up (f64.rs:127-171) spends on top of the integer step `bits + (sign ? -1 : +1)` is the guard "NaN and +inf
    // stay" — and in exactly those two cases the unguarded step yields a NaN pattern, which survives every later add and
    // step.  So the sums run with the unguarded step (4 instructions instead of 9, no per-term operand tests: 80 -> ~30
    // instructions per interval MAC) and ONE test of the finished sum (no NaN) validates every term; a sum that fails is
    // recomputed with the general mac().  Same operations on the same values wherever the test passes => same bits.
    GFT_HD static bool fin_ok(V v) { return !maybe_special(v); }
    GFT_HD static double up_fin(double x) {
        const double t = x + 0.0;  // -0 -> +0
        const long long b = f64_bits(t);
        return bits_f64(b + ((b >> 63) | 1LL));
    }
    GFT_HD static V widen_fin(double lo, double hi) { return Iv{-up_fin(-lo), up_fin(hi)}; }
    GFT_HD static V mul_fin(V a, V b) {
        const double p = a.lo * b.lo, q = a.lo * b.hi, r = a.hi * b.lo, s = a.hi * b.hi;
        return widen_fin(__builtin_fmin(__builtin_fmin(p, q), __builtin_fmin(r, s)), __builtin_fmax(__builtin_fmax(p, q), __builtin_fmax(r, s)));
    }
    GFT_HD static V mac_fin(V acc, V a, V b) {
        const V m = mul_fin(a, b);
        return widen_fin(acc.lo + m.lo, acc.hi + m.hi);
    }
    GFT_HD static bool fin_result_ok(V v) { return !is_nan(v); }
    GFT_HD static bool pos_first_ok(V m) { return m.lo > 0.0; }  // the first product's lower bound stayed positive
    GFT_HD static bool pos_result_ok(V v) { return v.hi < bits_f64(0x7ff0000000000000LL); }  // no overflow on the way (NaN fails too)
    GFT_HD static V div(V a, V b) {                                           // :199-234
        if (is_nan(a) || is_nan(b)) {
            double n = bits_f64(0x7ff8000000000000LL);
            return Iv{n, n};
        }
        if (is_zero(a) && !is_zero(b)) return a;
        if (is_one(b)) return a;
        const double inf = bits_f64(0x7ff0000000000000LL);
        double lo = inf, hi = -inf;
        if (b.lo <= 0.0 && 0.0 <= b.hi) {
            if (0.0 <= a.lo) hi = inf; else lo = -inf;
            if (a.hi <= 0.0) lo = -inf; else hi = inf;
        }
        double p = a.lo / b.lo, q = a.lo / b.hi, r = a.hi / b.lo, s = a.hi / b.hi;
        lo = fmin_ref(fmin_ref(fmin_ref(fmin_ref(lo, p), q), r), s);
        hi = fmax_ref(fmax_ref(fmax_ref(fmax_ref(hi, p), q), r), s);
        return widen(lo, hi);
    }
    GFT_HD static V exp(V a) {                                                // :264-269
        if (is_zero(a)) return one();
        return widen(::exp(a.lo), ::exp(a.hi));
    }
    GFT_HD static V log(V a) {                                                // :271-276
        if (is_one(a)) return zero();
        return widen(::log(a.lo), ::log(a.hi));
    }
};

}  // namespace gft
