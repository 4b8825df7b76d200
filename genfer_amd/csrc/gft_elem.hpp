// Element types of the Taylor tensors on the device: F64 (src/number/f64.rs) and
// Interval<F64> (src/interval.rs) with identical static interfaces, so every structural kernel
// is written once.  Interval tensors are stored as two planes (lo, hi) `plane` doubles apart.
//
// The whole library is compiled with -ffp-contract=off: the reference never fuses a*b+c and
// Interval widening assumes separately rounded operations.  FMA is used only where a kernel
// asks for it explicitly (the tiled f64 convolution).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace gft {

struct Scalar2 {  // a scalar crossing the host->kernel boundary: {v,unused} or {lo,hi}
    double a, b;
};

struct EF64 {
    typedef double V;
    static constexpr int W = 1;
    __device__ static V ld(const double* p, size_t, size_t i) { return p[i]; }
    __device__ static void st(double* p, size_t, size_t i, V v) { p[i] = v; }
    __device__ static V from(Scalar2 s) { return s.a; }
    __device__ static V zero() { return 0.0; }
    __device__ static V one() { return 1.0; }
    __device__ static V from_u32(unsigned u) { return (double)u; }  // f64.rs:19-24
    __device__ static bool is_zero(V x) { return x == 0.0; }        // f64.rs:181-183
    __device__ static bool eq(V a, V b) { return a == b; }
    __device__ static V neg(V a) { return -a; }
    __device__ static V add(V a, V b) { return a + b; }
    __device__ static V sub(V a, V b) { return a - b; }
    __device__ static V mul(V a, V b) { return a * b; }
    __device__ static V div(V a, V b) { return a / b; }
    __device__ static V mac(V acc, V a, V b) { return acc + a * b; }  // separate multiply and add (-ffp-contract=off)
    __device__ static V mulw(V a, V b) { return a * b; }   // "wave-checked" variants: plain ops for f64
    __device__ static V addw(V a, V b) { return a + b; }
    __device__ static V add0(V b) { return 0.0 + b; }      // (0 + b): turns -0 into +0, so it is not skipped
    __device__ static V exp(V a) { return ::exp(a); }  // f64.rs:54-56
    __device__ static V log(V a) { return ::log(a); }  // f64.rs:59-61
};

// f64.rs:127-171 — integer arithmetic on the bits.
// next_up(x): bits + (+1 | -1 by sign) is right for every input except -0.0, NaN and +inf.  Canonicalising the
// zero first (x + 0.0 == +0.0 for both zeros) removes one fix-up, and !(x < +inf) covers NaN and +inf with a single
// compare; next_down(x) == -next_up(-x) for every input (f64.rs:127-171 is symmetric), so one routine serves both.
__device__ inline double next_up(double x) {
    const double t = x + 0.0;  // -0 -> +0 (NaN stays NaN, everything else unchanged)
    const long long bits = __double_as_longlong(t);
    const long long step = (bits >> 63) | 1LL;  // +1 for the positive half, -1 for the negative half
    const long long r = bits + step;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    return !(t < inf) ? x : __longlong_as_double(r);  // NaN / +inf: unchanged (the reference returns x itself)
}
__device__ inline double next_down(double x) { return -next_up(-x); }
__device__ inline double fmin_ref(double a, double b) { return a < b ? a : b; }  // f64.rs:68-74
__device__ inline double fmax_ref(double a, double b) { return a > b ? a : b; }  // f64.rs:77-83
__device__ inline bool finite_d(double x) { return (x - x) == 0.0; }

struct Iv {
    double lo, hi;
};

struct EIv {
    typedef Iv V;
    static constexpr int W = 2;
    __device__ static V ld(const double* p, size_t plane, size_t i) { return Iv{p[i], p[plane + i]}; }
    __device__ static void st(double* p, size_t plane, size_t i, V v) {
        p[i] = v.lo;
        p[plane + i] = v.hi;
    }
    __device__ static V from(Scalar2 s) { return Iv{s.a, s.b}; }
    __device__ static V zero() { return Iv{0.0, 0.0}; }
    __device__ static V one() { return Iv{1.0, 1.0}; }
    __device__ static V from_u32(unsigned u) { return Iv{(double)u, (double)u}; }  // interval.rs:80-85
    __device__ static V widen(double lo, double hi) { return Iv{next_down(lo), next_up(hi)}; }  // :28-31
    __device__ static bool is_zero(V x) { return x.lo == 0.0 && x.hi == 0.0; }   // :100-103
    __device__ static bool is_one(V x) { return x.lo == 1.0 && x.hi == 1.0; }    // :112-115
    __device__ static bool is_finite(V x) { return finite_d(x.lo) && finite_d(x.hi); }
    __device__ static bool is_nan(V x) { return x.lo != x.lo || x.hi != x.hi; }
    __device__ static bool eq(V a, V b) { return a.lo == b.lo && a.hi == b.hi; }
    __device__ static V neg(V a) { return Iv{-a.hi, -a.lo}; }                     // :117-124
    // add / mul are written as "compute the general result, then select" (no control flow): the hot
    // convolution loops run them once per MAC and divergent early returns cost more than the selects.
    __device__ static V add(V a, V b) {                                           // :126-139
        V g = widen(a.lo + b.lo, a.hi + b.hi);
        const bool za = is_zero(a), zb = is_zero(b);
        g.lo = zb ? a.lo : g.lo;
        g.hi = zb ? a.hi : g.hi;
        g.lo = za ? b.lo : g.lo;
        g.hi = za ? b.hi : g.hi;
        return g;
    }
    __device__ static V sub(V a, V b) { return add(a, neg(b)); }                  // :148-155
    __device__ static V mul(V a, V b) {                                           // :164-190
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
    __device__ static V mul_general_finite(V a, V b) {
        const double p = a.lo * b.lo, q = a.lo * b.hi, r = a.hi * b.lo, s = a.hi * b.hi;
        return widen(__builtin_fmin(__builtin_fmin(p, q), __builtin_fmin(r, s)),
                     __builtin_fmax(__builtin_fmax(p, q), __builtin_fmax(r, s)));
    }
    // acc + a*b as the reference computes it (mul, then add, each with its short-circuits).  When NO lane of the
    // wave holds an operand that can trigger a short-circuit (exact 0 / 1 / -1, inf/NaN, an exactly zero
    // accumulator) the general formulas are the whole story and the ~50 selects and compares of the branch-free
    // versions are skipped: 135 -> ~80 VALU instructions per interval MAC.  One wave-uniform branch (ballot).
    __device__ static bool maybe_special(V v) {
        const bool point = v.lo == v.hi;
        const bool unitish = v.lo == 0.0 || __builtin_fabs(v.lo) == 1.0;
        return (point && unitish) || !is_finite(v);
    }
    __device__ static V mac(V acc, V a, V b) {
        const bool sp = maybe_special(a) || maybe_special(b) || is_zero(acc);
        if (__builtin_amdgcn_ballot_w64(sp) == 0) {
            const V m = mul_general_finite(a, b);
            return widen(acc.lo + m.lo, acc.hi + m.hi);
        }
        return add(acc, mul(a, b));
    }
    // mul / add with the same wave-uniform shortcut as mac(): the general formula when no lane can short-circuit
    __device__ static V mulw(V a, V b) {
        if (__builtin_amdgcn_ballot_w64(maybe_special(a) || maybe_special(b)) == 0) return mul_general_finite(a, b);
        return mul(a, b);
    }
    __device__ static V addw(V a, V b) {
        if (__builtin_amdgcn_ballot_w64(is_zero(a) || is_zero(b)) == 0) return widen(a.lo + b.lo, a.hi + b.hi);
        return add(a, b);
    }
    __device__ static V add0(V b) { return b; }  // [0,0] + b returns b unchanged (interval.rs:126-139)
    __device__ static V div(V a, V b) {                                           // :199-234
        if (is_nan(a) || is_nan(b)) {
            double n = __longlong_as_double(0x7ff8000000000000LL);
            return Iv{n, n};
        }
        if (is_zero(a) && !is_zero(b)) return a;
        if (is_one(b)) return a;
        const double inf = __longlong_as_double(0x7ff0000000000000LL);
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
    __device__ static V exp(V a) {                                                // :264-269
        if (is_zero(a)) return one();
        return widen(::exp(a.lo), ::exp(a.hi));
    }
    __device__ static V log(V a) {                                                // :271-276
        if (is_one(a)) return zero();
        return widen(::log(a.lo), ::log(a.hi));
    }
};

}  // namespace gft
