// Host interpreter, part 2: the scalar `Number` types the evaluator itself computes with
// (program constants, evaluation points, moment post-processing) — F64 (src/number/f64.rs) and
// Interval<F64> (src/interval.rs) — and the reference's float formatting (ryu, f64.rs:41-45).
// Tensor arithmetic never happens here; it goes through the C ABI (gfh_backend.hpp).
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>

#include "../gft_fmt.hpp"

namespace gfh {

using gftfmt::fmt_f64;

inline double next_up(double x) {  // f64.rs:127-147
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    if (std::isnan(x) || bits == 0x7ff0000000000000ULL) return x;
    uint64_t abs = bits & 0x7fffffffffffffffULL;
    uint64_t next = (abs == 0) ? 0x1ULL : (bits == abs ? bits + 1 : bits - 1);
    double r;
    std::memcpy(&r, &next, 8);
    return r;
}
inline double next_down(double x) {  // f64.rs:150-171
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    if (std::isnan(x) || bits == 0xfff0000000000000ULL) return x;
    uint64_t abs = bits & 0x7fffffffffffffffULL;
    uint64_t next = (abs == 0) ? 0x8000000000000001ULL : (bits == abs ? bits - 1 : bits + 1);
    double r;
    std::memcpy(&r, &next, 8);
    return r;
}

struct F64 {
    static constexpr int WIDTH = 1;
    double v = 0.0;
    F64() {}
    F64(double x) : v(x) {}
    static F64 zero() { return F64(0.0); }
    static F64 one() { return F64(1.0); }
    static F64 from_u32(uint32_t u) { return F64((double)u); }
    static F64 from_ratio(uint64_t n, uint64_t d) { return F64((double)n / (double)d); }  // f64.rs:49-51
    static F64 infinity() { return F64(std::numeric_limits<double>::infinity()); }
    static F64 nan() { return F64(std::numeric_limits<double>::quiet_NaN()); }
    bool is_zero() const { return v == 0.0; }
    bool is_one() const { return v == 1.0; }
    bool is_nan() const { return std::isnan(v); }
    bool is_finite() const { return std::isfinite(v); }
    bool is_infinite() const { return std::isinf(v); }
    F64 exp() const { return F64(std::exp(v)); }
    F64 log() const { return F64(std::log(v)); }
    F64 sqrt() const { return F64(std::sqrt(v)); }
    F64 abs() const { return F64(std::fabs(v)); }
    F64 pow(uint32_t e) const { return F64(__builtin_powi(v, (int)e)); }  // f64::powi (f64.rs:64-66)
    F64 min(const F64& o) const { return v < o.v ? *this : o; }           // f64.rs:68-74
    F64 max(const F64& o) const { return v > o.v ? *this : o; }           // f64.rs:77-83
    F64 next_up() const { return F64(gfh::next_up(v)); }
    F64 next_down() const { return F64(gfh::next_down(v)); }
    double to_f64() const { return v; }
    std::string str() const { return fmt_f64(v); }
    bool operator==(const F64& o) const { return v == o.v; }
    bool operator!=(const F64& o) const { return !(v == o.v); }
    bool operator<(const F64& o) const { return v < o.v; }
    bool operator<=(const F64& o) const { return v <= o.v; }
    bool operator>(const F64& o) const { return v > o.v; }
    bool operator>=(const F64& o) const { return v >= o.v; }
    // partial_cmp(..) != Some(Less)
    bool not_less_than(const F64& o) const { return !(v < o.v); }
    void store(double* b) const { b[0] = v; }
    static F64 load(const double* b) { return F64(b[0]); }
    void store_plane(double* d, size_t, size_t i) const { d[i] = v; }
    static F64 load_plane(const double* d, size_t, size_t i) { return F64(d[i]); }
};
inline F64 operator-(F64 a) { return F64(-a.v); }
inline F64 operator+(F64 a, F64 b) { return F64(a.v + b.v); }
inline F64 operator-(F64 a, F64 b) { return F64(a.v - b.v); }
inline F64 operator*(F64 a, F64 b) { return F64(a.v * b.v); }
inline F64 operator/(F64 a, F64 b) { return F64(a.v / b.v); }

// interval.rs:12-15, instantiated at F64 (the only instantiation this build supports).
struct Interval {
    static constexpr int WIDTH = 2;
    F64 lo, hi;
    Interval() {}
    Interval(F64 l, F64 h) : lo(l), hi(h) {}
    static Interval exact(F64 l, F64 h) { return Interval(l, h); }
    static Interval precisely(F64 x) { return Interval(x, x); }
    static Interval widen(F64 l, F64 h) { return Interval(l.next_down(), h.next_up()); }  // :28-31
    static Interval zero() { return Interval(0.0, 0.0); }
    static Interval one() { return Interval(1.0, 1.0); }
    static Interval from_u32(uint32_t u) { return Interval((double)u, (double)u); }
    static Interval from_ratio(uint64_t n, uint64_t d) {  // number.rs:24-32 (trait default)
        Interval two32 = from_u32(UINT32_MAX) + one();
        Interval numer = from_u32((uint32_t)n) + from_u32((uint32_t)(n >> 32)) * two32;
        Interval denom = from_u32((uint32_t)d) + from_u32((uint32_t)(d >> 32)) * two32;
        return numer / denom;
    }
    static Interval infinity() { return Interval(F64::infinity(), F64::infinity()); }
    static Interval nan() { return Interval(F64::nan(), F64::nan()); }
    bool is_zero() const { return lo.is_zero() && hi.is_zero(); }
    bool is_one() const { return lo.is_one() && hi.is_one(); }
    bool is_finite() const { return lo.is_finite() && hi.is_finite(); }
    bool is_nan() const { return lo.is_nan() || hi.is_nan(); }
    bool is_infinite() const { return lo.is_infinite() || hi.is_infinite(); }
    bool contains(F64 x) const { return lo <= x && x <= hi; }
    Interval unite(F64 x) const { return Interval(lo.min(x), hi.max(x)); }  // union, :38-41
    bool extract_point(F64& out) const { if (lo == hi) { out = lo; return true; } return false; }
    F64 center() const { return (lo + hi) / F64(2.0); }
    Interval ensure_lower_bound(F64 nl) const { return lo < nl ? Interval(nl, hi) : *this; }  // :62-69
    Interval ensure_upper_bound(F64 nh) const { return hi > nh ? Interval(lo, nh) : *this; }  // :71-78
    Interval exp() const { return is_zero() ? one() : widen(lo.exp(), hi.exp()); }
    Interval log() const { return is_one() ? zero() : widen(lo.log(), hi.log()); }
    Interval pow(uint32_t e) const {  // :278-285
        Interval r = widen(lo.pow(e), hi.pow(e));
        return contains(F64::zero()) ? r.unite(F64::zero()) : r;
    }
    Interval min(const Interval& o) const { return Interval(lo.min(o.lo), hi.min(o.hi)); }
    Interval max(const Interval& o) const { return Interval(lo.max(o.lo), hi.max(o.hi)); }
    Interval abs() const {
        Interval r = widen(lo.abs(), hi.abs());
        return contains(F64::zero()) ? r.unite(F64::zero()) : r;
    }
    Interval sqrt() const {  // :304-311
        F64 l = lo < F64::zero() ? F64::zero() : lo.sqrt();
        return widen(l, hi.sqrt());
    }
    std::string str() const { return "[" + lo.str() + ", " + hi.str() + "]"; }
    bool operator==(const Interval& o) const { return lo == o.lo && hi == o.hi; }
    bool operator!=(const Interval& o) const { return !(*this == o); }
    // PartialOrd (:242-254)
    bool operator<(const Interval& o) const { return !(lo == o.lo && hi == o.hi) && hi <= o.lo; }
    bool operator>(const Interval& o) const { return !(lo == o.lo && hi == o.hi) && !(hi <= o.lo) && lo >= o.hi; }
    bool not_less_than(const Interval& o) const { return !(*this < o); }
    void store(double* b) const { b[0] = lo.v; b[1] = hi.v; }
    static Interval load(const double* b) { return Interval(b[0], b[1]); }
    void store_plane(double* d, size_t n, size_t i) const { d[i] = lo.v; d[n + i] = hi.v; }
    static Interval load_plane(const double* d, size_t n, size_t i) { return Interval(d[i], d[n + i]); }

    friend Interval operator-(Interval a) { return Interval(-a.hi, -a.lo); }
    friend Interval operator+(Interval a, Interval b) {  // :126-139
        if (a.is_zero()) return b;
        if (b.is_zero()) return a;
        return widen(a.lo + b.lo, a.hi + b.hi);
    }
    friend Interval operator-(Interval a, Interval b) { return a + (-b); }
    friend Interval operator*(Interval a, Interval b) {  // :164-190
        if ((a.is_zero() && b.is_finite()) || (a.is_finite() && b.is_zero())) return zero();
        if (a.is_one()) return b;
        if (b.is_one()) return a;
        if ((-a).is_one()) return -b;
        if ((-b).is_one()) return -a;
        F64 p = a.lo * b.lo, q = a.lo * b.hi, r = a.hi * b.lo, s = a.hi * b.hi;
        return widen(p.min(q).min(r).min(s), p.max(q).max(r).max(s));
    }
    friend Interval operator/(Interval a, Interval b) {  // :199-234
        if (a.is_nan() || b.is_nan()) return nan();
        if (a.is_zero() && !b.is_zero()) return a;
        if (b.is_one()) return a;
        F64 lo = F64::infinity(), hi = -F64::infinity();
        if (b.contains(F64::zero())) {
            if (F64::zero() <= a.lo) hi = F64::infinity(); else lo = -F64::infinity();
            if (a.hi <= F64::zero()) lo = -F64::infinity(); else hi = F64::infinity();
        }
        F64 p = a.lo / b.lo, q = a.lo / b.hi, r = a.hi / b.lo, s = a.hi / b.hi;
        lo = lo.min(p).min(q).min(r).min(s);
        hi = hi.max(p).max(q).max(r).max(s);
        return widen(lo, hi);
    }
};

}  // namespace gfh
