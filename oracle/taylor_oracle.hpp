// TEST INFRASTRUCTURE ONLY — CPU oracle for the multivariate-Taylor hot path.
//
// This header is a scalar, single-threaded C++ restatement of the reference's
// `TaylorPoly<T>` (src/multivariate_taylor.rs), `F64` (src/number/f64.rs) and
// `Interval<T>` (src/interval.rs).  It exists to CHECK the HIP product path; nothing in
// `genfer_amd/` may include, link or call it (only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg do).
//
// Parity pinning: the reference is Rust and cannot be built in this image (no cargo/rustc), so the oracle is pinned by
//   * the reference's own literal unit-test vectors (src/multivariate_taylor.rs:733-1513 and
//     src/univariate_taylor.rs, transcribed to tests/golden/unit_vectors.json), reproduced bit-exactly by
//     tests/test_reference_unit_vectors.py;
//   * 109 of the reference's `.sgcl` -> `.expect` report snapshots, byte for byte, through the host interpreter
//     (tests/test_e2e_snapshots.py);
//   * exact rational known-answers generated with fractions.Fraction (tests/golden/make_exact_kats.py ->
//     tests/test_exact_kats.py), independent of the reference's code;
//   * for Interval<F64> (no reference vector exists for `--bounds`): exact-rational ENCLOSURE known-answers and a
//     bit-for-bit cross-check against the interpreter's separately written Interval (tests/test_interval_pins.py).
//
// Faithfulness rules followed here:
//   * same loop nests and the same floating-point summation order as the reference;
//   * separate multiply and add (compile with -ffp-contract=off; Rust never fuses);
//   * integer shape/degree bookkeeping identical (usize, saturating ops, usize::MAX =
//     "untruncated").
//   * ndarray 0.15.6 (third-party, Cargo.toml:17, not under /root/reference) is restated
//     where its algorithm is observable: `sum_axis` (slab-by-slab ascending, with the
//     2-D/unit-stride branch that sums each lane with the 8-way unrolled fold).
//
// Every function cites the reference lines it follows as `mt:<lines>` (=
// src/multivariate_taylor.rs), `f64:<lines>` (src/number/f64.rs), `iv:<lines>`
// (src/interval.rs).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

namespace orc {

using usize = std::size_t;
constexpr usize UMAX = std::numeric_limits<usize>::max();

[[noreturn]] inline void panic(const std::string& msg) { throw std::runtime_error(msg); }

inline usize sat_sub(usize a, usize b) { return a > b ? a - b : 0; }

// ---------------------------------------------------------------------------------------
// Scalar types
// ---------------------------------------------------------------------------------------

// f64:11 — IEEE binary64, round-to-nearest, plain + - * / (f64:202-262).
struct F64 {
    double v;
    F64() : v(0.0) {}
    F64(double x) : v(x) {}
    static F64 zero() { return F64(0.0); }                  // f64:175-178
    static F64 one() { return F64(1.0); }                   // f64:187-190
    static F64 from_u32(uint32_t u) { return F64((double)u); }  // f64:19-24
    bool is_zero() const { return v == 0.0; }               // f64:181-183
    bool is_one() const { return v == 1.0; }                // num_traits::One default
    F64 exp() const { return F64(std::exp(v)); }            // f64:54-56
    F64 log() const { return F64(std::log(v)); }            // f64:59-61
    bool operator==(const F64& o) const { return v == o.v; }  // derived PartialEq
};
inline F64 operator-(F64 a) { return F64(-a.v); }
inline F64 operator+(F64 a, F64 b) { return F64(a.v + b.v); }
inline F64 operator-(F64 a, F64 b) { return F64(a.v - b.v); }
inline F64 operator*(F64 a, F64 b) { return F64(a.v * b.v); }
inline F64 operator/(F64 a, F64 b) { return F64(a.v / b.v); }

// f64:127-147 — next_up by integer arithmetic on the bits.
inline double next_up(double x) {
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    const uint64_t inf_bits = 0x7ff0000000000000ULL;
    if (std::isnan(x) || bits == inf_bits) return x;
    uint64_t abs = bits & 0x7fffffffffffffffULL;
    uint64_t next = (abs == 0) ? 0x1ULL : (bits == abs ? bits + 1 : bits - 1);
    double r;
    std::memcpy(&r, &next, 8);
    return r;
}
// f64:150-171
inline double next_down(double x) {
    uint64_t bits;
    std::memcpy(&bits, &x, 8);
    const uint64_t ninf_bits = 0xfff0000000000000ULL;
    if (std::isnan(x) || bits == ninf_bits) return x;
    uint64_t abs = bits & 0x7fffffffffffffffULL;
    uint64_t next = (abs == 0) ? 0x8000000000000001ULL : (bits == abs ? bits - 1 : bits + 1);
    double r;
    std::memcpy(&r, &next, 8);
    return r;
}
// f64:68-83 — note: `if self < other {self} else {other}` (NaN falls to `other`).
inline double f_min(double a, double b) { return a < b ? a : b; }
inline double f_max(double a, double b) { return a > b ? a : b; }

// iv:12-15 — nearest rounding + one-ULP outward widening (iv:28-31); NOT directed rounding.
struct Interval {
    double lo, hi;
    Interval() : lo(0.0), hi(0.0) {}
    Interval(double l, double h) : lo(l), hi(h) {}
    static Interval exact(double l, double h) { return Interval(l, h); }       // iv:19-21
    static Interval precisely(double x) { return Interval(x, x); }              // iv:24-26
    static Interval widen(double l, double h) { return Interval(next_down(l), next_up(h)); }  // iv:28-31
    static Interval zero() { return Interval(0.0, 0.0); }                       // iv:95-98
    static Interval one() { return Interval(1.0, 1.0); }                        // iv:107-110
    static Interval from_u32(uint32_t u) { return Interval((double)u, (double)u); }  // iv:80-85
    static Interval nan() { double n = std::numeric_limits<double>::quiet_NaN(); return Interval(n, n); }
    bool is_zero() const { return lo == 0.0 && hi == 0.0; }                     // iv:100-103
    bool is_one() const { return lo == 1.0 && hi == 1.0; }                      // iv:112-115
    bool is_finite() const { return std::isfinite(lo) && std::isfinite(hi); }   // iv:316-318
    bool is_nan() const { return std::isnan(lo) || std::isnan(hi); }            // iv:320-322
    bool contains(double x) const { return lo <= x && x <= hi; }                // iv:33-36
    bool operator==(const Interval& o) const { return lo == o.lo && hi == o.hi; }  // derived PartialEq
    Interval exp() const {                                                       // iv:264-269
        if (is_zero()) return one();
        return widen(std::exp(lo), std::exp(hi));
    }
    Interval log() const {                                                       // iv:271-276
        if (is_one()) return zero();
        return widen(std::log(lo), std::log(hi));
    }
};
inline Interval operator-(Interval a) { return Interval(-a.hi, -a.lo); }         // iv:117-124
inline Interval operator+(Interval a, Interval b) {                              // iv:126-139
    if (a.is_zero()) return b;
    if (b.is_zero()) return a;
    return Interval::widen(a.lo + b.lo, a.hi + b.hi);
}
inline Interval operator-(Interval a, Interval b) { return a + (-b); }           // iv:148-155
inline Interval operator*(Interval a, Interval b) {                              // iv:164-190
    if ((a.is_zero() && b.is_finite()) || (a.is_finite() && b.is_zero())) return Interval::zero();
    if (a.is_one()) return b;
    if (b.is_one()) return a;
    if ((-a).is_one()) return -b;
    if ((-b).is_one()) return -a;
    double p = a.lo * b.lo, q = a.lo * b.hi, r = a.hi * b.lo, s = a.hi * b.hi;
    return Interval::widen(f_min(f_min(f_min(p, q), r), s), f_max(f_max(f_max(p, q), r), s));
}
inline Interval operator/(Interval a, Interval b) {                              // iv:199-234
    if (a.is_nan() || b.is_nan()) return Interval::nan();
    if (a.is_zero() && !b.is_zero()) return a;
    if (b.is_one()) return a;
    const double inf = std::numeric_limits<double>::infinity();
    double lo = inf, hi = -inf;
    if (b.contains(0.0)) {
        if (0.0 <= a.lo) hi = inf; else lo = -inf;
        if (a.hi <= 0.0) lo = -inf; else hi = inf;
    }
    double p = a.lo / b.lo, q = a.lo / b.hi, r = a.hi / b.lo, s = a.hi / b.hi;
    lo = f_min(f_min(f_min(f_min(lo, p), q), r), s);
    hi = f_max(f_max(f_max(f_max(hi, p), q), r), s);
    return Interval::widen(lo, hi);
}

// ---------------------------------------------------------------------------------------
// Strided N-d views (what ndarray's ArrayViewD gives the reference)
// ---------------------------------------------------------------------------------------

inline usize numel(const std::vector<usize>& shape) {
    usize n = 1;
    for (usize s : shape) n *= s;
    return n;
}

inline std::vector<usize> c_strides(const std::vector<usize>& shape) {
    std::vector<usize> st(shape.size(), 1);
    for (usize i = shape.size(); i-- > 1;) st[i - 1] = st[i] * shape[i];
    return st;
}

template <class S>
struct View {
    S* p;
    std::vector<usize> shape;
    std::vector<usize> stride;
    usize ndim() const { return shape.size(); }
    usize len() const { return numel(shape); }
    bool is_empty() const { return len() == 0; }
    usize len_of(usize ax) const { return shape[ax]; }
    // index_axis(Axis(0), i): drop the leading axis.
    View index0(usize i) const {
        View r;
        r.p = p + i * stride[0];
        r.shape.assign(shape.begin() + 1, shape.end());
        r.stride.assign(stride.begin() + 1, stride.end());
        return r;
    }
    // slice_each_axis(0..lens[ax]) — leading sub-block
    View lead_block(const std::vector<usize>& lens) const {
        View r = *this;
        for (usize a = 0; a < shape.size(); ++a) r.shape[a] = lens[a];
        return r;
    }
    S& first() const { return *p; }
};

// Visit every element of a view in logical (row-major) order.
template <class S, class F>
inline void for_each(const View<S>& v, F&& f) {
    usize nd = v.ndim();
    usize total = v.len();
    if (total == 0) return;
    std::vector<usize> idx(nd, 0);
    for (usize c = 0; c < total; ++c) {
        usize off = 0;
        for (usize a = 0; a < nd; ++a) off += idx[a] * v.stride[a];
        f(v.p[off]);
        for (usize a = nd; a-- > 0;) {
            if (++idx[a] < v.shape[a]) break;
            idx[a] = 0;
        }
    }
}

// Visit pairs of elements of two equally shaped views (ndarray Zip / add_assign order).
template <class S, class S2, class F>
inline void zip_each(const View<S>& a, const View<S2>& b, F&& f) {
    usize nd = a.ndim();
    if (b.ndim() != nd) panic("zip_each: ndim mismatch");
    for (usize i = 0; i < nd; ++i)
        if (a.shape[i] != b.shape[i]) panic("zip_each: shape mismatch");
    usize total = a.len();
    if (total == 0) return;
    std::vector<usize> idx(nd, 0);
    for (usize c = 0; c < total; ++c) {
        usize oa = 0, ob = 0;
        for (usize x = 0; x < nd; ++x) {
            oa += idx[x] * a.stride[x];
            ob += idx[x] * b.stride[x];
        }
        f(a.p[oa], b.p[ob]);
        for (usize x = nd; x-- > 0;) {
            if (++idx[x] < a.shape[x]) break;
            idx[x] = 0;
        }
    }
}

template <class S>
inline std::vector<typename std::remove_const<S>::type> collect(const View<S>& v) {
    std::vector<typename std::remove_const<S>::type> out;
    out.reserve(v.len());
    for_each(v, [&](const S& x) { out.push_back(x); });
    return out;
}

// Owned row-major array.
template <class S>
struct Arr {
    std::vector<usize> shape;
    std::vector<S> data;
    Arr() {}
    explicit Arr(const std::vector<usize>& sh) : shape(sh), data(numel(sh), S::zero()) {}  // zeros
    usize ndim() const { return shape.size(); }
    usize len() const { return data.size(); }
    View<S> view() { return View<S>{data.data(), shape, c_strides(shape)}; }
    View<const S> cview() const { return View<const S>{data.data(), shape, c_strides(shape)}; }
    static Arr from_view(const View<const S>& v) {  // to_owned()
        Arr a;
        a.shape = v.shape;
        a.data = collect(v);
        return a;
    }
};

template <class S>
inline View<const S> as_const(const View<S>& v) {
    return View<const S>{v.p, v.shape, v.stride};
}

// ---------------------------------------------------------------------------------------
// TaylorPoly
// ---------------------------------------------------------------------------------------

// mt:958-969
inline bool extract_1d_len(const std::vector<usize>& shape, usize& out) {
    bool found = false;
    for (usize len : shape) {
        if (len != 1) {
            if (found) return false;
            found = true;
            out = len;
        }
    }
    return found;
}

// mt:971-982 — zs[k] starts at 0 and accumulates x[j]*y[k-j] for ascending j.
template <class S>
inline std::vector<S> mul_1d(const std::vector<S>& xs, const std::vector<S>& ys, usize n) {
    std::vector<S> zs(n, S::zero());
    for (usize k = 0; k < n; ++k) {
        usize lo = sat_sub(k + 1, ys.size());
        usize hi = std::min(k + 1, xs.size());
        for (usize j = lo; j < hi; ++j) zs[k] = zs[k] + xs[j] * ys[k - j];
    }
    return zs;
}

// mt:984-1012 — accumulating truncated N-d Cauchy product.
template <class S>
void mul_rec(const View<const S>& xs, const View<const S>& ys, const View<S>& res) {
    if (res.is_empty()) return;
    if (res.ndim() == 0) {
        res.first() = res.first() + xs.first() * ys.first();
        return;
    }
    usize n;
    if (extract_1d_len(res.shape, n)) {
        std::vector<S> out = mul_1d(collect(xs), collect(ys), n);
        usize i = 0;
        for_each(res, [&](S& z) { z = z + out[i++]; });
        return;
    }
    usize n0 = res.len_of(0);
    for (usize k = 0; k < n0; ++k) {
        View<S> z = res.index0(k);
        usize lo = sat_sub(k + 1, ys.len_of(0));
        usize hi = std::min(k + 1, xs.len_of(0));
        for (usize j = lo; j < hi; ++j) mul_rec(xs.index0(j), ys.index0(k - j), z);
    }
}

template <class S>
struct TaylorPoly {
    Arr<S> coeffs;
    std::vector<usize> degrees_p1;

    // mt:23-31 (debug_assert in the reference; always checked here)
    void check_invariants() const {
        if (coeffs.ndim() != degrees_p1.size()) panic("invariant: ndim != degrees_p1.len()");
        for (usize v = 0; v < degrees_p1.size(); ++v)
            if (!(0 < coeffs.shape[v] && coeffs.shape[v] <= degrees_p1[v]))
                panic("invariant: 0 < shape[v] <= degrees_p1[v] violated");
    }

    TaylorPoly() {}
    TaylorPoly(Arr<S> c, std::vector<usize> d) : coeffs(std::move(c)), degrees_p1(std::move(d)) {  // mt:33-41
        check_invariants();
    }
    static TaylorPoly from_coeffs(Arr<S> c) {  // mt:43-46
        std::vector<usize> sh = c.shape;
        return TaylorPoly(std::move(c), sh);
    }
    static TaylorPoly from_scalar(S x) {  // mt:626-630
        Arr<S> a;
        a.data.push_back(x);
        return TaylorPoly(std::move(a), {});
    }
    static TaylorPoly zero() { return from_scalar(S::zero()); }  // mt:639-641
    static TaylorPoly one() { return from_scalar(S::one()); }    // mt:649-651

    usize num_vars() const { return degrees_p1.size(); }  // mt:48-51
    bool is_constant() const { return coeffs.len() == 1; }  // mt:68-70
    usize len_of(usize v) const { return v < degrees_p1.size() ? degrees_p1[v] : UMAX; }  // mt:72-79
    bool is_zero() const { return coeffs.len() == 1 && coeffs.data[0].is_zero(); }  // mt:643-645
    bool is_one() const { return coeffs.len() == 1 && coeffs.data[0].is_one(); }    // mt:653-655

    // derived PartialEq (mt:10): array equality (shape + elements) and degrees equality.
    bool operator==(const TaylorPoly& o) const {
        if (degrees_p1 != o.degrees_p1 || coeffs.shape != o.coeffs.shape) return false;
        for (usize i = 0; i < coeffs.data.size(); ++i)
            if (!(coeffs.data[i] == o.coeffs.data[i])) return false;
        return true;
    }

    // mt:81-89
    TaylorPoly extend_to_dim(usize ndim, usize degree_p1) const {
        TaylorPoly r = *this;
        if (r.coeffs.ndim() > ndim) panic("extend_to_dim: ndim too small");
        while (r.coeffs.shape.size() < ndim) r.coeffs.shape.push_back(1);
        r.degrees_p1.resize(ndim, degree_p1);
        r.check_invariants();
        return r;
    }

    // mt:91-112 (test-only in the reference): zero-extend the stored array to `new_size`.
    TaylorPoly extend(const std::vector<usize>& new_size) const {
        if (degrees_p1.size() > new_size.size()) panic("extend: too few dims");
        Arr<S> src = coeffs;
        while (src.shape.size() < new_size.size()) src.shape.push_back(1);
        for (usize v = 0; v < src.shape.size(); ++v)
            if (src.shape[v] > new_size[v]) panic("extend: shape exceeds new size");
        Arr<S> out(new_size);
        zip_each(out.view().lead_block(src.shape), src.cview(), [](S& a, const S& b) { a = b; });
        return TaylorPoly(std::move(out), new_size);
    }

    // mt:114-127
    static std::vector<usize> min_degrees_p1(const TaylorPoly& a, const TaylorPoly& b) {
        std::vector<usize> d(std::max(a.degrees_p1.size(), b.degrees_p1.size()), UMAX);
        for (usize v = 0; v < d.size(); ++v) {
            if (v < a.degrees_p1.size()) d[v] = std::min(d[v], a.degrees_p1[v]);
            if (v < b.degrees_p1.size()) d[v] = std::min(d[v], b.degrees_p1[v]);
        }
        return d;
    }
    // mt:129-148
    static std::vector<usize> max_shape(const TaylorPoly& a, const TaylorPoly& b) {
        std::vector<usize> sh(std::max(a.coeffs.ndim(), b.coeffs.ndim()), 1);
        for (usize v = 0; v < sh.size(); ++v) {
            if (v < a.coeffs.ndim()) sh[v] = std::max(sh[v], a.coeffs.shape[v]);
            if (v < b.coeffs.ndim()) sh[v] = std::max(sh[v], b.coeffs.shape[v]);
            if (v < a.degrees_p1.size()) sh[v] = std::min(sh[v], a.degrees_p1[v]);
            if (v < b.degrees_p1.size()) sh[v] = std::min(sh[v], b.degrees_p1[v]);
        }
        return sh;
    }
    // mt:150-170
    static std::vector<usize> sum_shape(const TaylorPoly& a, const TaylorPoly& b) {
        std::vector<usize> sh(std::max(a.coeffs.ndim(), b.coeffs.ndim()), 0);
        for (usize v = 0; v < sh.size(); ++v) {
            if (v < a.coeffs.ndim()) sh[v] += a.coeffs.shape[v] - 1;
            if (v < b.coeffs.ndim()) sh[v] += b.coeffs.shape[v] - 1;
            sh[v] += 1;
            if (v < a.degrees_p1.size()) sh[v] = std::min(sh[v], a.degrees_p1[v]);
            if (v < b.degrees_p1.size()) sh[v] = std::min(sh[v], b.degrees_p1[v]);
        }
        return sh;
    }

    // Replace coeffs by its leading block of shape `lens` (slice_axis_inplace 0..len).
    void slice_lead(const std::vector<usize>& lens) {
        if (lens == coeffs.shape) return;
        coeffs = Arr<S>::from_view(coeffs.cview().lead_block(lens));
    }

    // mt:172-181
    TaylorPoly remove_last_variable() const {
        if (num_vars() == 0) panic("remove_last_variable: no variables");
        usize v = num_vars() - 1;
        Arr<S> c = coeffs;
        if (v < c.ndim()) {
            // index_axis_inplace(Axis(v), 0) on the LAST axis: keep index 0 of it.
            std::vector<usize> lens = c.shape;
            lens[v] = 1;
            c = Arr<S>::from_view(c.cview().lead_block(lens));
            c.shape.pop_back();
        }
        std::vector<usize> d = degrees_p1;
        d.pop_back();
        return TaylorPoly(std::move(c), d);
    }

    // mt:183-193
    TaylorPoly truncate_to_degree_p1(usize degree_p1) const {
        TaylorPoly r = *this;
        std::vector<usize> lens = r.coeffs.shape;
        for (usize v = 0; v < r.num_vars(); ++v) {
            r.degrees_p1[v] = std::min(r.degrees_p1[v], degree_p1);
            if (v < r.coeffs.ndim() && r.coeffs.shape[v] > degree_p1) lens[v] = degree_p1;
        }
        r.slice_lead(lens);
        return r;
    }
    // mt:195-204
    void truncate_degrees_p1(const std::vector<usize>& degs) {
        std::vector<usize> lens = coeffs.shape;
        for (usize v = 0; v < num_vars(); ++v) {
            degrees_p1[v] = std::min(degrees_p1[v], degs[v]);
            if (v < coeffs.ndim() && coeffs.shape[v] > degs[v]) lens[v] = degs[v];
        }
        slice_lead(lens);
    }

    // mt:208-216
    static TaylorPoly zero_with(const std::vector<usize>& degs) {
        Arr<S> a(std::vector<usize>(degs.size(), 1));
        return TaylorPoly(std::move(a), degs);
    }
    // mt:219-225
    static TaylorPoly from_u32_with(uint32_t c, const std::vector<usize>& degs) {
        Arr<S> a;
        a.data.push_back(S::from_u32(c));
        return TaylorPoly(std::move(a), degs);  // invariant requires degs empty, as in the reference
    }
    // mt:228-237
    static TaylorPoly var_at_zero(usize v, usize len) {
        std::vector<usize> sh(v + 1, 1);
        sh[v] = 2;
        Arr<S> a(sh);
        if (len > 1) a.data[1] = S::one();
        return TaylorPoly(std::move(a), std::vector<usize>(v + 1, len));
    }
    // mt:239-248
    static TaylorPoly var(usize v, S x, usize len) {
        std::vector<usize> sh(v + 1, 1);
        sh[v] = std::min<usize>(len, 2);
        Arr<S> a(sh);
        a.data[0] = x;
        if (len > 1) a.data[1] = S::one();
        return TaylorPoly(std::move(a), std::vector<usize>(v + 1, len));
    }
    // mt:250-259
    static TaylorPoly var_with_degrees_p1(usize v, S x, const std::vector<usize>& degs) {
        std::vector<usize> sh(degs.size(), 1);
        sh[v] = 2;
        Arr<S> a(sh);
        std::vector<usize> st = c_strides(sh);
        a.data[0] = x;
        if (degs[v] > 1) a.data[st[v]] = S::one();
        return TaylorPoly(std::move(a), degs);
    }

    // mt:262-269
    bool extract_constant(S& out) const {
        if (coeffs.len() == 1) {
            out = coeffs.data[0];
            return true;
        }
        return false;
    }

    // mt:275-294 — does not recognise constants.
    bool extract_linear(S& c, S& m, usize& var) const {
        std::vector<usize> st = c_strides(coeffs.shape);
        for (usize v = 0; v < coeffs.ndim(); ++v) {
            if (coeffs.shape[v] < 2) continue;
            bool ok = true;
            usize nd = coeffs.ndim();
            std::vector<usize> idx(nd, 0);
            for (usize lin = 0; lin < coeffs.len() && ok; ++lin) {
                bool others_zero = true;
                for (usize a = 0; a < nd; ++a)
                    if (a != v && idx[a] != 0) others_zero = false;
                bool is_first_of_slab = others_zero;  // first element of the slab view
                bool allowed_nonzero = (idx[v] <= 1) && is_first_of_slab;
                if (!allowed_nonzero && !coeffs.data[lin].is_zero()) ok = false;
                for (usize a = nd; a-- > 0;) {
                    if (++idx[a] < coeffs.shape[a]) break;
                    idx[a] = 0;
                }
            }
            if (ok) {
                c = coeffs.data[0];
                m = coeffs.data[st[v]];
                var = v;
                return true;
            }
        }
        return false;
    }

    S constant_term() const { return coeffs.data[0]; }  // mt:296-299

    // mt:314-339
    S coefficient(const std::vector<usize>& index) const {
        usize consumed = 0;  // number of leading axes indexed away
        usize off = 0;
        std::vector<usize> st = c_strides(coeffs.shape);
        for (usize v = 0; v < index.size(); ++v) {
            usize idx = index[v];
            if (!(idx < len_of(v))) panic("index out of bounds");
            if (v >= coeffs.ndim()) {
                if (idx != 0) return S::zero();
            } else if (idx >= coeffs.shape[v]) {
                return S::zero();
            } else {
                off += idx * st[v];
                consumed++;
            }
        }
        if (consumed != coeffs.ndim()) panic("index is too short");
        return coeffs.data[off];
    }

    // Helpers: slab range [lo,hi) along axis v as an owned array (axis kept).
    Arr<S> slab_range(usize v, usize lo, usize hi) const {
        View<const S> w = coeffs.cview();
        w.p += lo * w.stride[v];
        w.shape[v] = hi - lo;
        return Arr<S>::from_view(w);
    }

    // mt:341-358
    TaylorPoly coefficients_of_term(usize v, usize order) const {
        if (v >= coeffs.ndim()) {
            if (order == 0) return *this;
            return zero_with(degrees_p1);
        }
        if (order >= coeffs.shape[v]) return zero_with(degrees_p1);
        return TaylorPoly(slab_range(v, order, order + 1), degrees_p1);
    }

    // mt:380-404
    TaylorPoly taylor_polynomial_terms(usize v, const std::vector<usize>& orders) const {
        usize max_order_p1 = 1;
        if (!orders.empty()) max_order_p1 = *std::max_element(orders.begin(), orders.end()) + 1;
        if (v >= coeffs.ndim()) {
            if (std::find(orders.begin(), orders.end(), (usize)0) != orders.end()) return *this;
            return zero_with(degrees_p1);
        }
        usize upper = std::min(coeffs.shape[v], max_order_p1);
        Arr<S> result = slab_range(v, 0, upper);
        std::vector<bool> keep(max_order_p1, false);
        for (usize o : orders) keep[o] = true;
        View<S> rv = result.view();
        for (usize i = 0; i < upper; ++i) {
            if (!keep[i]) {
                View<S> s = rv;
                s.p += i * s.stride[v];
                s.shape[v] = 1;
                for_each(s, [](S& x) { x = S::zero(); });
            }
        }
        return TaylorPoly(std::move(result), degrees_p1);
    }

    // mt:457-481
    TaylorPoly derivative(usize v, usize n) const {
        if (!(v < num_vars() && n < len_of(v))) panic("derivative: bad var/order");
        if (v >= coeffs.ndim()) {
            if (n == 0) return *this;
            return zero_with(degrees_p1);
        }
        std::vector<usize> d = degrees_p1;
        d[v] = sat_sub(d[v], n);
        if (n >= coeffs.shape[v]) return zero_with(d);
        Arr<S> result = slab_range(v, n, coeffs.shape[v]);
        S ff = S::one();
        for (usize i = 1; i <= n; ++i) ff = ff * S::from_u32((uint32_t)i);
        View<S> rv = result.view();
        for (usize k = 0; k < result.shape[v]; ++k) {
            View<S> s = rv;
            s.p += k * s.stride[v];
            s.shape[v] = 1;
            for_each(s, [&](S& x) { x = x * ff; });
            ff = ff * (S::from_u32((uint32_t)(n + k + 1)) / S::from_u32((uint32_t)(k + 1)));
        }
        return TaylorPoly(std::move(result), d);
    }

    // mt:484-509
    TaylorPoly taylor_expansion_of_coeff(usize v, usize n) const {
        if (!(v < num_vars() && n < len_of(v))) panic("taylor_expansion_of_coeff: bad var/order");
        if (v >= coeffs.ndim()) {
            if (n == 0) return *this;
            return zero_with(degrees_p1);
        }
        std::vector<usize> d = degrees_p1;
        d[v] = sat_sub(d[v], n);
        if (n >= coeffs.shape[v]) return zero_with(d);
        Arr<S> result = slab_range(v, n, coeffs.shape[v]);
        S factor = S::one();
        View<S> rv = result.view();
        for (usize k = 1; k < result.shape[v]; ++k) {
            factor = factor * (S::from_u32((uint32_t)(n + k)) / S::from_u32((uint32_t)k));
            View<S> s = rv;
            s.p += k * s.stride[v];
            s.shape[v] = 1;
            for_each(s, [&](S& x) { x = x * factor; });
        }
        return TaylorPoly(std::move(result), d);
    }

    // ndarray 0.15.6 numeric_util::unrolled_fold (third-party; restated from its published
    // source): eight partial sums, combined as ((((0+(p0+p4))+(p1+p5))+(p2+p6))+(p3+p7)), tail added in order.
    static S unrolled_sum(const std::vector<S>& xs) {
        S acc = S::zero();
        S p[8];
        for (auto& q : p) q = S::zero();
        usize i = 0, n = xs.size();
        while (n - i >= 8) {
            for (usize u = 0; u < 8; ++u) p[u] = p[u] + xs[i + u];
            i += 8;
        }
        acc = acc + (p[0] + p[4]);
        acc = acc + (p[1] + p[5]);
        acc = acc + (p[2] + p[6]);
        acc = acc + (p[3] + p[7]);
        for (; i < n; ++i) acc = acc + xs[i];
        return acc;
    }

    // ndarray 0.15.6 ArrayBase::sum_axis (third-party; call sites mt:523,531), axis removed.
    // `unit_stride` says whether the summed axis has memory stride 1 in the reference's array
    // (true when it is the last axis of a standard-layout array, or every later axis has length 1).
    static Arr<S> sum_axis(const View<const S>& a, usize axis) {
        usize n = a.shape[axis];
        std::vector<usize> rshape;
        for (usize i = 0; i < a.ndim(); ++i)
            if (i != axis) rshape.push_back(a.shape[i]);
        Arr<S> res(rshape);
        bool unit_stride = true;
        for (usize i = axis + 1; i < a.ndim(); ++i)
            if (a.shape[i] != 1) unit_stride = false;
        if (a.ndim() == 2 && unit_stride) {
            usize other = 1 - axis;
            for (usize i = 0; i < a.shape[other]; ++i) {
                View<const S> lane = a;
                lane.p += i * lane.stride[other];
                lane.shape[other] = 1;
                res.data[i] = unrolled_sum(collect(lane));
            }
        } else {
            for (usize i = 0; i < n; ++i) {
                View<const S> slab = a;
                slab.p += i * slab.stride[axis];
                slab.shape.erase(slab.shape.begin() + axis);
                slab.stride.erase(slab.stride.begin() + axis);
                zip_each(res.view(), slab, [](S& r, const S& x) { r = r + x; });
            }
        }
        return res;
    }

    // mt:514-536
    TaylorPoly shift_down(usize v, usize n) const {
        if (!(v < num_vars() && n < len_of(v))) panic("shift_down: bad var/order");
        if (v >= coeffs.ndim()) return *this;
        std::vector<usize> d = degrees_p1;
        d[v] = sat_sub(d[v], n);
        Arr<S> result;
        if (coeffs.shape[v] <= n + 1) {
            result = sum_axis(coeffs.cview(), v);
            result.shape.insert(result.shape.begin() + v, 1);
        } else {
            result = slab_range(v, n, coeffs.shape[v]);
            View<const S> head = coeffs.cview();
            head.shape[v] = n;
            Arr<S> s = sum_axis(head, v);
            View<S> slab0 = result.view();
            slab0.shape.erase(slab0.shape.begin() + v);
            slab0.stride.erase(slab0.stride.begin() + v);
            zip_each(slab0, s.cview(), [](S& r, const S& x) { r = r + x; });
        }
        return TaylorPoly(std::move(result), d);
    }

    // mt:540-580
    TaylorPoly subst_var(usize v, const TaylorPoly& subst) const {
        if (v >= coeffs.ndim()) return *this;
        std::vector<usize> d = min_degrees_p1(*this, subst);
        if (subst.is_zero()) return TaylorPoly(slab_range(v, 0, 1), d);
        S c, m;
        usize w;
        if (subst.extract_linear(c, m, w)) {
            if (v == w && c.is_zero()) {
                S factor = S::one();
                std::vector<usize> lens = coeffs.shape;
                for (usize a = 0; a < lens.size(); ++a) lens[a] = std::min(lens[a], d[a]);
                Arr<S> result = Arr<S>::from_view(coeffs.cview().lead_block(lens));
                View<S> rv = result.view();
                for (usize i = 0; i < result.shape[v]; ++i) {
                    View<S> s = rv;
                    s.p += i * s.stride[v];
                    s.shape[v] = 1;
                    for_each(s, [&](S& x) { x = x * factor; });
                    factor = factor * m;
                }
                return TaylorPoly(std::move(result), d);
            }
        }
        TaylorPoly res = zero_with(d);
        Arr<S> cs = coeffs;
        while (cs.shape.size() < d.size()) cs.shape.push_back(1);
        for (usize i = cs.shape[v]; i-- > 0;) {
            View<const S> chunk = cs.cview();
            chunk.p += i * chunk.stride[v];
            chunk.shape[v] = 1;
            std::vector<usize> lens = chunk.shape;
            for (usize a = 0; a < lens.size(); ++a) lens[a] = std::min(lens[a], d[a]);
            TaylorPoly coeff(Arr<S>::from_view(chunk.lead_block(lens)), d);
            res = add(mul(res, subst), coeff);
        }
        return res;
    }

    // mt:589-608
    TaylorPoly mul_var(S m, usize v, const std::vector<usize>& shape, const std::vector<usize>& degs) const {
        usize upper = std::min(shape[v] - 1, coeffs.shape[v]);
        Arr<S> src = slab_range(v, 0, upper);
        for (auto& x : src.data) x = x * m;
        Arr<S> result(shape);
        std::vector<usize> lens = src.shape;
        for (usize a = 0; a < lens.size(); ++a) lens[a] = std::min(lens[a], shape[a]);
        View<S> dst = result.view();
        dst.p += 1 * dst.stride[v];
        dst.shape = lens;  // axis v: 1..=upper has `upper` entries == lens[v]
        if (numel(lens) > 0) zip_each(dst, src.cview().lead_block(lens), [](S& a, const S& b) { a = b; });
        return TaylorPoly(std::move(result), degs);
    }

    // mt:611-623
    TaylorPoly mul_linear(S c, S m, usize v, const std::vector<usize>& shape, const std::vector<usize>& degs) const {
        if (c.is_zero()) return mul_var(m, v, shape, degs);
        return add(mul_var(m, v, shape, degs), mul(*this, from_scalar(c)));
    }

    // mt:832-852
    static void broadcast(TaylorPoly& xs, TaylorPoly& ys) {
        if (xs.degrees_p1.size() < ys.degrees_p1.size())
            xs.degrees_p1.insert(xs.degrees_p1.end(), ys.degrees_p1.begin() + xs.degrees_p1.size(), ys.degrees_p1.end());
        else if (ys.degrees_p1.size() < xs.degrees_p1.size())
            ys.degrees_p1.insert(ys.degrees_p1.end(), xs.degrees_p1.begin() + ys.degrees_p1.size(), xs.degrees_p1.end());
        while (xs.coeffs.shape.size() < ys.coeffs.shape.size()) xs.coeffs.shape.push_back(1);
        while (ys.coeffs.shape.size() < xs.coeffs.shape.size()) ys.coeffs.shape.push_back(1);
    }

    // mt:854-882
    static TaylorPoly add(TaylorPoly self, TaylorPoly other) {
        std::vector<usize> rd = min_degrees_p1(self, other);
        broadcast(self, other);
        self.truncate_degrees_p1(rd);
        other.truncate_degrees_p1(rd);
        if (other.coeffs.len() == 1) {
            self.coeffs.data[0] = self.coeffs.data[0] + other.coeffs.data[0];
            return TaylorPoly(std::move(self.coeffs), rd);
        }
        if (self.coeffs.len() == 1) {
            other.coeffs.data[0] = other.coeffs.data[0] + self.coeffs.data[0];
            return TaylorPoly(std::move(other.coeffs), rd);
        }
        std::vector<usize> shape = max_shape(self, other);
        self.truncate_degrees_p1(shape);
        other.truncate_degrees_p1(shape);
        Arr<S> result(shape);
        zip_each(result.view().lead_block(self.coeffs.shape), self.coeffs.cview(), [](S& r, const S& x) { r = r + x; });
        zip_each(result.view().lead_block(other.coeffs.shape), other.coeffs.cview(), [](S& r, const S& x) { r = r + x; });
        return TaylorPoly(std::move(result), rd);
    }

    // mt:902-909
    static TaylorPoly neg(TaylorPoly self) {
        for (auto& x : self.coeffs.data) x = -x;
        return self;
    }

    // mt:911-937
    static TaylorPoly sub(TaylorPoly self, TaylorPoly other) {
        std::vector<usize> rd = min_degrees_p1(self, other);
        broadcast(self, other);
        self.truncate_degrees_p1(rd);
        other.truncate_degrees_p1(rd);
        if (other.coeffs.len() == 1) {
            self.coeffs.data[0] = self.coeffs.data[0] - other.coeffs.data[0];
            return TaylorPoly(std::move(self.coeffs), rd);
        }
        if (self.coeffs.len() == 1) {
            other.coeffs.data[0] = other.coeffs.data[0] - self.coeffs.data[0];
            for (auto& x : other.coeffs.data) x = -x;
            return TaylorPoly(std::move(other.coeffs), rd);
        }
        std::vector<usize> shape = max_shape(self, other);
        Arr<S> result(shape);
        zip_each(result.view().lead_block(self.coeffs.shape), self.coeffs.cview(), [](S& r, const S& x) { r = r + x; });
        zip_each(result.view().lead_block(other.coeffs.shape), other.coeffs.cview(), [](S& r, const S& x) { r = r - x; });
        return TaylorPoly(std::move(result), rd);
    }

    // mt:1014-1072
    static TaylorPoly mul(TaylorPoly self, TaylorPoly other) {
        std::vector<usize> d = min_degrees_p1(self, other);
        if (self.is_zero() || other.is_zero()) return zero_with(d);
        broadcast(self, other);
        std::vector<usize> shape = sum_shape(self, other);
        self.truncate_degrees_p1(d);
        other.truncate_degrees_p1(d);
        if (self.is_one()) return other;
        if (other.is_one()) return self;
        S c;
        if (self.extract_constant(c)) {
            for (auto& x : other.coeffs.data) x = c * x;
            return other;
        }
        if (other.extract_constant(c)) {
            for (auto& x : self.coeffs.data) x = c * x;
            return self;
        }
        S m;
        usize v;
        if (self.extract_linear(c, m, v)) {
            std::vector<usize> sh = other.coeffs.shape;
            sh[v] = std::min(d[v], sh[v] + 1);
            return other.mul_linear(c, m, v, sh, d);
        }
        if (other.extract_linear(c, m, v)) {
            std::vector<usize> sh = self.coeffs.shape;
            sh[v] = std::min(d[v], sh[v] + 1);
            return self.mul_linear(c, m, v, sh, d);
        }
        Arr<S> result(shape);
        mul_rec<S>(self.coeffs.cview(), other.coeffs.cview(), result.view());
        return TaylorPoly(std::move(result), d);
    }

    // mt:1162-1192
    static void div_rec(const View<const S>& xs, const View<const S>& ys, const View<S>& res) {
        if (xs.is_empty()) return;
        if (res.ndim() == 0) {
            res.first() = xs.first() / ys.first();
            return;
        }
        usize n0 = res.len_of(0);
        for (usize k = 0; k < n0; ++k) {
            View<S> current = res.index0(k);
            usize lo = sat_sub(k + 1, ys.len_of(0));
            for (usize j = lo; j < k; ++j)
                mul_rec<S>(as_const(res.index0(j)), ys.index0(k - j), current);
            for_each(current, [](S& x) { x = -x; });
            if (k < xs.len_of(0)) {
                View<const S> xk = xs.index0(k);
                zip_each(current.lead_block(xk.shape), xk, [](S& r, const S& x) { r = r + x; });
            }
            Arr<S> copy = Arr<S>::from_view(as_const(current));
            for_each(current, [](S& x) { x = S::zero(); });
            div_rec(copy.cview(), ys.index0(0), current);
        }
    }

    // mt:1194-1231
    static TaylorPoly div(TaylorPoly self, TaylorPoly other) {
        broadcast(self, other);
        std::vector<usize> d = min_degrees_p1(self, other);
        self.truncate_degrees_p1(d);
        other.truncate_degrees_p1(d);
        if (other.is_one()) return self;
        S c;
        if (other.extract_constant(c)) {
            for (auto& x : self.coeffs.data) x = x / c;
            return self;
        }
        std::vector<usize> rs = d;
        for (usize i = 0; i < rs.size(); ++i)
            if (other.coeffs.shape[i] == 1) rs[i] = self.coeffs.shape[i];
        Arr<S> result(rs);
        div_rec(self.coeffs.cview(), other.coeffs.cview(), result.view());
        return TaylorPoly(std::move(result), d);
    }

    // mt:1270-1283
    static std::vector<S> exp_1d(const std::vector<S>& xs, usize n) {
        std::vector<S> res(n, S::zero());
        res[0] = xs[0].exp();
        for (usize k = 1; k < n; ++k) {
            S sum = S::zero();
            usize hi = std::min(xs.size(), k + 1);
            for (usize j = 1; j < hi; ++j) sum = sum + xs[j] * S::from_u32((uint32_t)j) * res[k - j];
            res[k] = sum / S::from_u32((uint32_t)k);
        }
        return res;
    }

    // mt:1285-1317
    static void exp_rec(const View<const S>& xs, const View<S>& res) {
        if (xs.is_empty()) return;
        if (res.ndim() == 0) {
            res.first() = xs.first().exp();
            return;
        }
        usize n;
        if (extract_1d_len(res.shape, n)) {
            std::vector<S> out = exp_1d(collect(xs), n);
            usize i = 0;
            for_each(res, [&](S& z) { z = out[i++]; });
            return;
        }
        exp_rec(xs.index0(0), res.index0(0));
        for (usize k = 1; k < res.len_of(0); ++k) {
            View<S> current = res.index0(k);
            usize hi = std::min(xs.len_of(0), k + 1);
            for (usize j = 1; j < hi; ++j) {
                Arr<S> scaled = Arr<S>::from_view(xs.index0(j));
                for (auto& x : scaled.data) x = x * S::from_u32((uint32_t)j);
                mul_rec<S>(scaled.cview(), as_const(res.index0(k - j)), current);
            }
            for_each(current, [&](S& x) { x = x / S::from_u32((uint32_t)k); });
        }
    }

    // mt:406-417
    TaylorPoly exp() const {
        std::vector<usize> rs = degrees_p1;
        for (usize i = 0; i < rs.size(); ++i)
            if (coeffs.shape[i] == 1) rs[i] = 1;
        Arr<S> result(rs);
        exp_rec(coeffs.cview(), result.view());
        return TaylorPoly(std::move(result), degrees_p1);
    }

    // mt:1319-1333
    static std::vector<S> log_1d(const std::vector<S>& xs, usize n) {
        std::vector<S> res(n, S::zero());
        res[0] = xs[0].log();
        for (usize k = 1; k < n; ++k) {
            S sum = S::zero();
            usize lo = std::max<usize>(sat_sub(k + 1, xs.size()), 1);
            for (usize j = lo; j < k; ++j) sum = sum + xs[k - j] * res[j] * S::from_u32((uint32_t)j);
            S xk = k < xs.size() ? xs[k] : S::zero();
            res[k] = (xk * S::from_u32((uint32_t)k) - sum) / xs[0] / S::from_u32((uint32_t)k);
        }
        return res;
    }

    // mt:1335-1386
    static void log_rec(const View<const S>& xs, const View<S>& res) {
        if (xs.is_empty()) return;
        if (res.ndim() == 0) {
            res.first() = xs.first().log();
            return;
        }
        usize dummy;
        if (extract_1d_len(xs.shape, dummy)) {
            usize n;
            if (!extract_1d_len(res.shape, n)) panic("log: result is not 1-d where input is (unwrap on None)");
            std::vector<S> out = log_1d(collect(xs), n);
            usize i = 0;
            for_each(res, [&](S& z) { z = out[i++]; });
            return;
        }
        log_rec(xs.index0(0), res.index0(0));
        for (usize k = 1; k < res.len_of(0); ++k) {
            View<S> current = res.index0(k);
            usize lo = std::max<usize>(sat_sub(k + 1, xs.len_of(0)), 1);
            for (usize j = lo; j < k; ++j) {
                Arr<S> scaled = Arr<S>::from_view(as_const(res.index0(j)));
                for (auto& x : scaled.data) x = x * S::from_u32((uint32_t)j);
                mul_rec<S>(xs.index0(k - j), scaled.cview(), current);
            }
            for_each(current, [](S& x) { x = -x; });
            if (k < xs.len_of(0)) {
                View<const S> xk = xs.index0(k);
                zip_each(current.lead_block(xk.shape), xk,
                         [&](S& r, const S& x) { r = r + S::from_u32((uint32_t)k) * x; });
            }
            TaylorPoly num(Arr<S>::from_view(as_const(current)), current.shape);
            TaylorPoly den(Arr<S>::from_view(xs.index0(0)), current.shape);
            TaylorPoly q = div(num, den);
            zip_each(current, q.coeffs.cview(), [](S& r, const S& x) { r = x; });
            for_each(current, [&](S& x) { x = x / S::from_u32((uint32_t)k); });
        }
    }

    // mt:419-430
    TaylorPoly log() const {
        std::vector<usize> rs = degrees_p1;
        for (usize i = 0; i < rs.size(); ++i)
            if (coeffs.shape[i] == 1) rs[i] = 1;
        Arr<S> result(rs);
        log_rec(coeffs.cview(), result.view());
        return TaylorPoly(std::move(result), degrees_p1);
    }

    // mt:433-451 (including the redundant final squaring of `base`)
    TaylorPoly pow(uint32_t e) const {
        if (e == 0) return one();
        if (e == 1) return *this;
        TaylorPoly res = one();
        TaylorPoly base = *this;
        while (e > 0) {
            if (e & 1) res = mul(res, base);
            base = mul(base, base);
            e >>= 1;
        }
        return res;
    }
};

}  // namespace orc
