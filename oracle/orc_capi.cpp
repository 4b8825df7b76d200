// TEST INFRASTRUCTURE ONLY — C API over the CPU oracle (oracle/taylor_oracle.hpp) so that
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it via ctypes.
// The entry points deliberately mirror include/gftaylor.h name-for-name (prefix `orc_` for
// f64, `orci_` for Interval<f64>) so one Python wrapper class serves both libraries.
//
// Scalars cross the boundary as `const double*` pointing at WIDTH doubles (1 for f64,
// {lo,hi} for intervals); coefficient data is plane-major: WIDTH planes of numel doubles.
#include <chrono>
#include <cstdio>
#include <string>

#include "taylor_oracle.hpp"

using namespace orc;

static thread_local std::string g_err;

template <class S> struct Tr;
template <> struct Tr<F64> {
    static constexpr int W = 1;
    static F64 load(const double* p) { return F64(p[0]); }
    static void store(F64 s, double* p) { p[0] = s.v; }
    static F64 load_plane(const double* d, usize n, usize i) { (void)n; return F64(d[i]); }
    static void store_plane(F64 s, double* d, usize n, usize i) { (void)n; d[i] = s.v; }
};
template <> struct Tr<Interval> {
    static constexpr int W = 2;
    static Interval load(const double* p) { return Interval(p[0], p[1]); }
    static void store(Interval s, double* p) { p[0] = s.lo; p[1] = s.hi; }
    static Interval load_plane(const double* d, usize n, usize i) { return Interval(d[i], d[n + i]); }
    static void store_plane(Interval s, double* d, usize n, usize i) { d[i] = s.lo; d[n + i] = s.hi; }
};

template <class S> using P = TaylorPoly<S>;

static std::vector<usize> vec(const size_t* p, size_t n) { return std::vector<usize>(p, p + n); }

#define ORC_TRY(expr)                         \
    try {                                     \
        return (expr);                        \
    } catch (const std::exception& e) {       \
        g_err = e.what();                     \
        return 0;                             \
    }

template <class S>
static P<S>* from_host(const double* data, const size_t* shape, const size_t* degs, size_t ndim) {
    Arr<S> a;
    a.shape = vec(shape, ndim);
    usize n = numel(a.shape);
    a.data.resize(n);
    for (usize i = 0; i < n; ++i) a.data[i] = Tr<S>::load_plane(data, n, i);
    return new P<S>(std::move(a), vec(degs, ndim));
}

// ORC_TRACE_SUBST=1 (diagnostics for the build's own design work, never on in tests): the zero pattern of the operand of every
// subst_var — how many coefficients are exactly zero, and how many non-zero ones each of the top slabs along v holds
template <class S>
static void trace_subst(const P<S>& a, size_t v, const P<S>& s) {
    static const bool on = getenv("ORC_TRACE_SUBST") != nullptr;
    if (!on || v >= a.coeffs.shape.size() || a.coeffs.len() < 8) return;
    const auto& sh = a.coeffs.shape;
    usize inner = 1;
    for (usize i = v + 1; i < sh.size(); ++i) inner *= sh[i];
    std::vector<usize> per(sh[v], 0);
    usize zeros = 0;
    for (usize lin = 0; lin < a.coeffs.len(); ++lin) {
        if (a.coeffs.data[lin].is_zero()) zeros++;
        else per[(lin / inner) % sh[v]]++;
    }
    std::string shs, tops;
    for (usize x : sh) shs += std::to_string(x) + "x";
    for (usize k = sh[v]; k-- > 0 && tops.size() < 60;) tops += std::to_string(per[k]) + " ";
    fprintf(stderr, "[orc subst] a %s v=%zu zeros=%zu/%zu top-slab nonzeros: %s| subst len %zu\n", shs.c_str(), v, zeros, a.coeffs.len(), tops.c_str(),
            s.coeffs.len());
}

// Display / Debug of a TaylorPoly (mt:632-636, 694-730).  Floats: shortest decimal that round-trips (searched with
// printf precision 1..17 — independent of the product's std::to_chars route), laid out like ryu's format64 (f64:41-45).
static std::string fmt_num(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    if (x == 0.0) return std::signbit(x) ? "-0.0" : "0.0";
    char buf[64];
    int prec = 1;
    for (; prec <= 17; ++prec) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, x);
        if (strtod(buf, nullptr) == x) break;
    }
    std::string s(buf);
    bool neg = s[0] == '-';
    if (neg) s = s.substr(1);
    size_t e = s.find('e');
    std::string digits;
    for (char c : s.substr(0, e))
        if (c != '.') digits.push_back(c);
    int exp10 = atoi(s.c_str() + e + 1), len = (int)digits.size(), k = exp10 - (len - 1), kk = len + k;
    std::string out;
    if (0 <= k && kk <= 16) out = digits + std::string(k, '0') + ".0";
    else if (0 < kk && kk <= 16) out = digits.substr(0, kk) + "." + digits.substr(kk);
    else if (-5 < kk && kk <= 0) out = "0." + std::string(-kk, '0') + digits;
    else if (len == 1) out = digits + "e" + std::to_string(kk - 1);
    else out = digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
    return neg ? "-" + out : out;
}
static std::string fmt_scalar(const F64& s) { return fmt_num(s.v); }
static std::string fmt_scalar(const Interval& s) { return "[" + fmt_num(s.lo) + ", " + fmt_num(s.hi) + "]"; }
// ndarray 0.15.6 `Display for ArrayBase` (arrayformat.rs format_array_inner / format_with_overflow): 0-d = the element;
// 1-d = "[a, b, c]"; n-d = sub-arrays separated by ",\n", (n-2) blank lines and (depth+1) spaces; from 500 elements on
// every axis is abbreviated to limit/2 items from each end around "..." (limits: last axis 11, next-to-last 11, others 6);
// any zero-length axis = "[[..]]" with nothing inside.
template <class S>
static void ndarray_walk(const P<S>& p, usize depth, usize offset, bool many, std::string& o) {
    const std::vector<usize>& shape = p.coeffs.shape;
    const usize nd = shape.size();
    if (depth == nd) {
        o += fmt_scalar(p.coeffs.data[offset]);
        return;
    }
    usize stride = 1;
    for (usize i = depth + 1; i < nd; ++i) stride *= shape[i];
    const usize from_last = nd - 1 - depth;
    const usize limit = !many ? (usize)-1 : (from_last <= 1 ? 11 : 6);
    std::string sep;
    if (from_last == 0) sep = ", ";
    else {
        sep = ",\n";
        for (usize i = 1; i < from_last; ++i) sep += "\n";
        for (usize i = 0; i <= depth; ++i) sep += " ";
    }
    std::vector<long> items;  // -1 = ellipsis
    if (shape[depth] <= limit) {
        for (usize i = 0; i < shape[depth]; ++i) items.push_back((long)i);
    } else {
        const usize edge = limit / 2;
        items.push_back(0);
        for (usize i = 1; i < edge; ++i) items.push_back((long)i);
        items.push_back(-1);
        for (usize i = shape[depth] - edge; i < shape[depth]; ++i) items.push_back((long)i);
    }
    o += "[";
    for (usize k = 0; k < items.size(); ++k) {
        if (k) o += sep;
        if (items[k] < 0) o += "...";
        else ndarray_walk(p, depth + 1, offset + (usize)items[k] * stride, many, o);
    }
    o += "]";
}
template <class S>
static std::string ndarray_display(const P<S>& p) {
    const usize nd = p.coeffs.shape.size();
    if (p.coeffs.data.empty()) return std::string(nd, '[') + std::string(nd, ']');
    std::string o;
    ndarray_walk(p, 0, 0, p.coeffs.data.size() >= 500, o);
    return o;
}
template <class S>
static std::string format_poly(const P<S>& p, bool debug) {
    std::string out;
    bool first = true;
    const std::vector<usize>& shape = p.coeffs.shape;
    std::vector<usize> idx(shape.size(), 0);
    for (usize lin = 0; lin < p.coeffs.data.size(); ++lin) {
        if (!p.coeffs.data[lin].is_zero()) {
            if (!first) out += " + ";
            first = false;
            out += fmt_scalar(p.coeffs.data[lin]);
            for (usize i = 0; i < idx.size(); ++i) {
                if (!idx[i]) continue;
                out += i < 26 ? std::string(1, (char)('a' + i)) : "x_" + std::to_string(i);
                if (idx[i] > 1) out += "^" + std::to_string(idx[i]);
            }
        }
        for (usize ax = idx.size(); ax-- > 0;) {
            if (++idx[ax] < shape[ax]) break;
            idx[ax] = 0;
        }
    }
    if (first) out = "0";
    if (!debug) return out;
    // `impl Debug` (multivariate_taylor.rs:632-636): "TaylorPoly({:?}, {})" of degrees_p1 and of the ndarray itself.
    std::string d = "TaylorPoly([";
    for (usize i = 0; i < p.degrees_p1.size(); ++i) d += (i ? ", " : "") + std::to_string(p.degrees_p1[i]);
    return d + "], " + ndarray_display(p) + ")";
}

#define DEFINE_API(PFX, S)                                                                                 \
    extern "C" {                                                                                           \
    const char* PFX##last_error() { return g_err.c_str(); }                                                \
    int PFX##width() { return Tr<S>::W; }                                                                  \
    void* PFX##from_host(const double* d, const size_t* sh, const size_t* dg, size_t nd) {                 \
        ORC_TRY((void*)from_host<S>(d, sh, dg, nd))                                                        \
    }                                                                                                      \
    void* PFX##scalar(const double* s) { ORC_TRY((void*)new P<S>(P<S>::from_scalar(Tr<S>::load(s)))) }     \
    void* PFX##from_u32(uint32_t c) { ORC_TRY((void*)new P<S>(P<S>::from_u32_with(c, {}))) }               \
    void* PFX##zero_with(const size_t* dg, size_t nd) { ORC_TRY((void*)new P<S>(P<S>::zero_with(vec(dg, nd)))) } \
    void* PFX##var(size_t v, const double* x, size_t len) {                                                \
        ORC_TRY((void*)new P<S>(P<S>::var(v, Tr<S>::load(x), len)))                                        \
    }                                                                                                      \
    void* PFX##var_at_zero(size_t v, size_t len) { ORC_TRY((void*)new P<S>(P<S>::var_at_zero(v, len))) }   \
    void* PFX##var_with_degrees_p1(size_t v, const double* x, const size_t* dg, size_t nd) {               \
        ORC_TRY((void*)new P<S>(P<S>::var_with_degrees_p1(v, Tr<S>::load(x), vec(dg, nd))))                \
    }                                                                                                      \
    void* PFX##clone(const void* p) { ORC_TRY((void*)new P<S>(*(const P<S>*)p)) }                          \
    void PFX##free(void* p) { delete (P<S>*)p; }                                                           \
    size_t PFX##num_vars(const void* p) { return ((const P<S>*)p)->num_vars(); }                           \
    size_t PFX##numel(const void* p) { return ((const P<S>*)p)->coeffs.len(); }                            \
    void PFX##shape(const void* p, size_t* out) {                                                          \
        const auto& s = ((const P<S>*)p)->coeffs.shape;                                                    \
        for (usize i = 0; i < s.size(); ++i) out[i] = s[i];                                                \
    }                                                                                                      \
    void PFX##degrees_p1(const void* p, size_t* out) {                                                     \
        const auto& s = ((const P<S>*)p)->degrees_p1;                                                      \
        for (usize i = 0; i < s.size(); ++i) out[i] = s[i];                                                \
    }                                                                                                      \
    int PFX##to_host(const void* p, double* out) {                                                         \
        const auto& d = ((const P<S>*)p)->coeffs.data;                                                     \
        for (usize i = 0; i < d.size(); ++i) Tr<S>::store_plane(d[i], out, d.size(), i);                   \
        return 0;                                                                                          \
    }                                                                                                      \
    size_t PFX##len_of(const void* p, size_t v) { return ((const P<S>*)p)->len_of(v); }                    \
    int PFX##is_constant(const void* p) { return ((const P<S>*)p)->is_constant(); }                        \
    int PFX##is_zero(const void* p) { return ((const P<S>*)p)->is_zero(); }                                \
    int PFX##is_one(const void* p) { return ((const P<S>*)p)->is_one(); }                                  \
    int PFX##equal(const void* a, const void* b) { return *(const P<S>*)a == *(const P<S>*)b; }            \
    long PFX##format(const void* p, int debug, char* out, size_t cap) {                                    \
        std::string s = format_poly(*(const P<S>*)p, debug != 0);                                          \
        if (out && cap) {                                                                                  \
            size_t n = std::min(cap - 1, s.size());                                                        \
            std::memcpy(out, s.data(), n);                                                                 \
            out[n] = 0;                                                                                    \
        }                                                                                                  \
        return (long)s.size();                                                                             \
    }                                                                                                      \
    int PFX##constant_term(const void* p, double* out) {                                                   \
        Tr<S>::store(((const P<S>*)p)->constant_term(), out);                                              \
        return 0;                                                                                          \
    }                                                                                                      \
    int PFX##extract_constant(const void* p, double* out) {                                                \
        S c;                                                                                               \
        if (!((const P<S>*)p)->extract_constant(c)) return 0;                                              \
        Tr<S>::store(c, out);                                                                              \
        return 1;                                                                                          \
    }                                                                                                      \
    int PFX##extract_linear(const void* p, double* c_out, double* m_out, size_t* v_out) {                  \
        S c, m;                                                                                            \
        usize v;                                                                                           \
        if (!((const P<S>*)p)->extract_linear(c, m, v)) return 0;                                          \
        Tr<S>::store(c, c_out);                                                                            \
        Tr<S>::store(m, m_out);                                                                            \
        *v_out = v;                                                                                        \
        return 1;                                                                                          \
    }                                                                                                      \
    int PFX##coefficient(const void* p, const size_t* idx, size_t n, double* out) {                        \
        try {                                                                                              \
            Tr<S>::store(((const P<S>*)p)->coefficient(vec(idx, n)), out);                                 \
            return 0;                                                                                      \
        } catch (const std::exception& e) {                                                                \
            g_err = e.what();                                                                              \
            return -1;                                                                                     \
        }                                                                                                  \
    }                                                                                                      \
    void* PFX##add(const void* a, const void* b) { ORC_TRY((void*)new P<S>(P<S>::add(*(const P<S>*)a, *(const P<S>*)b))) } \
    void* PFX##sub(const void* a, const void* b) { ORC_TRY((void*)new P<S>(P<S>::sub(*(const P<S>*)a, *(const P<S>*)b))) } \
    void* PFX##mul(const void* a, const void* b) { ORC_TRY((void*)new P<S>(P<S>::mul(*(const P<S>*)a, *(const P<S>*)b))) } \
    void* PFX##div(const void* a, const void* b) { ORC_TRY((void*)new P<S>(P<S>::div(*(const P<S>*)a, *(const P<S>*)b))) } \
    void* PFX##neg(const void* a) { ORC_TRY((void*)new P<S>(P<S>::neg(*(const P<S>*)a))) }                \
    void* PFX##exp(const void* a) { ORC_TRY((void*)new P<S>(((const P<S>*)a)->exp())) }                    \
    void* PFX##log(const void* a) { ORC_TRY((void*)new P<S>(((const P<S>*)a)->log())) }                    \
    void* PFX##pow(const void* a, uint32_t e) { ORC_TRY((void*)new P<S>(((const P<S>*)a)->pow(e))) }       \
    void* PFX##derivative(const void* a, size_t v, size_t n) { ORC_TRY((void*)new P<S>(((const P<S>*)a)->derivative(v, n))) } \
    void* PFX##taylor_expansion_of_coeff(const void* a, size_t v, size_t n) {                              \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->taylor_expansion_of_coeff(v, n)))                        \
    }                                                                                                      \
    void* PFX##shift_down(const void* a, size_t v, size_t n) { ORC_TRY((void*)new P<S>(((const P<S>*)a)->shift_down(v, n))) } \
    void* PFX##derivative_truncated(const void* a, size_t v, size_t n, size_t d) {                         \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->derivative(v, n).truncate_to_degree_p1(d)))               \
    }                                                                                                      \
    void* PFX##observe_step(const void* a, size_t v, const double* x, const double* c, size_t d) {         \
        /* the unfused reference sequence, literally (generating_function.rs:684-689 evaluated by :557-566, :628-632) */ \
        ORC_TRY((void*)new P<S>(P<S>::mul(P<S>::mul(((const P<S>*)a)->derivative(v, 1).truncate_to_degree_p1(d),   \
                                                   P<S>::var(v, Tr<S>::load(x), d)),                            \
                                        P<S>::from_scalar(Tr<S>::load(c)))))                                  \
    }                                                                                                      \
    void* PFX##add_scaled(const void* a, const void* b, const double* c) {                                 \
        ORC_TRY((void*)new P<S>(P<S>::add(*(const P<S>*)a, P<S>::mul(*(const P<S>*)b, P<S>::from_scalar(Tr<S>::load(c)))))) \
    }                                                                                                      \
    void* PFX##observe_chain(const void* a, size_t v, const double* x, const double* cs, size_t n, size_t d) { \
        /* n times the unfused reference sequence, innermost first, each level one degree lower */           \
        try {                                                                                              \
            P<S> r = *(const P<S>*)a;                                                                      \
            for (size_t i = 0; i < n; ++i)                                                                 \
                r = P<S>::mul(P<S>::mul(r.derivative(v, 1).truncate_to_degree_p1(d + (n - 1 - i)),           \
                                        P<S>::var(v, Tr<S>::load(x), d + (n - 1 - i))),                     \
                              P<S>::from_scalar(Tr<S>::load(cs + i * Tr<S>::W)));                         \
            return (void*)new P<S>(r);                                                                     \
        } catch (const std::exception& e) {                                                                \
            g_err = e.what();                                                                              \
            return (void*)0;                                                                               \
        }                                                                                                  \
    }                                                                                                      \
    void* PFX##derive_scale(const void* a, size_t v, const double* c, size_t d) {                          \
        /* the unfused reference sequence (generating_function.rs:703-706 evaluated by :628-632 and Mul) */   \
        ORC_TRY((void*)new P<S>(P<S>::mul(((const P<S>*)a)->derivative(v, 1).truncate_to_degree_p1(d),       \
                                        P<S>::from_scalar(Tr<S>::load(c)))))                                  \
    }                                                                                                      \
    void* PFX##subst_var(const void* a, size_t v, const void* s) {                                         \
        trace_subst(*(const P<S>*)a, v, *(const P<S>*)s);                                                  \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->subst_var(v, *(const P<S>*)s)))                          \
    }                                                                                                      \
    void* PFX##coefficients_of_term(const void* a, size_t v, size_t o) {                                   \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->coefficients_of_term(v, o)))                             \
    }                                                                                                      \
    void* PFX##taylor_polynomial_terms(const void* a, size_t v, const size_t* orders, size_t n) {          \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->taylor_polynomial_terms(v, vec(orders, n))))             \
    }                                                                                                      \
    void* PFX##truncate_to_degree_p1(const void* a, size_t d) {                                            \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->truncate_to_degree_p1(d)))                               \
    }                                                                                                      \
    void* PFX##remove_last_variable(const void* a) { ORC_TRY((void*)new P<S>(((const P<S>*)a)->remove_last_variable())) } \
    void* PFX##extend_to_dim(const void* a, size_t nd, size_t d) {                                         \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->extend_to_dim(nd, d)))                                   \
    }                                                                                                      \
    void* PFX##extend(const void* a, const size_t* ns, size_t n) {                                         \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->extend(vec(ns, n))))                                     \
    }                                                                                                      \
    void* PFX##mul_var(const void* a, const double* m, size_t v, const size_t* sh, const size_t* dg, size_t n) { \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->mul_var(Tr<S>::load(m), v, vec(sh, n), vec(dg, n))))     \
    }                                                                                                      \
    void* PFX##mul_linear(const void* a, const double* c, const double* m, size_t v, const size_t* sh,     \
                          const size_t* dg, size_t n) {                                                    \
        ORC_TRY((void*)new P<S>(((const P<S>*)a)->mul_linear(Tr<S>::load(c), Tr<S>::load(m), v, vec(sh, n), vec(dg, n)))) \
    }                                                                                                      \
    }

DEFINE_API(orc_, F64)
DEFINE_API(orci_, Interval)

// ---------------------------------------------------------------------------------------
// Raw array entry points (no handles): the reference's general product `mul`
// (mt:984-1012) on caller-owned row-major buffers, and a timed variant used ONLY as
// bench.py's `cpu_baseline` leg.
// ---------------------------------------------------------------------------------------
extern "C" {

// res[k] += sum_j xs[j]*ys[k-j], identical loop nest / summation order to mt:971-1012.
int orc_mul_raw(const double* xs, const size_t* xshape, const double* ys, const size_t* yshape, double* res,
                const size_t* rshape, size_t ndim) {
    try {
        static_assert(sizeof(F64) == sizeof(double), "F64 must be layout-compatible with double");
        View<const F64> xv{reinterpret_cast<const F64*>(xs), vec(xshape, ndim), c_strides(vec(xshape, ndim))};
        View<const F64> yv{reinterpret_cast<const F64*>(ys), vec(yshape, ndim), c_strides(vec(yshape, ndim))};
        View<F64> rv{reinterpret_cast<F64*>(res), vec(rshape, ndim), c_strides(vec(rshape, ndim))};
        mul_rec<F64>(xv, yv, rv);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// Same product restricted to the leading-axis output slabs [k_lo, k_hi) (each slab is
// independent, mt:1001-1011), timed with a steady clock.  Returns seconds; writes the number of
// multiply-accumulates performed to *macs.  `res` must hold the full result shape.
double orc_mul_slabs_timed(const double* xs, const size_t* xshape, const double* ys, const size_t* yshape, double* res,
                           const size_t* rshape, size_t ndim, size_t k_lo, size_t k_hi, double* macs) {
    View<const F64> xv{reinterpret_cast<const F64*>(xs), vec(xshape, ndim), c_strides(vec(xshape, ndim))};
    View<const F64> yv{reinterpret_cast<const F64*>(ys), vec(yshape, ndim), c_strides(vec(yshape, ndim))};
    View<F64> rv{reinterpret_cast<F64*>(res), vec(rshape, ndim), c_strides(vec(rshape, ndim))};
    // MAC count of slab k: prod over axes of #valid j.
    auto pairs_1d = [](usize sx, usize sy, usize k) -> double {
        usize lo = sat_sub(k + 1, sy), hi = std::min(k + 1, sx);
        return hi > lo ? (double)(hi - lo) : 0.0;
    };
    double inner = 1.0;
    for (usize a = 1; a < ndim; ++a) {
        double s = 0.0;
        for (usize k = 0; k < rshape[a]; ++k) s += pairs_1d(xshape[a], yshape[a], k);
        inner *= s;
    }
    double total = 0.0;
    auto t0 = std::chrono::steady_clock::now();
    for (usize k = k_lo; k < k_hi && k < rshape[0]; ++k) {
        View<F64> z = rv.index0(k);
        usize lo = sat_sub(k + 1, yv.len_of(0));
        usize hi = std::min(k + 1, xv.len_of(0));
        for (usize j = lo; j < hi; ++j) mul_rec<F64>(xv.index0(j), yv.index0(k - j), z);
        total += pairs_1d(xshape[0], yshape[0], k) * inner;
    }
    auto t1 = std::chrono::steady_clock::now();
    *macs = total;
    return std::chrono::duration<double>(t1 - t0).count();
}

// The rows (k0, k1) for k1 in [k1_lo, k1_hi) of the same product — a slab cut along its second axis, so that the heaviest
// slabs of a 64^4 product can be spread over host threads by the tests.  Per output element the terms arrive exactly as
// in mt:984-1012: j0 ascending, inside it j1 ascending, each a recursive product of the remaining axes into z[k0][k1].
// Requires rank >= 3 and at least two non-unit result axes after the first (otherwise the reference's 1-d shortcut,
// mt:996-1000, applies to the whole slab and a row cut is not the same computation).  Returns 0, or -1 with orc_last_error.
int orc_mul_rows(const double* xs, const size_t* xshape, const double* ys, const size_t* yshape, double* res,
                 const size_t* rshape, size_t ndim, size_t k0, size_t k1_lo, size_t k1_hi) {
    try {
        if (ndim < 3) panic("orc_mul_rows: rank >= 3 required");
        usize nonunit = 0;
        for (usize a = 1; a < ndim; ++a) nonunit += rshape[a] != 1;
        if (nonunit < 2) panic("orc_mul_rows: the slab would take the 1-d shortcut");
        View<const F64> xv{reinterpret_cast<const F64*>(xs), vec(xshape, ndim), c_strides(vec(xshape, ndim))};
        View<const F64> yv{reinterpret_cast<const F64*>(ys), vec(yshape, ndim), c_strides(vec(yshape, ndim))};
        View<F64> rv{reinterpret_cast<F64*>(res), vec(rshape, ndim), c_strides(vec(rshape, ndim))};
        if (k0 >= rshape[0]) return 0;
        View<F64> slab = rv.index0(k0);
        usize lo0 = sat_sub(k0 + 1, yv.len_of(0)), hi0 = std::min(k0 + 1, xv.len_of(0));
        for (usize j0 = lo0; j0 < hi0; ++j0) {
            View<const F64> xj = xv.index0(j0), yj = yv.index0(k0 - j0);
            for (usize k1 = k1_lo; k1 < k1_hi && k1 < rshape[1]; ++k1) {
                View<F64> z = slab.index0(k1);
                usize lo1 = sat_sub(k1 + 1, yj.len_of(0)), hi1 = std::min(k1 + 1, xj.len_of(0));
                for (usize j1 = lo1; j1 < hi1; ++j1) mul_rec<F64>(xj.index0(j1), yj.index0(k1 - j1), z);
            }
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// The same row cut for Interval<F64> tensors (BASELINE configs[4] at its headline size: the tests spread the leading slabs
// {0, 63, 127} of the 128^3 interval product over the host's threads).  Layout: ARRAY OF INTERVALS — element i is
// {lo, hi} at doubles 2 i, 2 i + 1 (the oracle's own Arr<Interval>; the planes of the handle API would have to be
// re-packed by every one of the concurrent calls) — for operands and result alike.
int orci_mul_rows(const double* xs, const size_t* xshape, const double* ys, const size_t* yshape, double* res,
                  const size_t* rshape, size_t ndim, size_t k0, size_t k1_lo, size_t k1_hi) {
    static_assert(sizeof(Interval) == 2 * sizeof(double), "Interval is {lo, hi}");
    try {
        if (ndim < 3) panic("orci_mul_rows: rank >= 3 required");
        usize nonunit = 0;
        for (usize a = 1; a < ndim; ++a) nonunit += rshape[a] != 1;
        if (nonunit < 2) panic("orci_mul_rows: the slab would take the 1-d shortcut");
        View<const Interval> xv{reinterpret_cast<const Interval*>(xs), vec(xshape, ndim), c_strides(vec(xshape, ndim))};
        View<const Interval> yv{reinterpret_cast<const Interval*>(ys), vec(yshape, ndim), c_strides(vec(yshape, ndim))};
        View<Interval> rv{reinterpret_cast<Interval*>(res), vec(rshape, ndim), c_strides(vec(rshape, ndim))};
        if (k0 >= rshape[0]) return 0;
        View<Interval> slab = rv.index0(k0);
        usize lo0 = sat_sub(k0 + 1, yv.len_of(0)), hi0 = std::min(k0 + 1, xv.len_of(0));
        for (usize j0 = lo0; j0 < hi0; ++j0) {
            View<const Interval> xj = xv.index0(j0), yj = yv.index0(k0 - j0);
            for (usize k1 = k1_lo; k1 < k1_hi && k1 < rshape[1]; ++k1) {
                View<Interval> z = slab.index0(k1);
                usize lo1 = sat_sub(k1 + 1, yj.len_of(0)), hi1 = std::min(k1 + 1, xj.len_of(0));
                for (usize j1 = lo1; j1 < hi1; ++j1) mul_rec<Interval>(xj.index0(j1), yj.index0(k1 - j1), z);
            }
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// One raw Interval<F64> operation (iv:117-234, 264-276) on scalars, for the interval pin tests: op 0 add, 1 sub,
// 2 mul, 3 div, 4 neg, 5 exp, 6 log.
int orci_scalar_op(int op, const double* a, const double* b, double* out) {
    Interval x(a[0], a[1]), y = b ? Interval(b[0], b[1]) : Interval(), r;
    switch (op) {
        case 0: r = x + y; break;
        case 1: r = x - y; break;
        case 2: r = x * y; break;
        case 3: r = x / y; break;
        case 4: r = -x; break;
        case 5: r = x.exp(); break;
        case 6: r = x.log(); break;
        default: return -1;
    }
    out[0] = r.lo;
    out[1] = r.hi;
    return 0;
}
}
