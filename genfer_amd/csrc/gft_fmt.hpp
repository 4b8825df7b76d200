// Text form of an f64 as the reference prints it (`impl Display for F64`, src/number/f64.rs:41-45: ryu 1.0.15), shared by
// the host interpreter's reports and by gft_format (the C ABI's `Display for TaylorPoly`).
#pragma once
#include <charconv>
#include <cmath>
#include <string>

namespace gftfmt {

// ryu::Buffer::format(f64): shortest round-trip digits; fixed notation when the decimal point position
// kk satisfies -5 < kk <= 16, else scientific d.ddde[-]x (ryu 1.0.15 `format64`).
inline std::string fmt_f64(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    if (x == 0.0) return std::signbit(x) ? "-0.0" : "0.0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);
    std::string s(buf, r.ptr);  // [-]d[.ddd]e[+-]XX, shortest digits
    bool neg = s[0] == '-';
    if (neg) s = s.substr(1);
    size_t epos = s.find('e');
    std::string mant = s.substr(0, epos);
    int exp10 = std::stoi(s.substr(epos + 1));
    std::string digits;
    for (char c : mant)
        if (c != '.') digits.push_back(c);
    int len = (int)digits.size();
    int k = exp10 - (len - 1);  // value = digits * 10^k
    int kk = len + k;           // position of the decimal point
    std::string out;
    if (0 <= k && kk <= 16) {
        out = digits + std::string(k, '0') + ".0";
    } else if (0 < kk && kk <= 16) {
        out = digits.substr(0, kk) + "." + digits.substr(kk);
    } else if (-5 < kk && kk <= 0) {
        out = "0." + std::string(-kk, '0') + digits;
    } else if (len == 1) {
        out = digits + "e" + std::to_string(kk - 1);
    } else {
        out = digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
    }
    return neg ? "-" + out : out;
}

// `impl Display for ArrayBase` of ndarray 0.15.6 (arrayformat.rs) — what `impl Debug for TaylorPoly` prints for
// `self.coeffs` (multivariate_taylor.rs:632-636).  Nested brackets, elements through their own Display, rows separated
// by ",\n" + one blank line per extra dimension + one space of indent per depth; arrays of 500 elements or more are
// abbreviated with "..." (6 leading sub-arrays / 11 rows / 11 columns at most, half from each end); an array with a
// zero-length axis prints as ndim empty bracket pairs; a 0-dimensional array prints its element.
template <class ElemFn>
inline void fmt_ndarray_rec(std::string& out, const size_t* shape, size_t ndim, size_t depth, size_t full_ndim, size_t base, const size_t limits[3],
                            ElemFn& elem) {
    if (ndim == 0) {
        out += elem(base);
        return;
    }
    size_t inner = 1;
    for (size_t i = 1; i < ndim; ++i) inner *= shape[i];
    const size_t rindex = full_ndim - depth - 1;  // 0 = last axis
    const size_t limit = limits[rindex == 0 ? 2 : (rindex == 1 ? 1 : 0)];
    std::string sep = ", ";
    if (ndim > 1) sep = ",\n" + std::string(ndim - 2, '\n') + std::string(depth + 1, ' ');
    auto item = [&](size_t i) {
        if (ndim == 1) out += elem(base + i);
        else fmt_ndarray_rec(out, shape + 1, ndim - 1, depth + 1, full_ndim, base + i * inner, limits, elem);
    };
    const size_t len = shape[0];
    out += "[";
    if (len <= limit) {
        for (size_t i = 0; i < len; ++i) {
            if (i) out += sep;
            item(i);
        }
    } else {
        const size_t edge = limit / 2;
        item(0);
        for (size_t i = 1; i < edge; ++i) {
            out += sep;
            item(i);
        }
        out += sep;
        out += "...";
        for (size_t i = len - edge; i < len; ++i) {
            out += sep;
            item(i);
        }
    }
    out += "]";
}
template <class ElemFn>
inline std::string fmt_ndarray(const size_t* shape, size_t ndim, ElemFn elem) {
    size_t n = 1;
    for (size_t i = 0; i < ndim; ++i) n *= shape[i];
    std::string out;
    if (n == 0) return std::string(ndim, '[') + std::string(ndim, ']');
    const size_t NOLIM = (size_t)-1;
    const size_t many[3] = {6, 11, 11}, all[3] = {NOLIM, NOLIM, NOLIM};  // stacked / next-to-last / last axis
    fmt_ndarray_rec(out, shape, ndim, 0, ndim, 0, n < 500 ? all : many, elem);
    return out;
}

}  // namespace gftfmt
