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

}  // namespace gftfmt
