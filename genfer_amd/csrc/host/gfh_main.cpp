// Host interpreter, part 7: the driver (src/main.rs) — parse, translate, simplify, evaluate moments and
// probability masses through the TaylorPoly backend, and print exactly the reference's report
// (main.rs:256-645; F64 Display = ryu shortest round-trip, f64.rs:41-45).
//
// C entry point:  int gfh_run(const char* source, const char* flags, const char* backend_lib,
//                             const char* backend_prefix, char** out_text, char** out_timings_json)
// `backend_lib`/`backend_prefix` select the library that implements the TaylorPoly C ABI
// (include/gftaylor.h): the product passes libgftaylor.so + "gft_" (or "gfti_" with --bounds).
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <sstream>

#include "gfh_semantics.hpp"
#include "gfh_number.hpp"

using namespace gfh;

namespace {

struct Args {
    bool bounds = false, no_simplify_gf = false, no_timing = false, no_probs = false;
    size_t unroll = 8;
    bool has_limit = false;
    size_t limit = 0;
    bool print_gf = false;
    std::string json_path;           // --json <path> (main.rs:91-93)
    std::string model_name = "model";  // file stem of the program (main.rs:603); set by the `genfer` executable
};

Args parse_flags(const std::string& flags) {
    Args a;
    std::istringstream is(flags);
    std::string tok;
    auto next_num = [&](const std::string& name) -> size_t {
        std::string v;
        if (!(is >> v)) throw std::runtime_error("missing value for " + name);
        return (size_t)std::stoull(v);
    };
    while (is >> tok) {
        if (tok == "-b" || tok == "--bounds") a.bounds = true;
        else if (tok == "--no-simplify-gf") a.no_simplify_gf = true;
        else if (tok == "--no-timing") a.no_timing = true;
        else if (tok == "--no-probs") a.no_probs = true;
        else if (tok == "-u" || tok == "--unroll") a.unroll = next_num(tok);
        else if (tok == "-l" || tok == "--limit") { a.has_limit = true; a.limit = next_num(tok); }
        else if (tok.rfind("--limit=", 0) == 0) { a.has_limit = true; a.limit = std::stoull(tok.substr(8)); }
        else if (tok.rfind("--unroll=", 0) == 0) a.unroll = std::stoull(tok.substr(9));
        else if (tok == "--print-gf") a.print_gf = true;
        else if (tok == "--json") { if (!(is >> a.json_path)) throw std::runtime_error("missing value for --json"); }
        else if (tok.rfind("--json=", 0) == 0) a.json_path = tok.substr(7);
        else if (tok == "--model-name") { if (!(is >> a.model_name)) throw std::runtime_error("missing value for --model-name"); }
        else if (tok == "-r" || tok == "--rational" || tok == "-s" || tok == "--symbolic" || tok == "--big-float" || tok == "-p" ||
                 tok == "--precision" || tok == "--print-program")
            throw std::runtime_error("flag " + tok + " selects a part of the reference that is out of scope here (f64 / interval Taylor path only)");
        else throw std::runtime_error("unknown flag " + tok);
    }
    return a;
}

typedef std::chrono::steady_clock Clock;

// Rust's `{}` of an f64 (the durations of print_json, main.rs:629-640): shortest round-trip digits, never scientific
// notation, no trailing ".0".
std::string fmt_rust_f64(double x) {
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);
    std::string s(buf, r.ptr);
    bool neg = s[0] == '-';
    if (neg) s = s.substr(1);
    size_t epos = s.find('e');
    std::string digits;
    for (char c : s.substr(0, epos))
        if (c != '.') digits.push_back(c);
    int exp10 = std::stoi(s.substr(epos + 1));
    int kk = exp10 + 1;  // position of the decimal point relative to the digit string
    std::string out;
    if (x == 0.0) out = "0";
    else if (kk <= 0) out = "0." + std::string(-kk, '0') + digits;
    else if ((size_t)kk >= digits.size()) out = digits + std::string(kk - digits.size(), '0');
    else out = digits.substr(0, kk) + "." + digits.substr(kk);
    return neg ? "-" + out : out;
}

struct Report {
    std::ostringstream out;
    std::ostringstream err;
    Args args;
    double t_gf = 0, t_moments = 0, t_probs = 0, t_total = 0;
    // what print_json needs (main.rs:595-645)
    bool json_ok = false;
    std::string j_total, j_mean, j_variance, j_stddev, j_skewness, j_kurtosis, j_masses;
    void elapsed(Clock::time_point start, const char* text, double* store) {  // main.rs:579-593
        double e = std::chrono::duration<double>(Clock::now() - start).count();
        if (store) *store = e;
        if (args.no_timing) return;
        char buf[64];
        if (e < 0.001) snprintf(buf, sizeof buf, "%.6fs", e);
        else if (e < 0.01) snprintf(buf, sizeof buf, "%.5fs", e);
        else if (e < 0.1) snprintf(buf, sizeof buf, "%.4fs", e);
        else snprintf(buf, sizeof buf, "%.3fs", e);
        out << text << buf << "\n";
    }
};

std::string in_interval(const Interval& iv, bool print_intervals) {  // main.rs:291-299
    F64 x;
    if (iv.extract_point(x)) return "= " + x.str();
    if (!print_intervals) return "= " + iv.center().str();
    return "∈ [" + iv.lo.str() + ", " + iv.hi.str() + "]";
}

bool support_to_interval(const SupportSet& s, Interval& out) {  // support.rs:267-288
    if (s.kind == SupportSet::EMPTY) return false;
    if (s.kind == SupportSet::RANGE) {
        out = Interval::exact(F64::from_u32(s.start), s.end ? F64::from_u32(*s.end) : F64::infinity());
        return true;
    }
    int64_t n; uint64_t d;
    s.istart.to_ratio(n, d);
    F64 lo = F64::from_ratio((uint64_t)n, d);
    F64 hi = F64::infinity();
    if (!s.iend.is_infinite()) { s.iend.to_ratio(n, d); hi = F64::from_ratio((uint64_t)n, d); }
    out = Interval::exact(lo, hi);
    return true;
}

struct Moments { Interval total, mean, raw2nd, raw3rd, raw4th, variance, stddev, central3rd, central4th, skewness, kurtosis; };

Moments moments_to_struct(const Interval& total, const std::vector<Interval>& m) {  // main.rs:512-546
    Moments r;
    r.total = total;
    r.raw2nd = m[1]; r.raw3rd = m[2]; r.raw4th = m[3];
    auto cm = moments_to_central_moments(m);
    r.mean = cm.first;
    r.central3rd = cm.second[1]; r.central4th = cm.second[2];
    auto sm = central_to_standardized_moments(cm.second);
    r.variance = sm.first;
    r.skewness = sm.second[0]; r.kurtosis = sm.second[1];
    r.stddev = r.variance.sqrt();
    for (auto& x : m) if (!x.not_less_than(Interval::zero())) throw std::runtime_error("moments must be non-negative for distributions supported on the natural numbers");
    if (!r.variance.not_less_than(Interval::zero())) throw std::runtime_error("variance must be non-negative");
    if (!r.kurtosis.not_less_than(Interval::zero())) throw std::runtime_error("kurtosis must be non-negative");
    return r;
}

void print_moments(Report& R, const Moments& m, bool pi) {  // main.rs:548-577
    auto& o = R.out;
    o << "Total measure:             Z " << in_interval(m.total, pi) << "\n";
    o << "Expected value:            E " << in_interval(m.mean, pi) << "\n";
    o << "2nd raw moment:         μ'_2 " << in_interval(m.raw2nd, pi) << "\n";
    o << "3rd raw moment:         μ'_3 " << in_interval(m.raw3rd, pi) << "\n";
    o << "4th raw moment:         μ'_4 " << in_interval(m.raw4th, pi) << "\n";
    o << "Standard deviation:        σ " << in_interval(m.stddev, pi) << "\n";
    o << "Variance (2nd central):    V " << in_interval(m.variance, pi) << "\n";
    o << "3rd central moment:      μ_3 " << in_interval(m.central3rd, pi) << "\n";
    o << "4th central moment:      μ_4 " << in_interval(m.central4th, pi) << "\n";
    o << "Skewness (3rd std moment): S " << in_interval(m.skewness, pi) << "\n";
    o << "Kurtosis (4th std moment): K " << in_interval(m.kurtosis, pi) << "\n";
}

const size_t MAX_PROB_LIMIT = 1000;

std::vector<Interval> print_probs(Report& R, const Interval& rest, const Interval& total_without_rest, const std::vector<Interval>& moments,
                 const SupportSet& var_info, const SupportSet& rest_info, bool uses_observe,
                 const std::function<std::vector<Interval>(size_t)>& probs_fn, Clock::time_point probs_start) {  // main.rs:384-473
    auto& o = R.out;
    o << "\n";
    Interval total = (total_without_rest + rest).ensure_upper_bound(F64::one());
    size_t limit;
    uint32_t lo, hi;
    if (R.args.has_limit) limit = R.args.limit;
    else if (total.is_zero()) limit = 1;
    else if (var_info.finite_nonempty_range(lo, hi)) limit = (size_t)hi + 1;
    else {
        auto cm = moments_to_central_moments(moments);
        double c4root = std::sqrt(std::sqrt(cm.second[2].hi.to_f64()));
        double lim = std::ceil(cm.first.hi.to_f64() + 4.0 * c4root);
        if (std::isfinite(lim)) limit = std::min((size_t)lim + 1, MAX_PROB_LIMIT);
        else {
            o << "Failed to find a limit automatically due to non-finite moments.\n";
            o << "Please specify a limit manually with `--limit`.\n";
            o << "Using a limit of 2 for now.\n";
            limit = 2;
        }
    }
    o << "Computing probabilities up to " << limit << "...\n";
    bool is_normalized = !uses_observe || total.is_one();
    Interval mass_missing = total_without_rest;
    std::vector<Interval> probs = probs_fn(limit);
    bool pi = R.args.bounds || !rest.is_zero();
    for (size_t i = 0; i < limit; ++i) {
        Interval p = probs[i];
        mass_missing = mass_missing - p;
        if (rest_info.contains((uint32_t)i)) p = p + rest;
        if (p < Interval::zero() || p > Interval::one())
            throw std::runtime_error("p(" + std::to_string(i) + ") = " + p.str() + " is not a probability");
        p = p.ensure_lower_bound(F64::zero()).ensure_upper_bound(F64::one());
        probs[i] = p;
        if (is_normalized) o << "p(" << i << ") " << in_interval(p, pi) << "\n";
        else {
            Interval np = (p / total).ensure_lower_bound(F64::zero()).ensure_upper_bound(F64::one());
            o << "Unnormalized: p(" << i << ")     " << in_interval(p, pi) << "\n";
            o << "Normalized:   p(" << i << ") / Z " << in_interval(np, pi) << "\n";
        }
    }
    SupportSet up_to = SupportSet::range(0, (uint32_t)(limit - 1));
    if (!rest_info.is_subset_of(up_to)) mass_missing = mass_missing + rest;
    if (var_info.is_subset_of(up_to)) mass_missing = Interval::zero();
    F64 mm_un = mass_missing.hi.max(F64::zero()).min(F64::one());
    F64 mm_n = (mass_missing / total).hi.max(F64::zero()).min(F64::one());
    if (is_normalized) o << "p(n) <= " << mm_un.str() << " for all n >= " << limit << "\n";
    else {
        o << "Unnormalized: p(n)     <= " << mm_un.str() << " for all n >= " << limit << "\n";
        o << "Normalized:   p(n) / Z <= " << mm_n.str() << " for all n >= " << limit << "\n";
    }
    R.elapsed(probs_start, "Time to compute probability masses: ", &R.t_probs);
    return probs;
}

void print_moments_and_probs_interval(Report& R, const std::function<Interval()>& rest_fn,
                                      const std::function<std::pair<Interval, std::vector<Interval>>(size_t)>& moments_fn,
                                      const std::function<std::vector<Interval>(size_t)>& probs_fn, const SupportSet& var_info,
                                      const SupportSet& rest_info, bool uses_observe, Clock::time_point inference_start) {  // main.rs:301-382
    auto& o = R.out;
    o << "Support is a subset of: " << var_info.str() << "\n\n";
    o << "Computing moments...\n";
    Interval rest = rest_fn().ensure_lower_bound(F64::zero()).ensure_upper_bound(F64::one()).unite(F64::zero());
    auto moment_start = Clock::now();
    auto tm = moments_fn(5);
    Interval total = tm.first.ensure_lower_bound(F64::zero()).ensure_upper_bound(F64::one());
    Interval total_without_rest = total;
    Interval max_rest = Interval::one() - total_without_rest;
    rest = rest.ensure_upper_bound(max_rest.hi);
    total = (total + rest).ensure_upper_bound(F64::one());
    std::vector<Interval> moments;
    for (auto& x : tm.second) moments.push_back(x.ensure_lower_bound(F64::zero()));
    Interval range;
    if (support_to_interval(rest_info, range)) {
        for (size_t i = 0; i < moments.size(); ++i) {
            F64 added = rest.hi * range.hi.pow((uint32_t)i + 1);
            moments[i] = moments[i] + Interval::exact(F64::zero(), added);
        }
    }
    Moments ms = moments_to_struct(total, moments);
    ms.variance = ms.variance.ensure_lower_bound(F64::zero());
    ms.stddev = ms.stddev.ensure_lower_bound(F64::zero());
    ms.kurtosis = ms.kurtosis.ensure_lower_bound(F64::zero());
    print_moments(R, ms, R.args.bounds || !rest.is_zero());
    R.elapsed(moment_start, "Time to compute moments: ", &R.t_moments);
    std::vector<Interval> probs;
    if (!(R.args.no_probs || !var_info.is_discrete() || total.is_zero()))
        probs = print_probs(R, rest, total_without_rest, moments, var_info, rest_info, uses_observe, probs_fn, Clock::now());
    R.elapsed(inference_start, "Total inference time: ", &R.t_total);
    if (!R.args.json_path.empty()) {  // main.rs:364-382: point values (interval centres), only without loop bounds
        if (rest.is_zero()) {
            R.json_ok = true;
            R.j_total = ms.total.center().str(); R.j_mean = ms.mean.center().str(); R.j_variance = ms.variance.center().str();
            R.j_stddev = ms.stddev.center().str(); R.j_skewness = ms.skewness.center().str(); R.j_kurtosis = ms.kurtosis.center().str();
            for (auto& p : probs) R.j_masses += p.center().str() + ", ";
            double total_s = std::chrono::duration<double>(Clock::now() - inference_start).count();
            std::ostringstream j;  // the reference's literal layout, trailing commas and all (main.rs:617-633)
            j << "\n{\n    \"model\": \"" << R.args.model_name << "\",\n    \"system\": \"genfer\",\n    \"time_gf_translation\": "
              << fmt_rust_f64(R.t_gf) << ",\n    \"total\": " << R.j_total << ",\n    \"mean\": " << R.j_mean << ",\n    \"variance\": "
              << R.j_variance << ",\n    \"stddev\": " << R.j_stddev << ",\n    \"skewness\": " << R.j_skewness << ",\n    \"kurtosis\": "
              << R.j_kurtosis << ",\n    \"time_moments\": " << fmt_rust_f64(R.t_moments) << ",\n    \"masses\": [" << R.j_masses
              << "],\n    \"time_probs\": " << fmt_rust_f64(R.t_probs) << ",\n    \"time_infer\": " << fmt_rust_f64(total_s) << ",\n}\n";
            FILE* f = fopen(R.args.json_path.c_str(), "w");
            if (!f) throw std::runtime_error("failed to write JSON file");
            fputs(j.str().c_str(), f);
            fclose(f);
        } else {
            R.err << "Could not write JSON file because results are only bounds due to the presence of loops.\n";
        }
    }
}

template <class T>
GfTranslation<T> translate(Report& R, const Program& program) {  // main.rs:229-254
    auto start = Clock::now();
    GfTransformer<T> tr;
    tr.out = &R.out;
    tr.with_unroll(R.args.unroll);
    GfTranslation<T> t = tr.semantics(program);
    if (!R.args.no_simplify_gf) {
        t.gf = t.gf.simplify();
        t.rest = t.rest.simplify();
    }
    if (R.args.print_gf) {  // main.rs:248-251
        R.out << "Generating function:\n" << t.gf.str() << "\n\n";
        R.out << "Remaining mass:\n" << t.rest.str() << "\n\n";
    }
    R.elapsed(start, "Time to construct the generating function: ", &R.t_gf);
    return t;
}

void run_f64(Report& R, const Program& program) {  // main.rs:187-227 + 256-289
    auto start = Clock::now();
    bool uses_observe = program.uses_observe();
    GfTranslation<F64> t = translate<F64>(R, program);
    size_t res = program.result;
    print_moments_and_probs_interval(
        R, [&] { return Interval::precisely(t.rest.eval(std::vector<F64>(t.var_info.num_vars(), F64::zero()), 1).constant_term()); },
        [&](size_t limit) {
            auto tm = moments_taylor(t.gf, res, t.var_info, limit);
            std::vector<Interval> ms;
            for (auto& m : tm.second) ms.push_back(Interval::precisely(m));
            return std::make_pair(Interval::precisely(tm.first), ms);
        },
        [&](size_t limit) {
            std::vector<Interval> ps;
            for (auto& p : probs_taylor(t.gf, res, t.var_info, limit)) ps.push_back(Interval::precisely(p));
            return ps;
        },
        t.var_info[res], t.rest_info[res], uses_observe, start);
}

void run_interval(Report& R, const Program& program) {  // main.rs:145-185
    auto start = Clock::now();
    bool uses_observe = program.uses_observe();
    GfTranslation<Interval> t = translate<Interval>(R, program);
    size_t res = program.result;
    print_moments_and_probs_interval(
        R, [&] { return t.rest.eval(std::vector<Interval>(t.var_info.num_vars(), Interval::zero()), 1).constant_term(); },
        [&](size_t limit) { return moments_taylor(t.gf, res, t.var_info, limit); },
        [&](size_t limit) { return probs_taylor(t.gf, res, t.var_info, limit); }, t.var_info[res], t.rest_info[res], uses_observe, start);
}

char* dup(const std::string& s) {
    char* p = (char*)malloc(s.size() + 1);
    std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

}  // namespace

static std::string g_last_stderr;

extern "C" {

// What the last gfh_run would have written to stderr (the reference's eprintln! lines).
const char* gfh_last_stderr() { return g_last_stderr.c_str(); }

// Returns 0 on success (*out_text = the report, exactly the reference's stdout), non-zero on error
// (*out_text = the message: parse errors and reference panics).  Caller frees with gfh_free.
int gfh_run(const char* source, const char* flags, const char* backend_lib, const char* backend_prefix, char** out_text,
            char** out_timings_json) {
    Report R;
    try {
        R.args = parse_flags(flags ? flags : "");
        Program program = parse_program(source);
        auto api = Api::load(backend_lib, backend_prefix);
        if (R.args.bounds) {
            Poly<Interval>::bind(api);
            run_interval(R, program);
        } else {
            Poly<F64>::bind(api);
            run_f64(R, program);
        }
        g_last_stderr = R.err.str();
        *out_text = dup(R.out.str());
        if (out_timings_json) {
            std::ostringstream j;
            j << "{\"time_gf_translation\": " << R.t_gf << ", \"time_moments\": " << R.t_moments << ", \"time_probs\": " << R.t_probs
              << ", \"time_infer\": " << R.t_total << "}";
            *out_timings_json = dup(j.str());
        }
        return 0;
    } catch (const std::exception& e) {
        *out_text = dup(R.out.str() + std::string("error: ") + e.what() + "\n");
        if (out_timings_json) *out_timings_json = nullptr;
        return 1;
    }
}

void gfh_free(char* p) { free(p); }

// One raw Interval<F64> operation of the interpreter's own number type (gfh_number.hpp, interval.rs:117-234,
// 264-276): op 0 add, 1 sub, 2 mul, 3 div, 4 neg, 5 exp, 6 log.  Used by tests/test_interval_pins.py to check that
// the three independent restatements of interval.rs in this repository (this one, the kernels' gft_elem.hpp and the
// test oracle) agree bit for bit.
int gfh_interval_op(int op, const double* a, const double* b, double* out) {
    Interval x = Interval::load(a), y = b ? Interval::load(b) : Interval(), r;
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
    r.store(out);
    return 0;
}
}
