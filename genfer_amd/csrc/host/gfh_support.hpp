// Host interpreter, part 3: integer/rational support analysis — `SupportSet` (src/support.rs),
// `VarSupport` and `SupportTransformer` (src/semantics/support.rs).  Pure integer / exact-rational
// bookkeeping: must be bit-exact with the reference.  Rationals (rug/GMP in the reference) only bound
// continuous supports; 128-bit numerator/denominator with gcd normalisation is ample for program literals.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "gfh_ppl.hpp"

namespace gfh {

struct Rational {  // src/number/rational.rs (Frac | PosInf | NegInf | NaR)
    enum Kind { FRAC, POS_INF, NEG_INF, NAR } kind = FRAC;
    __int128 n = 0, d = 1;
    static __int128 gcd128(__int128 a, __int128 b) {
        if (a < 0) a = -a;
        if (b < 0) b = -b;
        while (b) { __int128 t = a % b; a = b; b = t; }
        return a;
    }
    static Rational frac(__int128 n, __int128 d) {
        if (d == 0) throw std::runtime_error("Rational: division by zero");
        if (d < 0) { n = -n; d = -d; }
        __int128 g = gcd128(n, d);
        if (g > 1) { n /= g; d /= g; }
        Rational r; r.n = n; r.d = d; return r;
    }
    static Rational from_int(uint64_t x) { return frac((__int128)x, 1); }
    static Rational from_ratio(uint64_t n, uint64_t d) { return frac((__int128)n, (__int128)d); }
    static Rational zero() { return frac(0, 1); }
    static Rational infinity() { Rational r; r.kind = POS_INF; return r; }
    static Rational neg_infinity() { Rational r; r.kind = NEG_INF; return r; }
    static Rational nar() { Rational r; r.kind = NAR; return r; }
    bool is_infinite() const { return kind == POS_INF || kind == NEG_INF; }
    int sign() const { return n > 0 ? 1 : (n < 0 ? -1 : 0); }
    // partial_cmp: -1, 0, 1 or 2 (= None)
    int cmp(const Rational& o) const {
        if (kind == FRAC && o.kind == FRAC) {
            __int128 l = n * o.d, r = o.n * d;
            return l < r ? -1 : (l > r ? 1 : 0);
        }
        if (kind != FRAC && o.kind != FRAC) {  // impl PartialOrd for Special (rational.rs:30-38)
            if ((kind == POS_INF && o.kind == POS_INF) || (kind == NEG_INF && o.kind == NEG_INF)) return 0;
            if (kind == NEG_INF && o.kind == POS_INF) return -1;
            return 2;
        }
        if (kind == NAR || o.kind == NAR) return 2;
        if (kind == FRAC) return o.kind == POS_INF ? -1 : 1;
        return kind == NEG_INF ? -1 : 1;
    }
    bool operator==(const Rational& o) const {  // NaR != NaR (rational.rs:21-28)
        if (kind == NAR || o.kind == NAR) return false;
        return kind == o.kind && (kind != FRAC || (n == o.n && d == o.d));
    }
    bool operator<(const Rational& o) const { return cmp(o) == -1; }
    bool operator>(const Rational& o) const { return cmp(o) == 1; }
    bool operator<=(const Rational& o) const { int c = cmp(o); return c == -1 || c == 0; }
    bool operator>=(const Rational& o) const { int c = cmp(o); return c == 1 || c == 0; }
    Rational min(const Rational& o) const { return *this < o ? *this : o; }  // rational.rs:370-376
    Rational max(const Rational& o) const { return *this > o ? *this : o; }  // :379-385
    friend Rational operator+(const Rational& a, const Rational& b) {       // :178-193
        if (a.kind == FRAC && b.kind == FRAC) return frac(a.n * b.d + b.n * a.d, a.d * b.d);
        if (a.kind == NAR || b.kind == NAR || (a.kind == POS_INF && b.kind == NEG_INF) || (a.kind == NEG_INF && b.kind == POS_INF)) return nar();
        return a.kind != FRAC ? a : b;
    }
    friend Rational operator-(const Rational& a, const Rational& b) {       // :203-223
        if (a.kind == FRAC && b.kind == FRAC) return frac(a.n * b.d - b.n * a.d, a.d * b.d);
        if (a.kind == NAR || b.kind == NAR || (a.kind == POS_INF && b.kind == POS_INF) || (a.kind == NEG_INF && b.kind == NEG_INF)) return nar();
        if (a.kind == POS_INF || b.kind == NEG_INF) return infinity();
        return neg_infinity();
    }
    friend Rational operator*(const Rational& a, const Rational& b) {       // :233-262
        if (a.kind == FRAC && b.kind == FRAC) return frac(a.n * b.n, a.d * b.d);
        if (a.kind == NAR || b.kind == NAR) return nar();
        if (a.kind != FRAC && b.kind != FRAC) return a.kind == b.kind ? infinity() : neg_infinity();
        const Rational& inf = a.kind != FRAC ? a : b;
        const Rational& fr = a.kind != FRAC ? b : a;
        if (fr.sign() == 0) return nar();
        bool pos = (inf.kind == POS_INF) == (fr.sign() > 0);
        return pos ? infinity() : neg_infinity();
    }
    std::string str() const {                                                  // :79-90
        if (kind == NAR) return "(not a rational)";
        if (kind == POS_INF) return "∞";
        if (kind == NEG_INF) return "-∞";
        auto i128 = [](__int128 v) {
            if (v == 0) return std::string("0");
            bool neg = v < 0;
            if (neg) v = -v;
            std::string s;
            while (v) { s.push_back((char)('0' + (int)(v % 10))); v /= 10; }
            if (neg) s.push_back('-');
            std::reverse(s.begin(), s.end());
            return s;
        };
        return d == 1 ? i128(n) : i128(n) + "/" + i128(d);
    }
    bool to_ratio(int64_t& num, uint64_t& den) const {
        if (kind != FRAC) return false;
        num = (int64_t)n; den = (uint64_t)d; return true;
    }
};

// src/support.rs:11-16
struct SupportSet {
    enum Kind { EMPTY, RANGE, INTERVAL } kind = EMPTY;
    uint32_t start = 0;
    std::optional<uint32_t> end;   // RANGE
    Rational istart, iend;         // INTERVAL

    static SupportSet empty() { return SupportSet(); }
    static SupportSet range(uint32_t s, std::optional<uint32_t> e) { SupportSet r; r.kind = RANGE; r.start = s; r.end = e; return r; }
    static SupportSet zero() { return range(0, 0u); }
    static SupportSet point(uint32_t x) { return range(x, x); }
    static SupportSet naturals() { return range(0, std::nullopt); }
    static SupportSet interval(const Rational& s, const Rational& e) {  // :146-151
        if (s > e) return empty();
        SupportSet r; r.kind = INTERVAL; r.istart = s; r.iend = e; return r;
    }
    static SupportSet nonneg_reals() { return interval(Rational::zero(), Rational::infinity()); }
    static SupportSet from_range_excl(uint32_t s, uint32_t e) { return e <= s ? empty() : range(s, e - 1); }  // From<Range<u32>>
    static SupportSet from_range_incl(uint32_t s, uint32_t e) { return s > e ? empty() : range(s, e); }       // From<RangeInclusive<u32>>

    bool is_empty() const { return kind == EMPTY; }
    bool is_zero() const { return kind == RANGE && start == 0 && end && *end == 0; }
    bool is_discrete() const { return kind != INTERVAL; }
    bool operator==(const SupportSet& o) const {
        if (kind != o.kind) return false;
        if (kind == RANGE) return start == o.start && end == o.end;
        if (kind == INTERVAL) return istart == o.istart && iend == o.iend;
        return true;
    }
    bool finite_nonempty_range(uint32_t& lo, uint32_t& hi) const {  // :136-141
        if (kind != RANGE || !end) return false;
        lo = start; hi = *end; return true;
    }
    SupportSet join(const SupportSet& o) const {  // :55-115
        if (kind == EMPTY) return o;
        if (o.kind == EMPTY) return *this;
        if (kind == RANGE && o.kind == RANGE)
            return range(std::min(start, o.start), (end && o.end) ? std::optional<uint32_t>(std::max(*end, *o.end)) : std::nullopt);
        if (kind == INTERVAL && o.kind == INTERVAL) { SupportSet r; r.kind = INTERVAL; r.istart = istart.min(o.istart); r.iend = iend.max(o.iend); return r; }
        if (kind == RANGE) {
            SupportSet r; r.kind = INTERVAL;
            r.istart = Rational::from_int(start).min(o.istart);
            r.iend = end ? Rational::from_int(*end).max(o.iend) : Rational::infinity();
            return r;
        }
        SupportSet r; r.kind = INTERVAL;
        r.istart = istart.min(Rational::from_int(o.start));
        r.iend = o.end ? iend.max(Rational::from_int(*o.end)) : Rational::infinity();
        return r;
    }
    SupportSet saturating_sub(uint32_t x) const {  // :117-130
        if (kind == EMPTY) return *this;
        if (kind == RANGE) return range(start > x ? start - x : 0, end ? std::optional<uint32_t>(*end > x ? *end - x : 0) : std::nullopt);
        SupportSet r; r.kind = INTERVAL;
        r.istart = (istart - Rational::from_int(x)).max(Rational::zero());
        r.iend = (iend - Rational::from_int(x)).max(Rational::zero());
        return r;
    }
    bool is_subset_of(const SupportSet& o) const {  // :157-187
        if (kind == EMPTY) return true;
        if (o.kind == EMPTY) return false;
        if (kind == INTERVAL && o.kind == RANGE) return false;
        if (kind == RANGE && o.kind == RANGE) return start >= o.start && (!o.end || (end && *end <= *o.end));
        if (kind == INTERVAL && o.kind == INTERVAL) return istart >= o.istart && iend <= o.iend;
        return Rational::from_int(start) >= o.istart && end && Rational::from_int(*end) <= o.iend;
    }
    void retain_only(std::vector<uint32_t> set) {  // :195-226
        std::sort(set.begin(), set.end());
        if (kind != RANGE) return;
        std::optional<uint32_t> ns, ne;
        for (uint32_t v : set)
            if (start <= v && v <= end.value_or(UINT32_MAX)) { if (!ns) ns = v; ne = v; }
        if (ns) *this = range(*ns, ne); else *this = empty();
    }
    void remove_all(std::vector<uint32_t> set) {  // :228-265
        std::sort(set.begin(), set.end());
        if (kind != RANGE || set.empty()) return;
        for (uint32_t v : set)
            if (v == start) start = v + 1;
        if (end) {
            for (auto it = set.rbegin(); it != set.rend(); ++it) {
                if (*it == *end) {
                    if (*it == 0) { end = 0u; start = 1; }
                    else end = *it - 1;
                }
            }
        }
        if (start > end.value_or(UINT32_MAX)) *this = empty();
    }
    bool contains(uint32_t i) const {  // :290-299
        if (kind == EMPTY) return false;
        if (kind == RANGE) return i >= start && (!end || i <= *end);
        Rational r = Rational::from_int(i);
        return r >= istart && r <= iend;
    }
    friend SupportSet operator+(const SupportSet& a, const SupportSet& b) {  // :380-439
        if (a.kind == EMPTY) return b;
        if (b.kind == EMPTY) return a;
        if (a.kind == RANGE && b.kind == RANGE) {
            uint64_t s = (uint64_t)a.start + b.start;
            std::optional<uint32_t> e;
            if (a.end && b.end) { uint64_t t = (uint64_t)*a.end + *b.end; if (t <= UINT32_MAX) e = (uint32_t)t; }
            return range(s > UINT32_MAX ? UINT32_MAX : (uint32_t)s, e);
        }
        SupportSet r; r.kind = INTERVAL;
        if (a.kind == INTERVAL && b.kind == INTERVAL) { r.istart = a.istart + b.istart; r.iend = a.iend + b.iend; }
        else if (a.kind == RANGE) { r.istart = Rational::from_int(a.start) + b.istart; r.iend = a.end ? Rational::from_int(*a.end) + b.iend : Rational::infinity(); }
        else { r.istart = a.istart + Rational::from_int(b.start); r.iend = b.end ? a.iend + Rational::from_int(*b.end) : Rational::infinity(); }
        return r;
    }
    SupportSet times(uint32_t f) const {  // Mul<u32> :449-465
        if (kind == EMPTY) return *this;
        if (kind == RANGE) return range(start * f, end ? std::optional<uint32_t>(*end * f) : std::nullopt);
        SupportSet r; r.kind = INTERVAL; r.istart = istart * Rational::from_int(f); r.iend = iend * Rational::from_int(f); return r;
    }
    std::string str() const {  // Display :351-376
        if (kind == EMPTY) return "∅";
        if (kind == RANGE) {
            if (end) return *end == start ? "{" + std::to_string(start) + "}" : "{" + std::to_string(start) + ", ..., " + std::to_string(*end) + "}";
            return "{" + std::to_string(start) + ", ...}";
        }
        if (iend == Rational::infinity()) return "[" + istart.str() + ", ∞)";
        return "[" + istart.str() + ", " + iend.str() + "]";
    }
};

// Distribution::support (src/ppl.rs:213-241)
inline SupportSet dist_support(const Distribution& d) {
    switch (d.kind) {
        case Distribution::Dirac: {
            uint32_t a;
            if (d.p.as_integer(a)) return SupportSet::point(a);
            return SupportSet::interval(Rational::from_ratio(d.p.numer, d.p.denom), Rational::from_ratio(d.p.numer, d.p.denom));
        }
        case Distribution::Bernoulli: case Distribution::BernoulliVarProb: return SupportSet::from_range_incl(0, 1);
        case Distribution::Binomial: return SupportSet::from_range_incl(0, d.n);
        case Distribution::Categorical: return SupportSet::from_range_excl(0, (uint32_t)d.ps.size());
        case Distribution::BinomialVarTrials: case Distribution::NegBinomialVarSuccesses: case Distribution::NegBinomial:
        case Distribution::Geometric: case Distribution::Poisson: case Distribution::PoissonVarRate: return SupportSet::naturals();
        case Distribution::Uniform: return SupportSet::from_range_excl(d.n, d.n2);
        case Distribution::Exponential: case Distribution::Gamma: return SupportSet::nonneg_reals();
        case Distribution::UniformCont:
            return SupportSet::interval(Rational::from_ratio(d.p.numer, d.p.denom), Rational::from_ratio(d.p2.numer, d.p2.denom));
    }
    return SupportSet::empty();
}

// src/semantics/support.rs:8-12
struct VarSupport {
    bool is_empty_ = true;
    size_t n_empty = 0;
    std::vector<SupportSet> prod;
    static VarSupport empty(size_t n) { VarSupport v; v.is_empty_ = true; v.n_empty = n; return v; }
    static VarSupport zero(size_t n) { VarSupport v; v.is_empty_ = false; v.prod.assign(n, SupportSet::zero()); return v; }
    static VarSupport from_vec(std::vector<SupportSet> s) { VarSupport v; v.is_empty_ = false; v.prod = std::move(s); v.normalize(); return v; }
    size_t num_vars() const { return is_empty_ ? n_empty : prod.size(); }
    const SupportSet& operator[](size_t v) const { static const SupportSet e = SupportSet::empty(); return is_empty_ ? e : prod.at(v); }
    bool operator==(const VarSupport& o) const {
        if (is_empty_ != o.is_empty_) return false;
        return is_empty_ ? n_empty == o.n_empty : prod == o.prod;
    }
    void push(const SupportSet& s) { if (is_empty_) n_empty++; else prod.push_back(s); }
    void normalize() {
        if (is_empty_) return;
        for (auto& s : prod)
            if (s.is_empty()) { size_t n = prod.size(); *this = empty(n); return; }
    }
    bool is_subset_of(const VarSupport& o) const {
        if (is_empty_) return true;
        if (o.is_empty_) return false;
        for (size_t i = 0; i < prod.size(); ++i)
            if (!prod[i].is_subset_of(o.prod[i])) return false;
        return true;
    }
    VarSupport join(const VarSupport& o) const {
        if (is_empty_) return o;
        if (o.is_empty_) return *this;
        std::vector<SupportSet> v;
        for (size_t i = 0; i < prod.size(); ++i) v.push_back(prod[i].join(o.prod.at(i)));
        return from_vec(std::move(v));
    }
    template <class F> void update(size_t v, F&& f) { if (!is_empty_) f(prod.at(v)); normalize(); }
    void set(size_t v, const SupportSet& s) { update(v, [&](SupportSet& x) { x = s; }); }
};

// src/semantics/support.rs:148-386
struct SupportTransformer {
    size_t unroll = 0;

    VarSupport init(const Program& p) { return VarSupport::zero(p.used_vars().num_vars()); }

    std::pair<VarSupport, VarSupport> transform_event(const Event& e, VarSupport init) {
        switch (e.kind) {
            case Event::InSet: {
                std::vector<uint32_t> set(e.set.begin(), e.set.end());
                VarSupport t = init, f = init;
                t.update(e.var, [&](SupportSet& s) { s.retain_only(set); });
                f.update(e.var, [&](SupportSet& s) { s.remove_all(set); });
                return {t, f};
            }
            case Event::DataFromDist: case Event::VarComparison: return {init, init};
            case Event::Complement: { auto r = transform_event(*e.sub[0], init); return {r.second, r.first}; }
            case Event::Intersection: {
                VarSupport els = VarSupport::empty(init.num_vars()), then = init;
                for (auto& s : e.sub) { auto r = transform_event(*s, then); then = r.first; els = els.join(r.second); }
                return {then, els};
            }
        }
        return {init, init};
    }

    VarSupport transform_statements(const std::vector<Statement>& stmts, VarSupport cur) {
        for (auto& s : stmts) cur = transform_statement(s, cur);
        return cur;
    }

    VarSupport transform_statement(const Statement& st, VarSupport init) {
        switch (st.kind) {
            case Statement::Sample: return transform_distribution(st.dist, st.var, init, st.add_previous_value);
            case Statement::Assign: {
                SupportSet ns = init[st.var];
                if (!st.add_previous_value) ns = SupportSet::zero();
                if (st.has_addend) ns = ns + init[st.addend_var].times(st.addend_factor);
                ns = ns + SupportSet::point(st.offset);
                init.set(st.var, ns);
                return init;
            }
            case Statement::Decrement: init.update(st.var, [&](SupportSet& s) { s = s.saturating_sub(st.offset); }); return init;
            case Statement::IfThenElse: {
                auto r = transform_event(*st.cond, init);
                return transform_statements(st.then, r.first).join(transform_statements(st.els, r.second));
            }
            case Statement::While: {
                size_t count = st.has_unroll ? st.unroll : unroll;
                size_t iters; VarSupport a, b;
                if (find_unroll_fixpoint(*st.cond, st.then, init, iters, a, b)) count = std::max(count, iters);
                VarSupport pre = init, rest = VarSupport::empty(init.num_vars());
                for (size_t i = 0; i < count; ++i) { auto r = one_iteration(pre, st.then, *st.cond); rest = rest.join(r.second); pre = r.first; }
                VarSupport inv = find_while_invariant(*st.cond, st.then, pre);
                auto r = transform_event(*st.cond, inv);
                return rest.join(r.second);
            }
            case Statement::Fail: return VarSupport::empty(init.num_vars());
            case Statement::Normalize: return transform_normalize(st.given_vars, 0, st.then, init);
        }
        return init;
    }

    static VarSupport transform_distribution(const Distribution& d, size_t v, VarSupport init, bool add_prev) {
        if (v == init.num_vars()) init.push(SupportSet::zero());
        if (!(v < init.num_vars())) throw std::runtime_error("assertion failed: v.id() < result.num_vars()");
        if (!add_prev) init.set(v, SupportSet::zero());
        SupportSet ds = dist_support(d);
        init.update(v, [&](SupportSet& s) { s = s + ds; });
        return init;
    }

    bool find_unroll_fixpoint(const Event& cond, const std::vector<Statement>& body, VarSupport pre, size_t& iters, VarSupport& pre_out, VarSupport& rest_out) {
        VarSupport rest = VarSupport::empty(pre.num_vars());
        for (size_t i = 0; i < 100; ++i) {
            auto r = one_iteration(pre, body, cond);
            rest = rest.join(r.second);
            if (pre == r.first) { iters = i; pre_out = pre; rest_out = rest; return true; }
            pre = r.first;
        }
        return false;
    }

    VarSupport find_while_invariant(const Event& cond, const std::vector<Statement>& body, VarSupport pre) {
        for (int i = 0; i < 100; ++i) {
            auto r = one_iteration(pre, body, cond);
            if (r.first.is_subset_of(pre)) return pre;
            pre = pre.join(r.first);
        }
        for (size_t i = 0; i <= 2 * pre.num_vars(); ++i) {
            auto r = one_iteration(pre, body, cond);
            if (r.first.is_subset_of(pre)) return pre;
            for (size_t v = 0; v < pre.num_vars(); ++v) pre.set(v, widen(pre[v], r.first[v]));
        }
        auto r = one_iteration(pre, body, cond);
        if (!r.first.is_subset_of(pre)) throw std::runtime_error("Widening failed.");
        return pre;
    }

    static SupportSet widen(const SupportSet& cur, const SupportSet& nw) {
        if (cur.kind != SupportSet::RANGE || nw.kind != SupportSet::RANGE) throw std::runtime_error("Cannot widen non-range supports");
        uint32_t s = cur.start <= nw.start ? cur.start : 0;
        std::optional<uint32_t> e;
        if (cur.end && nw.end && *nw.end <= *cur.end) e = cur.end;
        return SupportSet::range(s, e);
    }

    std::pair<VarSupport, VarSupport> one_iteration(VarSupport init, const std::vector<Statement>& body, const Event& cond) {
        auto r = transform_event(cond, init);
        return {transform_statements(body, r.first), r.second};
    }

    VarSupport transform_normalize(const std::vector<size_t>& given, size_t idx, const std::vector<Statement>& block, VarSupport vi) {
        if (idx == given.size()) return transform_statements(block, vi);
        size_t v = given[idx];
        uint32_t lo, hi;
        if (!vi[v].finite_nonempty_range(lo, hi))
            throw std::runtime_error("Cannot normalize with respect to variable `" + var_name(v) + "`, because its value could not be proven to be bounded.");
        VarSupport joined = VarSupport::empty(vi.num_vars());
        for (uint32_t i = lo;; ++i) {
            VarSupport nv = vi;
            nv.set(v, SupportSet::point(i));
            joined = joined.join(transform_normalize(given, idx + 1, block, nv));
            if (i == hi) break;
        }
        return joined;
    }
};

}  // namespace gfh
