// Host interpreter, part 4: SGCL abstract syntax (src/ppl.rs) and parser (src/parser.rs).  The parser
// is a hand-written recursive descent that reproduces the reference's nom grammar alternative by
// alternative (ordered choice with backtracking; `cut` = fatal error), including `loop n {}` unrolling
// at parse time (parser.rs:540-551) and exact decimal -> ratio conversion (parser.rs:41-68).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace gfh {

inline std::string var_name(size_t i) {  // ppl.rs:112-122
    if (i < 26) return std::string(1, (char)('a' + i));
    return "x_" + std::to_string(i);
}

struct PosRatio {  // ppl.rs:34-74
    uint64_t numer = 0, denom = 1;
    PosRatio() {}
    PosRatio(uint64_t n, uint64_t d) : numer(n), denom(d) {}
    PosRatio complement() const {
        if (!(numer <= denom)) throw std::runtime_error("assertion failed: self.numer <= self.denom");
        return PosRatio(denom - numer, denom);
    }
    bool as_integer(uint32_t& out) const {
        if (denom != 0 && numer % denom == 0) {
            uint64_t q = numer / denom;
            if (q <= UINT32_MAX) { out = (uint32_t)q; return true; }
        }
        return false;
    }
};

struct VarRange {  // ppl.rs:124-182
    size_t n = 0;
    static VarRange of(size_t var) { return VarRange{var + 1}; }
    VarRange unite(const VarRange& o) const { return VarRange{n > o.n ? n : o.n}; }
    VarRange add(size_t var) const { return unite(of(var)); }
    VarRange remove(size_t var) const { return var + 1 == n ? VarRange{var} : *this; }
    size_t num_vars() const { return n; }
};

struct Distribution {  // ppl.rs:184-211
    enum Kind { Dirac, Bernoulli, BernoulliVarProb, BinomialVarTrials, Binomial, Categorical, NegBinomialVarSuccesses,
                NegBinomial, Geometric, Poisson, PoissonVarRate, Uniform, Exponential, Gamma, UniformCont } kind = Dirac;
    PosRatio p, p2;             // p: prob / rate / lambda / start / shape; p2: end / rate (Gamma)
    uint32_t n = 0, n2 = 0;     // n: trials / successes / start; n2: end
    size_t var = 0;             // parameter variable
    std::vector<PosRatio> ps;   // Categorical
    VarRange used_vars() const {
        switch (kind) {
            case BernoulliVarProb: case BinomialVarTrials: case NegBinomialVarSuccesses: case PoissonVarRate: return VarRange::of(var);
            default: return VarRange{};
        }
    }
};

struct Event {  // ppl.rs:306-313
    enum Kind { InSet, VarComparison, DataFromDist, Complement, Intersection } kind = Intersection;
    enum Cmp { Eq, Lt, Le } cmp = Eq;
    size_t var = 0, var2 = 0;
    std::vector<uint32_t> set;
    uint32_t data = 0;
    Distribution dist;
    std::vector<std::shared_ptr<Event>> sub;

    typedef std::shared_ptr<Event> P;
    static P in_set(size_t v, std::vector<uint32_t> s) { auto e = std::make_shared<Event>(); e->kind = InSet; e->var = v; e->set = std::move(s); return e; }
    static P var_cmp(size_t a, Cmp c, size_t b) { auto e = std::make_shared<Event>(); e->kind = VarComparison; e->var = a; e->cmp = c; e->var2 = b; return e; }
    static P data_from(uint32_t d, const Distribution& dist) { auto e = std::make_shared<Event>(); e->kind = DataFromDist; e->data = d; e->dist = dist; return e; }
    static P complement(P e) {  // ppl.rs:358-364
        if (e->kind == Complement) return e->sub[0];
        auto r = std::make_shared<Event>(); r->kind = Complement; r->sub.push_back(e); return r;
    }
    static P intersection(std::vector<P> es) {  // ppl.rs:384-398
        std::vector<P> conj;
        for (auto& e : es) {
            if (e->kind == Intersection) conj.insert(conj.end(), e->sub.begin(), e->sub.end());
            else conj.push_back(e);
        }
        if (conj.size() == 1) return conj[0];
        auto r = std::make_shared<Event>(); r->kind = Intersection; r->sub = std::move(conj); return r;
    }
    static P disjunction(std::vector<P> es) {  // ppl.rs:400-406
        if (es.size() == 1) return es[0];
        std::vector<P> neg;
        for (auto& e : es) neg.push_back(complement(e));
        return complement(intersection(std::move(neg)));
    }
    static P always() { return intersection({}); }
    static P never() { return complement(always()); }

    VarRange used_vars() const {
        switch (kind) {
            case InSet: return VarRange::of(var);
            case VarComparison: return VarRange::of(var).add(var2);
            case DataFromDist: return dist.used_vars();
            case Complement: return sub[0]->used_vars();
            case Intersection: { VarRange r; for (auto& e : sub) r = r.unite(e->used_vars()); return r; }
        }
        return VarRange{};
    }
};

struct Statement {  // ppl.rs:449-487
    enum Kind { Sample, Assign, Decrement, IfThenElse, While, Fail, Normalize } kind = Fail;
    size_t var = 0;
    Distribution dist;
    bool add_previous_value = false;
    bool has_addend = false;
    uint32_t addend_factor = 1;
    size_t addend_var = 0;
    uint32_t offset = 0;
    Event::P cond;
    std::vector<Statement> then, els;  // then: also the body of While / Normalize
    bool has_unroll = false;
    size_t unroll = 0;
    std::vector<size_t> given_vars;

    bool uses_observe() const {  // ppl.rs:596-608
        switch (kind) {
            case Sample: case Assign: case Decrement: return false;
            case IfThenElse: for (auto& s : then) if (s.uses_observe()) return true;
                             for (auto& s : els) if (s.uses_observe()) return true; return false;
            case While: case Normalize: for (auto& s : then) if (s.uses_observe()) return true; return false;
            case Fail: return true;
        }
        return false;
    }
    VarRange used_vars() const {  // ppl.rs:610-644
        VarRange r;
        switch (kind) {
            case Sample: return dist.used_vars().add(var);
            case Assign: return VarRange::of(var).unite(has_addend ? VarRange::of(addend_var) : VarRange{});
            case Decrement: return VarRange::of(var);
            case IfThenElse: r = cond->used_vars(); for (auto& s : then) r = r.unite(s.used_vars()); for (auto& s : els) r = r.unite(s.used_vars()); return r;
            case While: r = cond->used_vars(); for (auto& s : then) r = r.unite(s.used_vars()); return r;
            case Fail: return r;
            case Normalize: for (auto& s : then) r = r.unite(s.used_vars()); return r;
        }
        return r;
    }
};

struct Program {  // ppl.rs:666-691
    std::vector<Statement> stmts;
    size_t result = 0;
    bool uses_observe() const { for (auto& s : stmts) if (s.uses_observe()) return true; return false; }
    VarRange used_vars() const { VarRange r; for (auto& s : stmts) r = r.unite(s.used_vars()); return r; }
};

// ------------------------------------------------------------------------------------------------
// Parser
// ------------------------------------------------------------------------------------------------
struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

class Parser {
    const std::string& src;
    size_t pos = 0;
    std::vector<std::string> vars;

    [[noreturn]] void fatal(const std::string& what) const {
        size_t line = 1;
        for (size_t i = 0; i < pos && i < src.size(); ++i) if (src[i] == '\n') line++;
        throw ParseError("Parse error:\n" + what + " at line " + std::to_string(line));
    }
    bool starts_with(const char* s) const { return src.compare(pos, std::strlen(s), s) == 0; }
    static bool is_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
    bool at_unicode_ws(size_t& len) const {  // Rust trim_start: ASCII ws + U+0085, U+00A0, U+1680, U+2000-200A, 2028, 2029, 202F, 205F, 3000
        unsigned char c = (unsigned char)src[pos];
        if (is_ws(c)) { len = 1; return true; }
        if (c == 0xC2 && pos + 1 < src.size()) { unsigned char d = (unsigned char)src[pos + 1]; if (d == 0x85 || d == 0xA0) { len = 2; return true; } }
        if (c == 0xE2 && pos + 2 < src.size()) {
            unsigned char d = (unsigned char)src[pos + 1], e = (unsigned char)src[pos + 2];
            if (d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) { len = 3; return true; }
            if (d == 0x81 && e == 0x9F) { len = 3; return true; }
        }
        if (c == 0xE1 && pos + 2 < src.size() && (unsigned char)src[pos + 1] == 0x9A && (unsigned char)src[pos + 2] == 0x80) { len = 3; return true; }
        if (c == 0xE3 && pos + 2 < src.size() && (unsigned char)src[pos + 1] == 0x80 && (unsigned char)src[pos + 2] == 0x80) { len = 3; return true; }
        return false;
    }
    void ws() {  // parser.rs:566-582
        for (;;) {
            size_t len;
            while (pos < src.size() && at_unicode_ws(len)) pos += len;
            if (starts_with("#=")) {
                size_t idx = src.find("=#", pos);
                if (idx == std::string::npos) throw ParseError("Unterminated comment: found opening `#=` but no closing `=#`");
                pos = idx + 2;
            } else if (pos < src.size() && src[pos] == '#') {
                while (pos < src.size() && src[pos] != '\n' && src[pos] != '\r') pos++;
            } else break;
        }
    }
    static bool is_alpha(unsigned char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
    static bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }
    static bool is_ident_rest(unsigned char c) { return is_alpha(c) || is_digit(c) || c == '_'; }
    bool ch(char c) { if (pos < src.size() && src[pos] == c) { pos++; return true; } return false; }
    bool tag(const char* s) { if (starts_with(s)) { pos += std::strlen(s); return true; } return false; }
    bool keyword(const char* k) {  // parser.rs:79-81
        size_t save = pos;
        if (tag(k) && !(pos < src.size() && is_ident_rest((unsigned char)src[pos]))) return true;
        pos = save;
        return false;
    }
    bool peek_keyword(const char* k) { size_t save = pos; bool ok = keyword(k); pos = save; return ok; }
    bool digits(std::string& out) {
        size_t s = pos;
        while (pos < src.size() && is_digit((unsigned char)src[pos])) pos++;
        if (pos == s) return false;
        out = src.substr(s, pos - s);
        return true;
    }
    template <class U> U parse_uint(const std::string& d, U maxv) {
        unsigned __int128 v = 0;
        for (char c : d) { v = v * 10 + (unsigned)(c - '0'); if (v > maxv) throw ParseError("number too large to fit in target type: " + d); }
        return (U)v;
    }
    bool natural(uint32_t& out) {  // parser.rs:18-22
        size_t save = pos;
        ws();
        std::string d;
        if (!digits(d)) { pos = save; return false; }
        ws();
        out = parse_uint<uint32_t>(d, UINT32_MAX);
        return true;
    }
    bool u64_natural(uint64_t& out) {
        size_t save = pos;
        ws();
        std::string d;
        if (!digits(d)) { pos = save; return false; }
        ws();
        out = parse_uint<uint64_t>(d, UINT64_MAX);
        return true;
    }
    bool natural_list(std::vector<uint32_t>& out) {  // parser.rs:30-39
        size_t save = pos;
        ws();
        if (!ch('[')) { pos = save; return false; }
        out.clear();
        uint32_t n;
        if (natural(n)) {
            out.push_back(n);
            for (;;) {
                size_t s2 = pos;
                if (!ch(',')) break;
                if (!natural(n)) { pos = s2; break; }
                out.push_back(n);
            }
        }
        if (!ch(']')) fatal("expected ']' in list of natural numbers");
        ws();
        return true;
    }
    bool pos_ratio(PosRatio& out) {  // parser.rs:41-68
        size_t save = pos;
        ws();
        size_t s1 = pos;
        uint64_t n, d;
        if (u64_natural(n)) {
            if (ch('/')) {
                if (!u64_natural(d)) fatal("expected denominator");
                out = PosRatio(n, d);
                ws();
                return true;
            }
        }
        pos = s1;
        std::string ip, fp;
        if (!digits(ip)) { pos = save; return false; }
        if (ch('.')) {
            if (!digits(fp)) fatal("expected digits after '.'");
            uint64_t den = 1;
            for (size_t i = 0; i < fp.size(); ++i) { if (den > UINT64_MAX / 10) throw ParseError("called `Option::unwrap()` on a `None` value (10^k overflow)"); den *= 10; }
            out = PosRatio(parse_uint<uint64_t>(ip + fp, UINT64_MAX), den);
        } else {
            out = PosRatio(parse_uint<uint64_t>(ip, UINT64_MAX), 1);
        }
        ws();
        return true;
    }
    bool identifier(std::string& out) {  // parser.rs:83-92
        size_t save = pos;
        ws();
        size_t s = pos;
        if (pos < src.size() && is_alpha((unsigned char)src[pos])) { while (pos < src.size() && is_alpha((unsigned char)src[pos])) pos++; }
        else if (pos < src.size() && src[pos] == '_') pos++;
        else { pos = save; return false; }
        while (pos < src.size() && is_ident_rest((unsigned char)src[pos])) pos++;
        out = src.substr(s, pos - s);
        ws();
        return true;
    }
    std::optional<size_t> find_var(const std::string& id) const {
        for (size_t i = 0; i < vars.size(); ++i) if (vars[i] == id) return i;
        return std::nullopt;
    }
    size_t find_or_create_var(const std::string& id) {
        if (auto v = find_var(id)) return *v;
        vars.push_back(id);
        return vars.size() - 1;
    }
    size_t expect_var(const std::string& id) const {
        if (auto v = find_var(id)) return *v;
        throw ParseError("Unknown variable " + id);
    }
    void semicolon() { ws(); if (!ch(';')) fatal("expected ';'"); }

    struct Operand { bool is_var; size_t var; uint32_t nat; };
    bool operand(Operand& o) {  // parser.rs:140-148
        uint32_t n;
        if (natural(n)) { o = Operand{false, 0, n}; return true; }
        std::string id;
        if (identifier(id)) { o = Operand{true, expect_var(id), 0}; return true; }
        return false;
    }
    static std::vector<uint32_t> upto(uint32_t n, bool incl) { std::vector<uint32_t> v; for (uint32_t i = 0; incl ? i <= n : i < n; ++i) { v.push_back(i); if (i == UINT32_MAX) break; } return v; }
    static Event::P event_eq(const Operand& l, const Operand& r) {  // parser.rs:150-159
        if (l.is_var && r.is_var) return Event::var_cmp(l.var, Event::Eq, r.var);
        if (l.is_var) return Event::in_set(l.var, {r.nat});
        if (r.is_var) return Event::in_set(r.var, {l.nat});
        return l.nat == r.nat ? Event::always() : Event::never();
    }
    static Event::P event_lt(const Operand& l, const Operand& r) {  // parser.rs:161-171
        if (l.is_var && r.is_var) return Event::var_cmp(l.var, Event::Lt, r.var);
        if (l.is_var) return Event::in_set(l.var, upto(r.nat, false));
        if (r.is_var) return Event::complement(Event::in_set(r.var, upto(l.nat, true)));
        return l.nat < r.nat ? Event::always() : Event::never();
    }
    static Event::P event_le(const Operand& l, const Operand& r) {  // parser.rs:173-185
        if (l.is_var && r.is_var) return Event::var_cmp(l.var, Event::Le, r.var);
        if (l.is_var) return Event::in_set(l.var, upto(r.nat, true));
        if (r.is_var) return Event::complement(Event::in_set(r.var, upto(l.nat, false)));
        return l.nat <= r.nat ? Event::always() : Event::never();
    }
    static Event::P event_in(const Operand& l, const std::vector<uint32_t>& ns) {  // parser.rs:187-193
        if (l.is_var) return Event::in_set(l.var, ns);
        for (uint32_t n : ns) if (n == l.nat) return Event::always();
        return Event::never();
    }
    Operand cut_operand() { Operand o; if (!operand(o)) fatal("expected comparee"); return o; }
    bool comparison(Event::P& out) {  // parser.rs:195-244
        size_t save = pos;
        Operand lhs;
        if (!operand(lhs)) { pos = save; return false; }
        std::vector<uint32_t> ns;
        if (ch('=')) { out = event_eq(lhs, cut_operand()); return true; }
        if (tag("<=") || tag("≤")) { out = event_le(lhs, cut_operand()); return true; }
        if (ch('<')) { out = event_lt(lhs, cut_operand()); return true; }
        if (keyword("in") || tag("∈")) { if (!natural_list(ns)) fatal("expected list of natural numbers"); out = event_in(lhs, ns); return true; }
        if (tag("!=") || tag("≠")) { out = Event::complement(event_eq(lhs, cut_operand())); return true; }
        if (tag(">=") || tag("≥")) { out = event_le(cut_operand(), lhs); return true; }
        if (ch('>')) { out = event_lt(cut_operand(), lhs); return true; }
        if (keyword("not in") || tag("∉")) { if (!natural_list(ns)) fatal("expected list of natural numbers"); out = Event::complement(event_in(lhs, ns)); return true; }
        pos = save;
        return false;
    }
    bool data_from_dist(Event::P& out) {  // parser.rs:246-251
        size_t save = pos;
        uint32_t data;
        if (!natural(data)) { pos = save; return false; }
        if (!ch('~')) { pos = save; return false; }
        Distribution d = distribution();
        out = Event::data_from(data, d);
        return true;
    }
    bool atomic_event(Event::P& out) {  // parser.rs:253-279
        size_t save = pos;
        if (tag("!") || keyword("not")) {
            Event::P e;
            if (!atomic_event(e)) fatal("expected simple event");
            out = Event::complement(e);
            return true;
        }
        pos = save;
        ws();
        if (ch('(')) {
            Event::P e = event_cut();
            ws();
            if (!ch(')')) fatal("expected ')'");
            out = e;
            return true;
        }
        pos = save;
        if (comparison(out)) return true;
        pos = save;
        if (data_from_dist(out)) return true;
        pos = save;
        return false;
    }
    Event::P event_cut() { Event::P e; if (!event(e)) fatal("expected event"); return e; }
    bool event(Event::P& out) {  // parser.rs:281-310
        Event::P e;
        if (!atomic_event(e)) return false;
        auto many1 = [&](const char* kw, const char* sym, std::vector<Event::P>& es) {
            for (;;) {
                size_t save = pos;
                ws();
                if (!(keyword(kw) || tag(sym))) { pos = save; break; }
                es.push_back(event_cut());
            }
            return !es.empty();
        };
        std::vector<Event::P> es;
        if (many1("and", "&&", es)) { es.insert(es.begin(), e); out = Event::intersection(es); return true; }
        if (many1("or", "||", es)) { es.insert(es.begin(), e); out = Event::disjunction(es); return true; }
        out = e;
        return true;
    }

    Distribution distribution() {  // parser.rs:379-495
        std::string name;
        if (!identifier(name)) fatal("expected distribution name");
        Distribution d;
        auto open = [&] { if (!ch('(')) fatal("expected '('"); };
        auto close = [&] { if (!ch(')')) fatal("expected ')'"); };
        auto cut_ratio = [&](PosRatio& r) { if (!pos_ratio(r)) fatal("expected real number"); };
        auto comma = [&] { if (!ch(',')) fatal("expected ','"); };
        std::string id;
        if (name == "Dirac") { open(); cut_ratio(d.p); close(); d.kind = Distribution::Dirac; }
        else if (name == "Bernoulli") {
            open();
            if (pos_ratio(d.p)) d.kind = Distribution::Bernoulli;
            else if (identifier(id)) { d.kind = Distribution::BernoulliVarProb; d.var = expect_var(id); }
            else fatal("expected probability or variable");
            close();
        } else if (name == "Binomial" || name == "NegBinomial") {
            open();
            bool neg = name == "NegBinomial";
            if (natural(d.n)) { comma(); if (!pos_ratio(d.p)) fatal("expected real number"); d.kind = neg ? Distribution::NegBinomial : Distribution::Binomial; }
            else if (identifier(id)) { comma(); if (!pos_ratio(d.p)) fatal("expected real number"); d.kind = neg ? Distribution::NegBinomialVarSuccesses : Distribution::BinomialVarTrials; d.var = expect_var(id); }
            else fatal("expected trials");
            close();
        } else if (name == "Categorical") {
            open();
            PosRatio r;
            if (!pos_ratio(r)) fatal("expected list of rational numbers");
            d.ps.push_back(r);
            for (;;) { size_t s = pos; if (!ch(',')) break; if (!pos_ratio(r)) { pos = s; break; } d.ps.push_back(r); }
            close();
            d.kind = Distribution::Categorical;
        } else if (name == "Geometric") { open(); cut_ratio(d.p); close(); d.kind = Distribution::Geometric; }
        else if (name == "Poisson") {
            open();
            if (pos_ratio(d.p)) {
                if (ch('*')) { if (!identifier(id)) fatal("expected identifier"); d.kind = Distribution::PoissonVarRate; d.var = expect_var(id); }
                else d.kind = Distribution::Poisson;
            } else if (identifier(id)) { d.p = PosRatio(1, 1); d.kind = Distribution::PoissonVarRate; d.var = expect_var(id); }
            else fatal("expected rate");
            close();
        } else if (name == "UniformDisc") {
            open();
            if (!natural(d.n)) fatal("expected natural number");
            comma();
            if (!natural(d.n2)) fatal("expected natural number");
            close();
            d.kind = Distribution::Uniform;
        } else if (name == "Exponential") { open(); cut_ratio(d.p); close(); d.kind = Distribution::Exponential; }
        else if (name == "Gamma" || name == "UniformCont") {
            open(); cut_ratio(d.p); comma(); cut_ratio(d.p2); close();
            d.kind = name == "Gamma" ? Distribution::Gamma : Distribution::UniformCont;
        } else throw ParseError("Unknown distribution " + name);
        return d;
    }

    std::vector<Statement> block() {  // parser.rs:584-596
        ws();
        if (!ch('{')) fatal("expected '{'");
        std::vector<Statement> out;
        std::vector<Statement> st;
        while (statement(st)) out.insert(out.end(), st.begin(), st.end());
        ws();
        if (!ch('}')) fatal("expected '}'");
        return out;
    }
    Statement if_event() {  // parser.rs:520-538
        if (!keyword("if")) fatal("expected if");
        Statement s;
        s.kind = Statement::IfThenElse;
        s.cond = event_cut();
        s.then = block();
        size_t save = pos;
        ws();
        if (keyword("else")) {
            ws();
            if (peek_keyword("if")) s.els.push_back(if_event());
            else s.els = block();
        } else pos = save;
        return s;
    }
    bool statement(std::vector<Statement>& out) {  // parser.rs:598-627
        size_t save = pos;
        out.clear();
        ws();
        if (peek_keyword("normalize")) {
            keyword("normalize");
            Statement s; s.kind = Statement::Normalize;
            std::string id;
            while (identifier(id)) s.given_vars.push_back(expect_var(id));
            s.then = block();
            out.push_back(s);
        } else if (peek_keyword("if")) {
            out.push_back(if_event());
        } else if (peek_keyword("observe")) {
            keyword("observe");
            Statement s; s.kind = Statement::IfThenElse;
            s.cond = event_cut();
            semicolon();
            Statement f; f.kind = Statement::Fail;
            s.els.push_back(f);
            out.push_back(s);
        } else if (peek_keyword("loop")) {
            keyword("loop");
            uint32_t count;
            if (!natural(count)) fatal("expected iteration count");
            std::vector<Statement> body = block();
            for (uint32_t i = 0; i < count; ++i) out.insert(out.end(), body.begin(), body.end());
        } else if (peek_keyword("while")) {
            keyword("while");
            Statement s; s.kind = Statement::While;
            s.cond = event_cut();
            size_t s2 = pos;
            ws();
            if (keyword("unroll")) { uint32_t u; if (!natural(u)) { pos = s2; } else { s.has_unroll = true; s.unroll = u; } }
            else pos = s2;
            s.then = block();
            out.push_back(s);
        } else if (peek_keyword("fail")) {
            keyword("fail");
            semicolon();
            Statement s; s.kind = Statement::Fail;
            out.push_back(s);
        } else {
            Statement s;
            if (!assign(s)) { pos = save; return false; }
            out.push_back(s);
        }
        ws();
        return true;
    }
    bool assign(Statement& s) {  // parser.rs:497-511
        std::string lhs;
        if (!identifier(lhs)) return false;
        if (starts_with("~") || starts_with("+~")) {
            s.kind = Statement::Sample;
            s.add_previous_value = !ch('~');
            if (s.add_previous_value) tag("+~");
            s.var = find_or_create_var(lhs);
            s.dist = distribution();
        } else if (starts_with("-=")) {
            tag("-=");
            s.kind = Statement::Decrement;
            if (!natural(s.offset)) return false;
            s.var = find_or_create_var(lhs);
        } else {
            if (tag(":=")) s.add_previous_value = false;
            else if (tag("+=")) s.add_previous_value = true;
            else return false;
            s.kind = Statement::Assign;
            // alt((factor * var (+ offset)?), natural)
            size_t save = pos;
            bool ok = false;
            {
                uint32_t factor = 1;
                size_t s1 = pos;
                uint32_t f;
                if (natural(f) && ch('*')) factor = f; else pos = s1;
                std::string w;
                if (identifier(w)) {
                    uint32_t off = 0;
                    if (ch('+')) { if (!natural(off)) fatal("expected natural number"); }
                    s.has_addend = true;
                    s.addend_factor = factor;
                    s.addend_var = expect_var(w);
                    s.offset = off;
                    ok = true;
                }
            }
            if (!ok) {
                pos = save;
                uint32_t n;
                if (!natural(n)) fatal("expected assignment right-hand side");
                s.has_addend = false;
                s.offset = n;
            }
            s.var = find_or_create_var(lhs);
        }
        semicolon();
        return true;
    }

  public:
    explicit Parser(const std::string& s) : src(s) {}
    Program program() {  // parser.rs:638-654
        Program p;
        std::vector<Statement> st;
        while (statement(st)) p.stmts.insert(p.stmts.end(), st.begin(), st.end());
        ws();
        if (!keyword("return")) fatal("expected return statement");
        std::string id;
        if (!identifier(id)) fatal("expected identifier");
        tag(";");
        ws();
        p.result = expect_var(id);
        ws();
        if (pos != src.size()) fatal("expected end of input");
        return p;
    }
};

inline Program parse_program(const std::string& src) { return Parser(src).program(); }

}  // namespace gfh
