// Host interpreter, part 6: program -> generating function (src/semantics/gf.rs).  Pure symbolic DAG
// construction; every formula below is the reference's PGF/MGF algebra line by line.
#pragma once
#include <iostream>

#include "gfh_genfun.hpp"

namespace gfh {

template <class T>
struct GfTranslation {  // gf.rs:12-19
    VarSupport var_info;
    GenFun<T> gf;
    GenFun<T> rest;
    VarSupport rest_info;
    static GfTranslation zero(size_t n) { return {VarSupport::empty(n), GenFun<T>::zero(), GenFun<T>::zero(), VarSupport::empty(n)}; }
    GfTranslation join(const GfTranslation& o) const {  // :37-44
        return {var_info.join(o.var_info), gf + o.gf, rest.max(o.rest), rest_info.join(o.rest_info)};
    }
    GfTranslation plus(const GfTranslation& o) const {  // :47-58
        return {var_info.join(o.var_info), gf + o.gf, rest + o.rest, rest_info.join(o.rest_info)};
    }
    void scale(const T& c) { gf = gf * GenFun<T>::constant(c); rest = rest * GenFun<T>::constant(c); }  // :61-65
};

template <class T>
GenFun<T> marginalize_out(size_t v, const GenFun<T>& gf, const VarSupport& vi) {  // gf.rs:637-649
    if (v >= vi.num_vars()) {
        if (v != vi.num_vars()) throw std::runtime_error("assertion failed: v.id() == var_info.num_vars()");
        return gf.substitute_var(v, GenFun<T>::one());
    }
    return gf.substitute_var(v, vi[v].is_discrete() ? GenFun<T>::one() : GenFun<T>::zero());
}
template <class T>
GenFun<T> marginalize_all(GenFun<T> gf, const VarSupport& vi) {  // gf.rs:651-657
    for (size_t v = 0; v < vi.num_vars(); ++v) gf = marginalize_out(v, gf, vi);
    return gf;
}

// Event::recognize_const_prob (ppl.rs:328-356)
template <class T>
bool recognize_const_prob(const Event& e, T& out) {
    switch (e.kind) {
        case Event::InSet: case Event::VarComparison: return false;
        case Event::DataFromDist:
            if (e.dist.kind == Distribution::Bernoulli) {
                if (e.data == 0) { PosRatio c = e.dist.p.complement(); out = T::from_ratio(c.numer, c.denom); }
                else if (e.data == 1) out = T::from_ratio(e.dist.p.numer, e.dist.p.denom);
                else out = T::zero();
                return true;
            }
            return false;
        case Event::Complement: { T p; if (!recognize_const_prob(*e.sub[0], p)) return false; out = T::one() - p; return true; }
        case Event::Intersection: {
            T r = T::one();
            for (auto& s : e.sub) { T p; if (!recognize_const_prob(*s, p)) return false; r = r * p; }
            out = r;
            return true;
        }
    }
    return false;
}

template <class T>
struct GfTransformer {
    typedef GenFun<T> G;
    typedef GfTranslation<T> Tr;
    size_t unroll = 0;
    SupportTransformer support;
    std::ostream* out = &std::cout;  // where the reference's println! warnings go

    void with_unroll(size_t u) { unroll = u; support.unroll = u; }

    Tr init(const Program& p) {
        VarSupport vi = support.init(p);
        return {vi, G::one(), G::zero(), VarSupport::empty(vi.num_vars())};
    }
    Tr semantics(const Program& p) { return transform_statements(p.stmts, init(p)); }
    Tr transform_statements(const std::vector<Statement>& stmts, Tr cur) {
        for (auto& s : stmts) cur = transform_statement(s, std::move(cur));
        return cur;
    }

    static G gf_in_set(size_t var, const std::vector<uint32_t>& set, const G& gf) {  // gf.rs:103-110
        if (set.size() == 1) return gf.taylor_coeff_at_zero(var, set[0]) * G::var(var).pow(set[0]);
        Dims orders(set.begin(), set.end());
        return gf.taylor_polynomial_at_zero(var, orders);
    }

    std::pair<Tr, Tr> transform_event(const Event& event, const Tr& init) {  // gf.rs:98-205
        VarSupport var_info = init.var_info, rest_info = init.rest_info;
        G rest = init.rest, gf0 = init.gf, gf;
        switch (event.kind) {
            case Event::InSet: gf = gf_in_set(event.var, event.set, gf0); break;
            case Event::VarComparison: {
                uint32_t lo1, hi1, lo2, hi2;
                bool f1 = var_info[event.var].finite_nonempty_range(lo1, hi1), f2 = var_info[event.var2].finite_nonempty_range(lo2, hi2);
                size_t scrutinee, other;
                bool reversed;
                uint32_t lo, hi;
                if (!f1 && !f2) throw std::runtime_error("Cannot compare two variables with infinite support.");
                if (!f1) { scrutinee = event.var2; other = event.var; reversed = false; lo = lo2; hi = hi2; }
                else if (!f2) { scrutinee = event.var; other = event.var2; reversed = true; lo = lo1; hi = hi1; }
                else if (hi1 - lo1 <= hi2 - lo2) { scrutinee = event.var; other = event.var2; reversed = true; lo = lo1; hi = hi1; }
                else { scrutinee = event.var2; other = event.var; reversed = false; lo = lo2; hi = hi2; }
                G result = G::zero();
                auto upto = [](uint32_t n, bool incl) { std::vector<uint32_t> v; for (uint32_t i = 0; incl ? i <= n : i < n; ++i) v.push_back(i); return v; };
                for (uint32_t i = lo;; ++i) {
                    G eq_i = gf_in_set(scrutinee, {i}, gf0);
                    G summand;
                    if (event.cmp == Event::Eq) summand = gf_in_set(other, {i}, eq_i);
                    else if (event.cmp == Event::Lt && !reversed) summand = gf_in_set(other, upto(i, false), eq_i);
                    else if (event.cmp == Event::Lt && reversed) summand = eq_i - gf_in_set(other, upto(i, true), eq_i);
                    else if (event.cmp == Event::Le && !reversed) summand = gf_in_set(other, upto(i, true), eq_i);
                    else summand = eq_i - gf_in_set(other, upto(i, false), eq_i);
                    result = result + summand;
                    if (i == hi) break;
                }
                gf = result;
                break;
            }
            case Event::DataFromDist: {
                T factor;
                if (recognize_const_prob(event, factor)) gf = G::constant(factor) * gf0;
                else gf = transform_data_from_dist(event.data, event.dist, var_info, gf0);
                break;
            }
            case Event::Complement: gf = transform_event(*event.sub[0], init).second.gf; break;
            case Event::Intersection: {
                Tr then = init;
                for (auto& e : event.sub) then = transform_event(*e, then).first;
                gf = then.gf;
                break;
            }
        }
        auto vs = support.transform_event(event, var_info);
        auto rs = support.transform_event(event, rest_info);
        return {Tr{vs.first, gf, rest, rs.first}, Tr{vs.second, init.gf - gf, rest, rs.second}};
    }

    Tr transform_statement(const Statement& st, Tr init) {  // gf.rs:208-356
        switch (st.kind) {
            case Statement::Sample: return transform_distribution(st.dist, st.var, init, st.add_previous_value);
            case Statement::Assign: {
                size_t v = st.var;
                G gf = init.gf;
                VarSupport var_info = init.var_info;
                G var = G::var(v);
                uint32_t v_exp = st.add_previous_value ? 1 : 0;
                bool has_w = false;
                size_t w = 0;
                G w_subst;
                if (st.has_addend) {
                    if (v == st.addend_var) v_exp += st.addend_factor;
                    else if (var_info[st.addend_var].is_discrete()) { has_w = true; w = st.addend_var; w_subst = G::var(w) * var.pow(st.addend_factor); }
                    else {
                        if (!(!var_info[v].is_discrete() || !st.add_previous_value)) throw std::runtime_error("cannot add a continuous to a discrete variable");
                        has_w = true; w = st.addend_var; w_subst = G::var(w) + var * G::from_u32(st.addend_factor);
                    }
                }
                if (var_info[v].is_discrete()) gf = gf.substitute_var(v, var.pow(v_exp));
                else gf = gf.substitute_var(v, var * G::from_u32(v_exp));
                if (has_w) gf = gf.substitute_var(w, w_subst);
                VarSupport nvi = support.transform_statement(st, var_info);
                VarSupport nri = support.transform_statement(st, init.rest_info);
                if (nvi[v].is_discrete()) gf = gf * var.pow(st.offset);
                else gf = gf * (var * G::from_u32(st.offset)).exp();
                return {nvi, gf, init.rest, nri};
            }
            case Statement::Decrement: {
                if (!init.var_info[st.var].is_discrete()) throw std::runtime_error("cannot decrement continuous variables");
                VarSupport nvi = support.transform_statement(st, init.var_info);
                VarSupport nri = support.transform_statement(st, init.rest_info);
                return {nvi, init.gf.shift_down_taylor_at_zero(st.var, st.offset), init.rest, nri};
            }
            case Statement::IfThenElse: {
                T factor;
                if (recognize_const_prob(*st.cond, factor)) {
                    Tr t = transform_statements(st.then, init);
                    Tr e = transform_statements(st.els, init);
                    t.scale(factor);
                    e.scale(T::one() - factor);
                    return t.plus(e);
                }
                auto be = transform_event(*st.cond, init);
                Tr t = transform_statements(st.then, be.first);
                Tr e = transform_statements(st.els, be.second);
                return t.join(e);
            }
            case Statement::While: {
                std::cerr << "WARNING: support for while loops is EXPERIMENTAL" << std::endl;
                *out << "WARNING: results are APPROXIMATE due to presence of loops: exact inference is only possible for loop-free programs\n";
                Tr result = Tr::zero(init.var_info.num_vars());
                Tr rest = init;
                size_t count = st.has_unroll ? st.unroll : unroll;
                for (size_t i = 0; i < count; ++i) {
                    auto ee = transform_event(*st.cond, rest);
                    result = result.join(ee.second);
                    rest = transform_statements(st.then, ee.first);
                }
                result.rest = result.rest + marginalize_all(rest.gf, rest.var_info);
                VarSupport inv = support.find_while_invariant(*st.cond, st.then, rest.var_info);
                auto ex = support.transform_event(*st.cond, inv);
                result.rest_info = result.rest_info.join(ex.second);
                result.var_info = result.var_info.join(result.rest_info);
                return result;
            }
            case Statement::Fail: return Tr::zero(init.var_info.num_vars());
            case Statement::Normalize: return transform_normalize(st.given_vars, 0, st.then, init);
        }
        return init;
    }

    static G compound_dist(const G& gf, const G& base, size_t sampled, size_t param, bool add_prev, bool param_discrete, const G& subst) {  // gf.rs:366-392
        if (sampled == param) {
            if (add_prev) return gf.substitute_var(param, param_discrete ? G::var(param) * subst : G::var(param) + subst);
            return gf.substitute_var(param, subst);
        }
        return base.substitute_var(param, param_discrete ? G::var(param) * subst : G::var(param) + subst);
    }

    static Tr transform_distribution(const Distribution& d, size_t v, const Tr& tr, bool add_prev) {  // gf.rs:395-541
        G base = add_prev ? tr.gf : marginalize_out(v, tr.gf, tr.var_info);
        VarSupport nvi = SupportTransformer::transform_distribution(d, v, tr.var_info, add_prev);
        VarSupport nri = SupportTransformer::transform_distribution(d, v, tr.rest_info, add_prev);
        const G& gf0 = tr.gf;
        G gf;
        auto ratio = [](const PosRatio& r) { return G::from_ratio(r); };
        switch (d.kind) {
            case Distribution::Dirac: {
                uint32_t a;
                G dirac = d.p.as_integer(a) ? G::var(v).pow(a) : (G::var(v) * ratio(d.p)).exp();
                gf = dirac * base;
                break;
            }
            case Distribution::Bernoulli: gf = (ratio(d.p) * G::var(v) + ratio(d.p.complement())) * base; break;
            case Distribution::BernoulliVarProb: {
                size_t w = d.var;
                G ptg = tr.var_info[w].is_discrete() ? gf0.derive(w, 1) * G::var(w) : gf0.derive(w, 1);
                G ptb = add_prev ? ptg : marginalize_out(v, ptg, tr.var_info);
                G v_term = nvi[v].is_discrete() ? G::var(v) : G::var(v).exp();
                gf = base + (v_term - G::one()) * ptb;
                break;
            }
            case Distribution::BinomialVarTrials: {
                G subst = ratio(d.p) * G::var(v) + ratio(d.p.complement());
                gf = compound_dist(gf0, base, v, d.var, add_prev, true, subst);
                break;
            }
            case Distribution::Binomial: gf = (ratio(d.p) * G::var(v) + ratio(d.p.complement())).pow(d.n) * base; break;
            case Distribution::Categorical: {
                G cat = G::zero();
                for (auto it = d.ps.rbegin(); it != d.ps.rend(); ++it) { cat = cat * G::var(v); cat = cat + ratio(*it); }
                gf = cat * base;
                break;
            }
            case Distribution::NegBinomialVarSuccesses: {
                G subst = ratio(d.p) / (G::one() - ratio(d.p.complement()) * G::var(v));
                gf = compound_dist(gf0, base, v, d.var, add_prev, true, subst);
                break;
            }
            case Distribution::NegBinomial: {
                G geo = ratio(d.p) / (G::one() - ratio(d.p.complement()) * G::var(v));
                gf = geo.pow(d.n) * base;
                break;
            }
            case Distribution::Geometric: gf = (ratio(d.p) / (G::one() - ratio(d.p.complement()) * G::var(v))) * base; break;
            case Distribution::Poisson: gf = (ratio(d.p) * (G::var(v) - G::one())).exp() * base; break;
            case Distribution::PoissonVarRate: {
                bool wd = tr.var_info[d.var].is_discrete();
                G subst = wd ? (ratio(d.p) * (G::var(v) - G::one())).exp() : ratio(d.p) * (G::var(v) - G::one());
                gf = compound_dist(gf0, base, v, d.var, add_prev, wd, subst);
                break;
            }
            case Distribution::Uniform: {
                if (!(d.n2 > d.n)) throw std::runtime_error("Uniform distribution cannot have length 0");
                uint32_t length = d.n2 - d.n;
                G weight = ratio(PosRatio(1, length));
                G uni = G::zero();
                for (uint32_t i = 0; i < length; ++i) uni = weight + G::var(v) * uni;
                uni = uni * G::var(v).pow(d.n);
                gf = uni * base;
                break;
            }
            case Distribution::Exponential: { G beta = ratio(d.p); gf = (beta / (beta - G::var(v))) * base; break; }
            case Distribution::Gamma: {
                G beta = ratio(d.p2);
                uint32_t shape;
                G gamma = d.p.as_integer(shape) ? (beta / (beta - G::var(v))).pow(shape)
                                                : (ratio(d.p) * (beta.log() - (beta - G::var(v)).log())).exp();
                gf = gamma * base;
                break;
            }
            case Distribution::UniformCont: {
                T width = T::from_ratio(d.p2.numer, d.p2.denom) - T::from_ratio(d.p.numer, d.p.denom);
                G x = G::constant(width) * G::var(v);
                gf = (G::uniform_mgf(x) * (ratio(d.p) * G::var(v)).exp()) * base;
                break;
            }
        }
        return {nvi, gf, tr.rest, nri};
    }

    G transform_data_from_dist(uint32_t data, const Distribution& d, const VarSupport& vi, const G& gf) {  // gf.rs:543-592
        if (d.kind == Distribution::BernoulliVarProb) {
            G ptg = vi[d.var].is_discrete() ? gf.derive(d.var, 1) * G::var(d.var) : gf.derive(d.var, 1);
            if (data == 0) return gf - ptg;
            if (data == 1) return ptg;
            return G::zero();
        }
        if (d.kind == Distribution::BinomialVarTrials) {
            G repl = G::from_ratio(d.p.complement()) * G::var(d.var);
            return gf.taylor_coeff(d.var, data).substitute_var(d.var, repl) * (G::from_ratio(d.p) * G::var(d.var)).pow(data);
        }
        size_t new_var = gf.used_vars().num_vars();
        Statement s;
        s.kind = Statement::Sample;
        s.var = new_var;
        s.dist = d;
        s.add_previous_value = false;
        Tr tr{vi, gf, G::zero(), VarSupport::empty(vi.num_vars())};
        Tr nt = transform_statement(s, tr);
        G g2 = nt.gf.taylor_coeff_at_zero(new_var, data);
        return marginalize_out(new_var, g2, nt.var_info);
    }

    Tr transform_normalize(const std::vector<size_t>& given, size_t idx, const std::vector<Statement>& block, const Tr& tr) {  // gf.rs:594-640
        if (idx == given.size()) {
            G total_before = marginalize_all(tr.gf, tr.var_info);
            G rest_before = tr.rest;
            Tr t = transform_statements(block, tr);
            G total_after = marginalize_all(t.gf, t.var_info);
            G rest_after = t.rest;
            G min_factor = total_before / (total_after + rest_after);
            G max_factor = (total_before + rest_before) / total_after;
            return {t.var_info, min_factor * t.gf, max_factor * t.rest, t.rest_info};
        }
        size_t v = given[idx];
        uint32_t lo, hi;
        if (!tr.var_info[v].finite_nonempty_range(lo, hi))
            throw std::runtime_error("Cannot normalize with respect to variable `" + var_name(v) + "`, because its value could not be proven to be bounded.");
        Tr joined = Tr::zero(tr.var_info.num_vars());
        for (uint32_t i = lo;; ++i) {
            G summand = tr.gf.taylor_coeff_at_zero(v, i) * G::var(v).pow(i);
            VarSupport vi = tr.var_info, ri = tr.rest_info;
            vi.set(v, SupportSet::point(i));
            ri.set(v, SupportSet::point(i));
            Tr s{vi, summand, tr.rest, ri};
            joined = joined.join(transform_normalize(given, idx + 1, block, s));
            if (i == hi) break;
        }
        return joined;
    }
};

}  // namespace gfh
