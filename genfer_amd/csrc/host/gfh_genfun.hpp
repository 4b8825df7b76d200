// Host interpreter, part 5: the generating-function DAG and its evaluator (src/generating_function.rs).
// `GenFun<T>` = shared immutable nodes (Rc in the reference); `simplify` folds polynomial sub-DAGs into a
// single Polynomial node; `eval` interprets the DAG, turning every arithmetic node into one TaylorPoly
// operation on the backend (gf.rs:548-668) — this is the sole caller of the hot path.
#pragma once
#include <algorithm>
#include <map>
#include <memory>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "gfh_backend.hpp"
#include "gfh_support.hpp"

namespace gfh {

template <class T>
struct GenFun {
    enum Kind { Var, Const, Add, Neg, Mul, Div, Polynomial, Exp, Log, Pow, UniformMgf, Subst, Derivative,
                TaylorPolynomial, TaylorCoeffAtZero, TaylorCoeff, ShiftTaylorAtZero, Max };
    struct Node {
        Kind kind;
        size_t var = 0;
        T c;
        uint32_t n = 0;          // Pow exponent
        size_t order = 0;        // Derivative / TaylorCoeff(AtZero) / Shift order
        Dims orders;             // TaylorPolynomial
        GenFun a, b;             // children (b: second operand / substitution)
        std::vector<T> coeffs;   // Polynomial
        Dims shape;
        // used_vars() of this (immutable) node, memoised: the translation of `observe` asks for it on the whole growing
        // DAG at every statement (semantics/gf.rs), which the reference answers with a fresh traversal each time
        mutable bool uv_known = false;
        mutable VarRange uv;
        // Const nodes: the backend handle of `c` (handles are immutable values), and the backend table it belongs to
        mutable Poly<T> const_tp;
        mutable const Api* const_api = nullptr;
        mutable bool const_tp_set = false;
    };
    Rc<const Node> p;

    static GenFun mk(Node n) { GenFun g; g.p = rc_allocate<const Node>(gft_small::Alloc<Node>(), std::move(n)); return g; }  // (eval builds nodes too: gf.rs:684-706)
    // the node kinds eval itself builds by the thousand (observation chains): filled in where they live, no Node moved twice
    static GenFun mk_in_place(Kind k, const GenFun* a, const GenFun* b, size_t var_, size_t order_) {
        auto sp = rc_allocate<Node>(gft_small::Alloc<Node>());
        sp->kind = k;
        if (a) sp->a = *a;
        if (b) sp->b = *b;
        sp->var = var_;
        sp->order = order_;
        GenFun g;
        g.p = std::move(sp);
        return g;
    }
    static GenFun var(size_t v) { return mk_in_place(Var, nullptr, nullptr, v, 0); }
    static GenFun constant(const T& x) { Node n; n.kind = Const; n.c = x; return mk(std::move(n)); }
    static GenFun zero() { return constant(T::zero()); }
    static GenFun one() { return constant(T::one()); }
    static GenFun from_u32(uint32_t u) { return constant(T::from_u32(u)); }
    static GenFun from_ratio(const PosRatio& r) { return constant(T::from_ratio(r.numer, r.denom)); }
    static GenFun polynomial(std::vector<T> coeffs, Dims shape) { Node n; n.kind = Polynomial; n.coeffs = std::move(coeffs); n.shape = std::move(shape); return mk(std::move(n)); }
    static GenFun un(Kind k, const GenFun& a) { return mk_in_place(k, &a, nullptr, 0, 0); }
    static GenFun bin(Kind k, const GenFun& a, const GenFun& b) { return mk_in_place(k, &a, &b, 0, 0); }
    GenFun exp() const { return un(Exp, *this); }
    GenFun log() const { return un(Log, *this); }
    GenFun pow(uint32_t e) const { Node n; n.kind = Pow; n.a = *this; n.n = e; return mk(std::move(n)); }
    GenFun max(const GenFun& g) const { return bin(Max, *this, g); }
    static GenFun uniform_mgf(const GenFun& g) { return un(UniformMgf, g); }
    GenFun derive(size_t v, size_t order) const { return mk_in_place(Derivative, this, nullptr, v, order); }
    GenFun taylor_polynomial_at_zero(size_t v, Dims orders) const { Node n; n.kind = TaylorPolynomial; n.a = *this; n.var = v; n.orders = std::move(orders); return mk(std::move(n)); }
    GenFun taylor_coeff_at_zero(size_t v, size_t order) const { Node n; n.kind = TaylorCoeffAtZero; n.a = *this; n.var = v; n.order = order; return mk(std::move(n)); }
    GenFun taylor_coeff(size_t v, size_t order) const { Node n; n.kind = TaylorCoeff; n.a = *this; n.var = v; n.order = order; return mk(std::move(n)); }
    GenFun shift_down_taylor_at_zero(size_t v, size_t order) const { Node n; n.kind = ShiftTaylorAtZero; n.a = *this; n.var = v; n.order = order; return mk(std::move(n)); }
    GenFun substitute_var(size_t v, const GenFun& val) const { Node n; n.kind = Subst; n.a = *this; n.var = v; n.b = val; return mk(std::move(n)); }
    friend GenFun operator+(const GenFun& a, const GenFun& b) { return bin(Add, a, b); }
    friend GenFun operator-(const GenFun& a) { return un(Neg, a); }
    friend GenFun operator-(const GenFun& a, const GenFun& b) { return a + (-b); }
    friend GenFun operator*(const GenFun& a, const GenFun& b) { return bin(Mul, a, b); }
    friend GenFun operator/(const GenFun& a, const GenFun& b) { return bin(Div, a, b); }

    // ---- Display (generating_function.rs:330-432, precedence :451-470; `--print-gf`) ---------------------------------
    static std::string var_name(size_t i) {  // ppl.rs:107-117
        if (i < 26) return std::string(1, (char)('a' + i));
        return "x_" + std::to_string(i);
    }
    // fmt_polynomial (multivariate_taylor.rs:694-724): non-zero coefficients in row-major order, "c" then every
    // variable with a non-zero exponent ("^e" above 1), joined by " + "; "0" for the zero polynomial
    static std::string fmt_polynomial(const std::vector<T>& coeffs, const Dims& shape) {
        std::string out;
        bool first = true;
        Dims idx(shape.size(), 0);
        for (size_t lin = 0; lin < coeffs.size(); ++lin) {
            if (!coeffs[lin].is_zero()) {
                if (!first) out += " + ";
                first = false;
                out += coeffs[lin].str();
                for (size_t i = 0; i < idx.size(); ++i) {
                    if (idx[i] == 0) continue;
                    out += var_name(i);
                    if (idx[i] > 1) out += "^" + std::to_string(idx[i]);
                }
            }
            for (size_t ax = idx.size(); ax-- > 0;) {
                if (++idx[ax] < shape[ax]) break;
                idx[ax] = 0;
            }
        }
        return first ? "0" : out;
    }
    static int precedence(Kind k) {
        switch (k) {
            case Add: case Neg: case Polynomial: return 0;
            case Mul: case Div: return 1;
            case Pow: return 2;
            default: return 10;
        }
    }
    std::string str(int parent_prec = 0) const {
        const Node& x = *p;
        const int cur = precedence(x.kind);
        std::string o;
        if (cur < parent_prec) o += "(";
        auto list = [](const Dims& v) {  // {:?} of a Vec<usize>
            std::string r = "[";
            for (size_t i = 0; i < v.size(); ++i) r += (i ? ", " : "") + std::to_string(v[i]);
            return r + "]";
        };
        switch (x.kind) {
            case Var: o += var_name(x.var); break;
            case Const: o += x.c.str(); break;
            case Add: o += x.a.str(cur) + " + " + x.b.str(cur); break;
            case Neg: o += "-" + x.a.str(cur + 1); break;
            case Mul: o += x.a.str(cur) + " * " + x.b.str(cur); break;
            case Div: o += x.a.str(cur) + " / " + x.b.str(cur + 1); break;
            case Polynomial: o += fmt_polynomial(x.coeffs, x.shape); break;
            case Exp: o += "exp(" + x.a.str(0) + ")"; break;
            case Log: o += "log(" + x.a.str(0) + ")"; break;
            case Pow: o += x.a.str(cur + 1) + "^" + std::to_string(x.n); break;
            case Max: o += "max(" + x.a.str(0) + ", " + x.b.str(0) + ")"; break;
            case UniformMgf: o += "uniform_mgf(" + x.a.str(0) + ")"; break;
            case Subst: o += "[" + var_name(x.var) + " -> " + x.b.str(0) + " in " + x.a.str(0) + "]"; break;
            case Derivative: o += "d_" + var_name(x.var) + "^" + std::to_string(x.order) + "(" + x.a.str(0) + ")"; break;
            case TaylorPolynomial: o += "taylor(" + x.a.str(0) + " of " + var_name(x.var) + "^i with i ∈ " + list(x.orders) + ")"; break;
            case TaylorCoeffAtZero: o += "coeff_at_zero(" + x.a.str(0) + " of " + var_name(x.var) + "^" + std::to_string(x.order) + ")"; break;
            case TaylorCoeff: o += "coeff(" + x.a.str(0) + " of " + var_name(x.var) + "^" + std::to_string(x.order) + ")"; break;
            case ShiftTaylorAtZero: o += "shift(" + x.a.str(0) + " of " + var_name(x.var) + " by " + std::to_string(x.order) + ")"; break;
        }
        if (cur < parent_prec) o += ")";
        return o;
    }

    // derived PartialEq (structural)
    bool operator==(const GenFun& o) const {
        if (p == o.p) return true;
        if (!p || !o.p) return false;
        const Node &x = *p, &y = *o.p;
        if (x.kind != y.kind) return false;
        switch (x.kind) {
            case Var: return x.var == y.var;
            case Const: return x.c == y.c;
            case Add: case Mul: case Div: case Max: return x.a == y.a && x.b == y.b;
            case Neg: case Exp: case Log: case UniformMgf: return x.a == y.a;
            case Pow: return x.a == y.a && x.n == y.n;
            case Polynomial: return x.shape == y.shape && x.coeffs == y.coeffs;
            case Subst: return x.a == y.a && x.var == y.var && x.b == y.b;
            case Derivative: case TaylorCoeffAtZero: case TaylorCoeff: case ShiftTaylorAtZero: return x.a == y.a && x.var == y.var && x.order == y.order;
            case TaylorPolynomial: return x.a == y.a && x.var == y.var && x.orders == y.orders;
        }
        return false;
    }
    bool operator!=(const GenFun& o) const { return !(*this == o); }

    // gf.rs:422-443
    VarRange used_vars() const {
        std::unordered_map<const Node*, VarRange> cache;
        return used_vars_with(cache);
    }
    VarRange used_vars_with(std::unordered_map<const Node*, VarRange>& cache) const {
        if (p->uv_known) return p->uv;
        const Node& x = *p;
        VarRange r;
        switch (x.kind) {
            case Var: r = VarRange::of(x.var); break;
            case Const: break;
            case Add: case Mul: case Div: case Max: r = x.a.used_vars_with(cache).unite(x.b.used_vars_with(cache)); break;
            case Neg: case Exp: case Log: case Pow: case UniformMgf: r = x.a.used_vars_with(cache); break;
            case Polynomial: r = VarRange{x.shape.size()}; break;
            case Subst: r = x.a.used_vars_with(cache).remove(x.var).unite(x.b.used_vars_with(cache)); break;
            case TaylorCoeffAtZero: r = x.a.used_vars_with(cache).remove(x.var); break;
            case Derivative: case TaylorPolynomial: case TaylorCoeff: case ShiftTaylorAtZero: r = x.a.used_vars_with(cache); break;
        }
        x.uv_known = true;
        x.uv = r;
        return r;
    }

    // ---- simplify (gf.rs:152-158, 474-545) ----------------------------------------------------------
    typedef Poly<T> TP;
    struct MaybeTP { bool some = false; TP v; };
    typedef std::unordered_map<const Node*, MaybeTP, std::hash<const Node*>, std::equal_to<const Node*>, gft_small::Alloc<std::pair<const Node* const, MaybeTP>>> SimplifyCache;
    GenFun simplify() const {
        SimplifyCache cache;
        cache.reserve(1u << 18);  // (switchpoint: 2e5 nodes — growing the table by rehashing was a tenth of its run time)
        MaybeTP r = simplify_with(cache);
        if (!r.some) return *this;
        Dims shape;
        std::vector<T> data = r.v.to_vector(&shape);
        return polynomial(std::move(data), shape);
    }
    MaybeTP simplify_with(SimplifyCache& cache) const {
        auto it = cache.find(p.get());
        if (it != cache.end()) return it->second;
        MaybeTP r = simplify_node(cache);
        cache[p.get()] = r;
        return r;
    }
    MaybeTP simplify_node(SimplifyCache& cache) const {
        const Node& x = *p;
        auto some = [](TP v) { MaybeTP m; m.some = true; m.v = v; return m; };
        MaybeTP none;
        switch (x.kind) {
            case Var: return some(TP::var_with_degrees_p1(x.var, T::zero(), Dims(x.var + 1, UMAX)));
            case Const: return some(TP::from(x.c));
            case Add: { auto g = x.a.simplify_with(cache), h = x.b.simplify_with(cache); return (g.some && h.some) ? some(g.v + h.v) : none; }
            case Neg: { auto g = x.a.simplify_with(cache); return g.some ? some(-g.v) : none; }
            case Mul: { auto g = x.a.simplify_with(cache), h = x.b.simplify_with(cache); return (g.some && h.some) ? some(g.v * h.v) : none; }
            case Div: {
                auto g = x.a.simplify_with(cache), h = x.b.simplify_with(cache);
                T c;
                if (g.some && h.some && h.v.extract_constant(c)) return some(g.v / h.v);
                return none;
            }
            case Polynomial: case Exp: case Log: case Max: case UniformMgf: return none;
            case Pow: { auto g = x.a.simplify_with(cache); return g.some ? some(g.v.pow(x.n)) : none; }
            case Subst: { auto g = x.a.simplify_with(cache), s = x.b.simplify_with(cache); return (g.some && s.some) ? some(g.v.subst_var(x.var, s.v)) : none; }
            case Derivative: { auto g = x.a.simplify_with(cache); return g.some ? some(g.v.derivative(x.var, x.order)) : none; }
            case TaylorPolynomial: { auto g = x.a.simplify_with(cache); return g.some ? some(g.v.taylor_polynomial_terms(x.var, x.orders)) : none; }
            case TaylorCoeffAtZero: {
                auto g = x.a.simplify_with(cache);
                if (!g.some) return none;
                TP res = g.v.coefficients_of_term(x.var, x.order);
                if (x.var + 1 == res.num_vars()) res = res.remove_last_variable();
                return some(res);
            }
            case TaylorCoeff: { auto g = x.a.simplify_with(cache); return g.some ? some(g.v.taylor_expansion_of_coeff(x.var, x.order)) : none; }
            case ShiftTaylorAtZero: { auto g = x.a.simplify_with(cache); return g.some ? some(g.v.shift_down(x.var, x.order)) : none; }
        }
        return none;
    }

    // ---- eval (gf.rs:180-222, 548-765) ------------------------------------------------------------------
    // `gf` keeps the node alive so its address cannot be recycled for another node while it is a cache key
    // (the reference stores `gf: self.clone()` for the same reason, gf.rs:212-219)
    // (the evaluator copies its input points at every Subst node and into every cache entry: small vectors from the small-block lists)
    typedef std::vector<T, gft_small::Alloc<T>> Inputs;
    struct EvalResult { GenFun gf; Inputs inputs; size_t degree_p1; TP output; };
    typedef std::unordered_map<const Node*, EvalResult, std::hash<const Node*>, std::equal_to<const Node*>, gft_small::Alloc<std::pair<const Node* const, EvalResult>>> EvalCache;

    TP eval(const std::vector<T>& inputs, size_t degree_p1) const {
        EvalCache cache;
        cache.reserve(1u << 18);
        struct ChainScope {  // the chains live as long as this evaluation
            ChainScope() { chain_table().clear(); }
            ~ChainScope() { chain_table().clear(); }
        } chains;
        return eval_with(Inputs(inputs.begin(), inputs.end()), degree_p1, cache);
    }
    TP eval_with(const Inputs& inputs, size_t degree_p1, EvalCache& cache) const {
        const bool shared = p.use_count() > 1;
        if (shared) {
            auto it = cache.find(p.get());
            if (it != cache.end() && it->second.inputs == inputs && it->second.degree_p1 == degree_p1) return it->second.output;
        }
        TP result = eval_node(inputs, degree_p1, cache);
        if (shared) cache[p.get()] = EvalResult{*this, inputs, degree_p1, result};
        return result;
    }
    TP eval_node(const Inputs& inputs, size_t degree_p1, EvalCache& cache) const {
        const Node& x = *p;
        switch (x.kind) {
            case Var: return TP::var(x.var, inputs.at(x.var), degree_p1);
            case Const: {  // (the handle of a constant is formed once per node and backend: programs evaluate their Const nodes 10^5 times)
                if (!x.const_tp_set || x.const_api != &TP::api()) {
                    x.const_tp = TP::from(x.c);
                    x.const_api = &TP::api();
                    x.const_tp_set = true;
                }
                return x.const_tp;
            }
            case Add: { TP g = x.a.eval_with(inputs, degree_p1, cache); TP h = x.b.eval_with(inputs, degree_p1, cache); return g + h; }
            case Neg: return -x.a.eval_with(inputs, degree_p1, cache);
            case Mul: {
                // (d/dv G) * v * const — one level of the compound-Poisson observation chain built by
                // eval_taylor_coeff_at_zero (gf.rs:684-689): evaluated by the backend's fused observe_step, which
                // performs exactly the three reference operations derivative/truncate, * var, * const.
                {
                    // the whole chain at once: walk down while the operand has the same shape for the same variable
                    auto step_of = [](const Node& n, size_t* v, const GenFun** inner) {
                        if (n.kind != Mul || n.b.p->kind != Const || n.a.p->kind != Mul) return false;
                        const Node& m = *n.a.p;
                        if (!(m.a.p->kind == Derivative && m.a.p->order == 1 && m.b.p->kind == Var && m.b.p->var == m.a.p->var)) return false;
                        *v = m.b.p->var;
                        *inner = &m.a.p->a;
                        return true;
                    };
                    size_t v = 0;
                    const GenFun* inner = nullptr;
                    if (step_of(x, &v, &inner)) {
                        Inputs cs{x.b.p->c};  // outermost first
                        size_t v2 = 0;
                        const GenFun* in2 = nullptr;
                        while (step_of(*inner->p, &v2, &in2) && v2 == v) {
                            cs.push_back(inner->p->b.p->c);
                            inner = in2;
                        }
                        std::reverse(cs.begin(), cs.end());  // innermost first
                        TP t = inner->eval_with(inputs, degree_p1 + cs.size(), cache);
                        if (cs.size() == 1) return t.observe_step(v, inputs.at(v), cs[0], degree_p1);
                        return t.observe_chain(v, inputs.at(v), cs, degree_p1);
                    }
                }
                // (d/dv G) * const — one level of the continuous-rate Poisson observation chain (gf.rs:703-706): the
                // backend's fused derive_scale = derivative / truncate (the Derivative arm below) and the constant factor
                if (x.b.p->kind == Const && x.a.p->kind == Derivative && x.a.p->order == 1) {
                    size_t v = x.a.p->var;
                    TP t = x.a.p->a.eval_with(inputs, degree_p1 + 1, cache);
                    return t.derive_scale(v, x.b.p->c, degree_p1);
                }
                TP g = x.a.eval_with(inputs, degree_p1, cache);
                TP h = x.b.eval_with(inputs, degree_p1, cache);
                return g * h;
            }
            case Div: { TP g = x.a.eval_with(inputs, degree_p1, cache); TP h = x.b.eval_with(inputs, degree_p1, cache); return g / h; }
            case Polynomial: {
                TP taylor = TP::from_array(x.coeffs, x.shape, Dims(x.shape.size(), UMAX));
                for (size_t v = 0; v < inputs.size(); ++v) taylor = taylor.subst_var(v, TP::var(v, inputs[v], degree_p1));
                size_t ndim = taylor.num_vars();
                if (ndim > inputs.size()) {
                    if (ndim != inputs.size() + 1) throw std::runtime_error("assertion failed: ndim == inputs.len() + 1");
                    taylor = taylor.remove_last_variable();
                }
                return taylor.extend_to_dim(inputs.size(), degree_p1).truncate_to_degree_p1(degree_p1);
            }
            case Exp: return x.a.eval_with(inputs, degree_p1, cache).exp();
            case Log: return x.a.eval_with(inputs, degree_p1, cache).log();
            case Max: {
                TP s = x.a.eval_with(inputs, degree_p1, cache), t = x.b.eval_with(inputs, degree_p1, cache);
                return TP::from(s.constant_term().max(t.constant_term()));
            }
            case Pow: return x.a.eval_with(inputs, degree_p1, cache).pow(x.n);
            case UniformMgf: {
                TP xx = x.a.eval_with(inputs, degree_p1, cache);
                if (xx.constant_term().is_zero()) {
                    TP y = TP::var_at_zero(0, degree_p1 + 1);
                    TP numerator = y.exp() - TP::one();
                    Dims shape;
                    std::vector<T> arr = numerator.to_vector(&shape);  // 1-d; divide by y: drop entry 0
                    std::vector<T> sliced(arr.begin() + 1, arr.end());
                    TP fraction = TP::from_array(sliced, Dims{sliced.size()}, Dims{degree_p1});
                    return fraction.subst_var(0, xx);
                }
                TP numerator = xx.exp() - TP::one();
                return (numerator / xx).truncate_to_degree_p1(degree_p1);
            }
            case Subst: {
                Inputs new_inputs = inputs;
                TP subst;
                T c;
                if (!subst_shortcut(*x.b.p, inputs, degree_p1, subst, c)) {
                    subst = x.b.eval_with(inputs, degree_p1, cache);
                    c = subst.constant_term();
                    subst = subst - TP::from(c);
                }
                if (x.var < inputs.size()) new_inputs[x.var] = c;
                else {
                    if (x.var != inputs.size()) throw std::runtime_error("assertion failed: v.id() == inputs.len()");
                    new_inputs.push_back(c);
                }
                TP taylor = x.a.eval_with(new_inputs, degree_p1, cache);
                TP result = taylor.subst_var(x.var, subst);
                if (taylor.shape().size() > inputs.size()) {
                    if (taylor.shape().size() != inputs.size() + 1) throw std::runtime_error("assertion failed: taylor.shape().len() == inputs.len() + 1");
                    result = result.remove_last_variable();
                }
                return result;
            }
            case Derivative: return x.a.eval_with(inputs, degree_p1 + x.order, cache).derivative_truncated(x.var, x.order, degree_p1);  // = .derivative(v, n).truncate_to_degree_p1(d), one call
            case TaylorPolynomial: {
                Inputs ni = inputs;
                ni.at(x.var) = T::zero();
                size_t max_order = 0;
                for (size_t o : x.orders) max_order = std::max(max_order, o);
                TP taylor = x.a.eval_with(ni, degree_p1 + max_order, cache);
                TP result = taylor.taylor_polynomial_terms(x.var, x.orders);
                result = result.subst_var(x.var, TP::var(x.var, inputs[x.var], degree_p1));
                return result.truncate_to_degree_p1(degree_p1);
            }
            case TaylorCoeffAtZero: return eval_taylor_coeff_at_zero(x.a, x.var, x.order, inputs, degree_p1, cache);
            case TaylorCoeff: return x.a.eval_with(inputs, degree_p1 + x.order, cache).taylor_expansion_of_coeff(x.var, x.order).truncate_to_degree_p1(degree_p1);
            case ShiftTaylorAtZero: {
                if (inputs.at(x.var).is_zero())
                    return x.a.eval_with(inputs, degree_p1 + x.order, cache).shift_down(x.var, x.order).truncate_to_degree_p1(degree_p1);
                Dims orders;
                for (size_t i = 0; i < x.order; ++i) orders.push_back(i);
                GenFun first = x.a.taylor_polynomial_at_zero(x.var, orders);
                GenFun add_mass = first.substitute_var(x.var, one());
                GenFun h = (x.a - first) / var(x.var).pow((uint32_t)x.order) + add_mass;
                return h.eval_with(inputs, degree_p1, cache);
            }
        }
        throw std::runtime_error("unreachable");
    }

    // The substitution of a Subst node is almost always a constant (marginalisation, `[t -> 1]`) or `k * var` (the scalings of
    // the observe statements, gf.rs:496): six TaylorPoly operations on one- and two-element tensors per node and input point
    // (from, var, mul, constant_term, from, sub) — 60 % of the 566 000 backend calls of a mixture run, all on the calling
    // thread.  Their values follow from the reference's own definitions on such tensors: `from(k) * var(v, x, d)` takes Mul's
    // constant path, k * e for both elements (mt:1041-1047; k * 1 included, k = 1 returns the operand, which k * e reproduces
    // bit for bit), `p - from(c)` touches element 0 only (mt:919-926).  So `subst - constant_term(subst)` is formed here with
    // the evaluator's own number type and handed to the backend as ONE tensor.  Anything else — k = 0 (Mul's zero shortcut
    // changes the shape), a degree below 2, other node kinds — takes the general path.  GFH_SUBST_SHORTCUT=0: always (A/B, tests).
    static bool subst_shortcut_on() {
        static const bool on = [] {
            const char* e = getenv("GFH_SUBST_SHORTCUT");
            return !e || e[0] != '0';
        }();
        return on;
    }
    static bool subst_shortcut(const Node& b, const Inputs& inputs, size_t degree_p1, TP& subst, T& c) {
        if (!subst_shortcut_on()) return false;
        if (b.kind == Const) {  // from(c) - from(c): two 0-dimensional tensors, element 0 (mt:919-926)
            c = b.c;
            subst = TP::from(c - c);
            return true;
        }
        if (b.kind != Mul || degree_p1 < 2) return false;
        const Node *kn = b.a.p.get(), *vn = b.b.p.get();
        if (kn->kind == Var && vn->kind == Const) std::swap(kn, vn);
        if (kn->kind != Const || vn->kind != Var || kn->c.is_zero() || vn->var >= inputs.size()) return false;
        const T e0 = kn->c * inputs[vn->var], e1 = kn->c * T::one();
        c = e0;
        subst = TP::affine(vn->var, e0 - c, e1, degree_p1);
        return true;
    }

    // recognisers (gf.rs:840-914)
    static bool recognize_discrete_poisson(const GenFun& g, size_t aux, size_t& pv, T& lambda, GenFun& inner) {
        const Node& s = *g.p;
        if (s.kind != Subst) return false;
        const Node& m = *s.b.p;
        if (m.kind != Mul) return false;
        if (m.a != var(s.var)) return false;
        const Node& e = *m.b.p;
        if (e.kind != Exp) return false;
        const Node& mm = *e.a.p;
        if (mm.kind != Mul) return false;
        if (mm.a.p->kind != Const) return false;
        if (mm.b == var(aux) - constant(T::one())) { pv = s.var; lambda = mm.a.p->c; inner = s.a; return true; }
        return false;
    }
    static bool recognize_continuous_poisson(const GenFun& g, size_t aux, size_t& pv, T& lambda, GenFun& inner) {
        const Node& s = *g.p;
        if (s.kind != Subst) return false;
        const Node& ad = *s.b.p;
        if (ad.kind != Add) return false;
        if (ad.a != var(s.var)) return false;
        const Node& mm = *ad.b.p;
        if (mm.kind != Mul) return false;
        if (mm.a.p->kind != Const) return false;
        if (mm.b == var(aux) - constant(T::one())) { pv = s.var; lambda = mm.a.p->c; inner = s.a; return true; }
        return false;
    }
    static bool recognize_negative_binomial(const GenFun& g, size_t aux, size_t& pv, T& pp, GenFun& inner) {
        const Node& s = *g.p;
        if (s.kind != Subst) return false;
        const Node& m = *s.b.p;
        if (m.kind != Mul) return false;
        if (m.a != var(s.var)) return false;
        const Node& d = *m.b.p;
        if (d.kind != Div) return false;
        if (d.a.p->kind != Const) return false;
        T pr = d.a.p->c;
        GenFun expected = one() - constant(T::one() - pr) * var(aux);
        if (d.b == expected) { pv = s.var; pp = pr; inner = s.a; return true; }
        return false;
    }

    // The observation chains of the two Poisson recognisers (gf.rs:684-706) depend on (g, v, order) only, and a program
    // evaluates the same observe statement at thousands of input points: the 3 * order + 4 nodes are built once per
    // top-level eval() and kept here.  While a chain is being evaluated it is TAKEN OUT of the table, so its nodes have
    // exactly the reference counts the freshly built chain has in the reference — eval_with's "shared" test, hence the
    // result cache and the sequence of TaylorPoly operations, are unchanged.
    struct ChainKey {
        const Node* n;
        size_t v, order;
        bool operator==(const ChainKey& o) const { return n == o.n && v == o.v && order == o.order; }
    };
    struct ChainKeyHash {
        size_t operator()(const ChainKey& k) const { return std::hash<const Node*>()(k.n) ^ (k.v * 0x9e3779b97f4a7c15ull) ^ (k.order * 0xc2b2ae3d27d4eb4full); }
    };
    struct ChainVal { GenFun g, gf; };  // g keeps the key's node alive (its address cannot be recycled while it is a key)
    typedef std::unordered_map<ChainKey, ChainVal, ChainKeyHash, std::equal_to<ChainKey>, gft_small::Alloc<std::pair<const ChainKey, ChainVal>>> ChainTable;
    static ChainTable& chain_table() {
        static thread_local ChainTable t;
        return t;
    }
    static TP eval_chain(const GenFun& g, size_t v, size_t order, GenFun gf, const Inputs& inputs, size_t degree_p1, EvalCache& cache) {
        TP r = gf.eval_with(inputs, degree_p1, cache).truncate_to_degree_p1(degree_p1);
        if (!chain_table_on()) return r;
        ChainTable& tab = chain_table();
        if (tab.size() >= 8192) tab.clear();
        tab[ChainKey{g.p.get(), v, order}] = ChainVal{g, std::move(gf)};
        return r;
    }

    static bool chain_table_on() {  // GFH_CHAIN_TABLE=0: every evaluation rebuilds its chain, as the reference does (A/B, tests)
        static const bool on = [] {
            const char* e = getenv("GFH_CHAIN_TABLE");
            return !e || e[0] != '0';
        }();
        return on;
    }
    static TP eval_taylor_coeff_at_zero(const GenFun& g, size_t v, size_t order, const Inputs& inputs, size_t degree_p1, EvalCache& cache) {
        if (chain_table_on()) {
            ChainTable& tab = chain_table();
            auto it = tab.find(ChainKey{g.p.get(), v, order});
            if (it != tab.end() && it->second.gf.p) {  // (the entry stays, emptied, while its chain is out)
                GenFun gf = std::move(it->second.gf);
                it->second.gf.p = nullptr;
                return eval_chain(g, v, order, std::move(gf), inputs, degree_p1, cache);
            }
        }
        size_t pv;
        T lambda;
        GenFun inner;
        // (the chain is handed over as its only owner: in the reference the local `replacement` is a second owner of that one
        // node while the chain is evaluated, which makes eval cache its value under a key nobody asks for again — a chain that
        // is kept must not have such an entry, or later evaluations would find it and skip operations the reference performs)
        if (recognize_discrete_poisson(g, v, pv, lambda, inner)) {
            GenFun gf = inner;
            for (size_t k = 1; k <= order; ++k) gf = gf.derive(pv, 1) * var(pv) * constant(lambda / T::from_u32((uint32_t)k));
            gf = gf.substitute_var(pv, constant((-lambda).exp()) * var(pv));
            return eval_chain(g, v, order, std::move(gf), inputs, degree_p1, cache);
        }
        if (recognize_continuous_poisson(g, v, pv, lambda, inner)) {
            GenFun gf = inner;
            for (size_t k = 1; k <= order; ++k) gf = gf.derive(pv, 1) * constant(lambda / T::from_u32((uint32_t)k));
            gf = gf.substitute_var(pv, var(pv) - constant(lambda));
            return eval_chain(g, v, order, std::move(gf), inputs, degree_p1, cache);
        }
        T pr;
        if (recognize_negative_binomial(g, v, pv, pr, inner)) {
            std::vector<T> lahs_cur(1, T::one());
            T one_mp = T::one() - pr;
            for (size_t d = 1; d <= order; ++d) {
                std::vector<T> next;
                for (size_t i = 0; i <= d; ++i) {
                    T lah_dm1_i = i < lahs_cur.size() ? lahs_cur[i] : T::zero();
                    T lah_dm1_im1 = (1 <= i && i <= lahs_cur.size()) ? lahs_cur[i - 1] : T::zero();
                    T lah_d_i = one_mp / T::from_u32((uint32_t)d) * (lah_dm1_i * T::from_u32((uint32_t)(d + i - 1)) + lah_dm1_im1);
                    next.push_back(lah_d_i);
                }
                lahs_cur = next;
            }
            TP sum = TP::zero_with(Dims(inputs.size(), degree_p1));
            Inputs ni = inputs;
            ni.at(pv) = pr * inputs[pv];
            TP inner_result = inner.eval_with(ni, degree_p1 + order, cache);
            TP p_pow = TP::one();
            TP pv_tp = TP::var(pv, inputs[pv], degree_p1);
            TP p_pv = TP::from(pr) * pv_tp;
            for (const T& lah : lahs_cur) {
                TP subst = TP::from(pr) * TP::var_at_zero(pv, degree_p1);
                sum = sum.add_scaled(inner_result.subst_var(pv, subst) * p_pow, lah);  // sum + (term * from(lah)), one pass
                p_pow = p_pow * p_pv;
                inner_result = inner_result.derivative(pv, 1);
            }
            return sum.truncate_to_degree_p1(degree_p1);
        }
        Inputs ni = inputs;
        TP result;
        if (v == ni.size()) {
            ni.push_back(T::zero());
            result = g.eval_with(ni, degree_p1 + order, cache).coefficients_of_term(v, order).remove_last_variable();
        } else {
            ni.at(v) = T::zero();
            result = g.eval_with(ni, degree_p1 + order, cache).coefficients_of_term(v, order);
        }
        return result.truncate_to_degree_p1(degree_p1);
    }
};

// ---- probabilities and moments (gf.rs:937-1086) ------------------------------------------------------------
template <class T>
std::vector<T> probs_taylor(const GenFun<T>& pgf, size_t v, const VarSupport& vi, size_t max_n) {
    if (!vi[v].is_discrete()) throw std::runtime_error("Can only compute probabilities for discrete variables");
    size_t nv = vi.num_vars();
    std::vector<T> substs;
    for (size_t i = 0; i < nv; ++i) substs.push_back(vi[i].is_discrete() ? T::one() : T::zero());
    substs[v] = T::zero();
    Poly<T> expansion = pgf.eval(substs, max_n + 1);
    Dims index(nv, 0);
    std::vector<T> probs;
    for (size_t i = 0; i < max_n; ++i) { index[v] = i; probs.push_back(expansion.coefficient(index)); }
    return probs;
}

template <class T>
std::pair<T, std::vector<T>> factorial_moments_to_moments(const std::vector<T>& fm) {
    size_t len = fm.size();
    std::vector<std::vector<T>> st(len, std::vector<T>(len, T::zero()));
    for (size_t n = 0; n < len; ++n) {
        st[n][0] = T::zero();
        st[n][n] = T::one();
        for (size_t k = 1; k < n; ++k) st[n][k] = st[n - 1][k - 1] + T::from_u32((uint32_t)k) * st[n - 1][k];
    }
    T total = fm[0];
    std::vector<T> moments(len - 1, T::zero());
    for (size_t n = 1; n < len; ++n)
        for (size_t k = 0; k <= n; ++k) moments[n - 1] = moments[n - 1] + st[n][k] * fm[k];
    for (auto& m : moments) m = m / total;
    return {total, moments};
}

template <class T>
std::pair<T, std::vector<T>> moments_taylor(const GenFun<T>& pgf, size_t v, const VarSupport& vi, size_t limit) {
    size_t nv = vi.num_vars();
    std::vector<T> substs;
    for (size_t i = 0; i < nv; ++i) substs.push_back(vi[i].is_discrete() ? T::one() : T::zero());
    Poly<T> expansion = pgf.eval(substs, limit);
    std::vector<T> result;
    Dims index(nv, 0);
    T factor = T::one();
    for (size_t i = 0; i < limit; ++i) {
        index[v] = i;
        result.push_back(expansion.coefficient(index) * factor);
        factor = factor * T::from_u32((uint32_t)(i + 1));
    }
    if (vi[v].is_discrete()) return factorial_moments_to_moments(result);
    T total = result[0];
    std::vector<T> moments;
    for (size_t i = 1; i < result.size(); ++i) moments.push_back(result[i] / total);
    return {total, moments};
}

template <class T>
std::pair<T, std::vector<T>> moments_to_central_moments(const std::vector<T>& moments) {
    size_t len = moments.size() + 1;
    T mean = moments[0];
    std::vector<std::vector<T>> bc(len, std::vector<T>(len, T::zero()));
    for (size_t n = 0; n < len; ++n) {
        bc[n][0] = T::one();
        bc[n][n] = T::one();
        for (size_t k = 1; k < n; ++k) bc[n][k] = bc[n - 1][k - 1] + bc[n - 1][k];
    }
    T neg_mean = -mean;
    std::vector<T> cm(len - 2, T::zero());
    for (size_t n = 2; n < len; ++n) {
        for (size_t k = 1; k <= n; ++k) cm[n - 2] = cm[n - 2] + bc[n][k] * neg_mean.pow((uint32_t)(n - k)) * moments[k - 1];
        cm[n - 2] = cm[n - 2] + neg_mean.pow((uint32_t)n);
    }
    return {mean, cm};
}

template <class T>
std::pair<T, std::vector<T>> central_to_standardized_moments(const std::vector<T>& cm) {
    T variance = cm[0];
    T sigma = variance.sqrt();
    std::vector<T> out;
    for (size_t i = 0; i + 1 < cm.size(); ++i) {
        const T& x = cm[i + 1];
        if (x.is_zero() && !variance.is_nan() && !variance.is_zero()) out.push_back(x);
        else {
            T sp = (i % 2 == 0) ? sigma.pow((uint32_t)(i + 3)) : variance.pow((uint32_t)((i + 3) / 2));
            out.push_back(x / sp);
        }
    }
    return {variance, out};
}

}  // namespace gfh
