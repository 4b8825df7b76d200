// Host interpreter, part 1: the TaylorPoly value type used by the evaluator — a thin
// RAII wrapper over the C ABI of include/gftaylor.h, bound at run time (dlopen + prefix) so the same
// interpreter drives libgftaylor.so (HIP, the product) or any other library exporting the same
// surface.  Mirrors the reference's `TaylorPoly<T>` method names (src/multivariate_taylor.rs).
#pragma once
#include <dlfcn.h>

#include "../gft_small_alloc.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <cstddef>
#include <cstdint>
#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace gfh {

typedef std::vector<size_t> Dims;
constexpr size_t UMAX = SIZE_MAX;

struct Api {
    void* lib = nullptr;
    int width = 1;
    const char* (*last_error)();
    int (*width_fn)();
    void* (*from_host)(const double*, const size_t*, const size_t*, size_t);
    void* (*scalar)(const double*);
    void* (*zero_with)(const size_t*, size_t);
    void* (*var)(size_t, const double*, size_t);
    void* (*var_at_zero)(size_t, size_t);
    void* (*var_with_degrees_p1)(size_t, const double*, const size_t*, size_t);
    void* (*clone)(const void*);
    void (*free)(void*);
    size_t (*num_vars)(const void*);
    size_t (*numel)(const void*);
    void (*shape)(const void*, size_t*);
    void (*degrees_p1)(const void*, size_t*);
    int (*to_host)(const void*, double*);
    int (*is_zero)(const void*);
    int (*is_one)(const void*);
    int (*constant_term)(const void*, double*);
    int (*extract_constant)(const void*, double*);
    int (*coefficient)(const void*, const size_t*, size_t, double*);
    void* (*add)(const void*, const void*);
    void* (*sub)(const void*, const void*);
    void* (*mul)(const void*, const void*);
    void* (*div)(const void*, const void*);
    void* (*neg)(const void*);
    void* (*add_scaled)(const void*, const void*, const double*);
    void* (*exp)(const void*);
    void* (*log)(const void*);
    void* (*pow)(const void*, uint32_t);
    void* (*derivative)(const void*, size_t, size_t);
    void* (*taylor_expansion_of_coeff)(const void*, size_t, size_t);
    void* (*shift_down)(const void*, size_t, size_t);
    void* (*subst_var)(const void*, size_t, const void*);
    void* (*observe_step)(const void*, size_t, const double*, const double*, size_t);
    void* (*derive_scale)(const void*, size_t, const double*, size_t);
    void* (*observe_chain)(const void*, size_t, const double*, const double*, size_t, size_t);
    void* (*derivative_truncated)(const void*, size_t, size_t, size_t);
    void* (*coefficients_of_term)(const void*, size_t, size_t);
    void* (*taylor_polynomial_terms)(const void*, size_t, const size_t*, size_t);
    void* (*truncate_to_degree_p1)(const void*, size_t);
    void* (*remove_last_variable)(const void*);
    void* (*extend_to_dim)(const void*, size_t, size_t);

    // One table per (library, prefix) for the life of the process: value wrappers keep a plain pointer to theirs (Poly::H).
    static std::shared_ptr<Api> load(const std::string& path, const std::string& prefix) {
        // (never destroyed: a wrapper dropped during static destruction still finds its table)
        static auto& loaded = *new std::map<std::pair<std::string, std::string>, std::shared_ptr<Api>>;
        static auto& mu = *new std::mutex;  // (two host threads may run programs side by side: bench.py's replicas do not, callers may)
        std::lock_guard<std::mutex> lock(mu);
        auto it = loaded.find({path, prefix});
        if (it != loaded.end()) return it->second;
        auto a = load_new(path, prefix);
        loaded[{path, prefix}] = a;
        return a;
    }
    static std::shared_ptr<Api> load_new(const std::string& path, const std::string& prefix) {
        auto a = std::make_shared<Api>();
        a->lib = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!a->lib) throw std::runtime_error(std::string("cannot load backend library: ") + dlerror());
        auto sym = [&](const char* name) -> void* {
            std::string full = prefix + name;
            void* p = dlsym(a->lib, full.c_str());
            if (!p) throw std::runtime_error("backend library does not export " + full);
            return p;
        };
#define GFH_BIND(f) a->f = reinterpret_cast<decltype(a->f)>(sym(#f))
        GFH_BIND(last_error);
        a->width_fn = reinterpret_cast<int (*)()>(sym("width"));
        GFH_BIND(from_host); GFH_BIND(scalar); GFH_BIND(zero_with); GFH_BIND(var); GFH_BIND(var_at_zero);
        GFH_BIND(var_with_degrees_p1); GFH_BIND(clone); GFH_BIND(free); GFH_BIND(num_vars); GFH_BIND(numel);
        GFH_BIND(shape); GFH_BIND(degrees_p1); GFH_BIND(to_host); GFH_BIND(is_zero); GFH_BIND(is_one);
        GFH_BIND(constant_term); GFH_BIND(extract_constant); GFH_BIND(coefficient); GFH_BIND(add); GFH_BIND(sub);
        GFH_BIND(mul); GFH_BIND(div); GFH_BIND(neg); GFH_BIND(add_scaled); GFH_BIND(exp); GFH_BIND(log); GFH_BIND(pow);
        GFH_BIND(derivative); GFH_BIND(taylor_expansion_of_coeff); GFH_BIND(shift_down); GFH_BIND(subst_var); GFH_BIND(observe_step); GFH_BIND(derive_scale); GFH_BIND(observe_chain); GFH_BIND(derivative_truncated);
        GFH_BIND(coefficients_of_term); GFH_BIND(taylor_polynomial_terms); GFH_BIND(truncate_to_degree_p1);
        GFH_BIND(remove_last_variable); GFH_BIND(extend_to_dim);
#undef GFH_BIND
        a->width = a->width_fn();
        return a;
    }
};

// GFH_TRACE_SIZES=1: histogram of TaylorPoly operations by (op, power-of-two bucket of the largest tensor involved),
// printed at exit — the op / size mix a program really issues (what the size-threshold dispatch is tuned on).
struct SizeTrace {
    bool on = getenv("GFH_TRACE_SIZES") != nullptr;
    std::map<std::pair<std::string, size_t>, size_t> counts;
    void hit(const char* op, size_t n) {
        size_t b = 1;
        while (b < n) b <<= 1;
        counts[{op, b}]++;
    }
    ~SizeTrace() {
        if (!on) return;
        for (auto& kv : counts) fprintf(stderr, "[gfh sizes] %-28s <=%-9zu %zu\n", kv.first.first.c_str(), kv.first.second, kv.second);
    }
};
inline SizeTrace& size_trace() { static SizeTrace t; return t; }

// The value type: shared immutable handle (clone = refcount, like the ABI's O(1) clone).
// Reference counts of the interpreter's own objects (GenFun nodes, handle wrappers) are NOT atomic, like the reference's `Rc`
// (generating_function.rs:14-16): an evaluation lives on one thread, and with the process's other threads around (the library's
// launch thread, Python's) std::shared_ptr would pay a locked increment / decrement 10^7 times per program (13 % of switchpoint's
// calling thread).  libstdc++'s shared_ptr with the single-threaded lock policy is exactly that type.
template <class U>
using Rc = std::__shared_ptr<U, __gnu_cxx::_S_single>;
template <class U, class A, class... Args>
inline Rc<U> rc_allocate(const A& a, Args&&... args) {
    return std::__allocate_shared<U, __gnu_cxx::_S_single>(a, std::forward<Args>(args)...);
}

template <class T>
class Poly {
    // (the wrapper keeps a plain pointer to its backend's table — tables live as long as the process, Api::load —, so a handle
    // that outlives a re-bind still frees itself through the library that made it, and 10^6 wrappers per program do not each
    // count a reference on the one table)
    struct H {
        const Api* api = nullptr;
        void* h = nullptr;
        H() {}
        H(const H&) = delete;
        H& operator=(const H&) = delete;
        ~H() { if (h) api->free(h); }
    };
    Rc<H> p_;
    static std::shared_ptr<Api>& api_slot() { static std::shared_ptr<Api> a; return a; }
    static Poly wrap(void* h) {
        auto& a = api_slot();
        if (!h) throw std::runtime_error(std::string("TaylorPoly backend error: ") + a->last_error());
        Poly r;
        r.p_ = rc_allocate<H>(gft_small::Alloc<H>());  // small-block lists: ../gft_small_alloc.hpp
        r.p_->api = a.get();
        r.p_->h = h;
        return r;
    }
    void* h() const { return p_->h; }
    static size_t nel(const Poly& p) { return api().numel(p.h()); }
    static Poly traced(const char* op, size_t n_in, Poly r) {
        if (size_trace().on) size_trace().hit(op, std::max(n_in, nel(r)));
        return r;
    }

  public:
    static void bind(std::shared_ptr<Api> a) {
        if (a->width != T::WIDTH) throw std::runtime_error("backend element width does not match the number type");
        api_slot() = a;
    }
    static Api& api() { return *api_slot(); }

    static Poly from_array(const std::vector<T>& data, const Dims& shape, const Dims& degs) {  // mt:33-41
        size_t n = data.size();
        std::vector<double> planes(n * T::WIDTH);
        for (size_t i = 0; i < n; ++i) data[i].store_plane(planes.data(), n, i);
        return wrap(api().from_host(planes.data(), shape.data(), degs.data(), shape.size()));
    }
    static Poly from(const T& x) { double b[2]; x.store(b); return wrap(api().scalar(b)); }          // mt:626-630
    // the two-element tensor [e0, e1] along axis v with degrees_p1 = [d] * (v + 1): what `var(v, x, d)` (mt:239-248, d >= 2)
    // becomes under elementwise operations (GenFun::subst_shortcut)
    static Poly affine(size_t v, const T& e0, const T& e1, size_t d) {
        double planes[2 * T::WIDTH];
        e0.store_plane(planes, 2, 0);
        e1.store_plane(planes, 2, 1);
        size_t shape[33], degs[33];
        if (v >= 32) throw std::runtime_error("more than 32 variables are not supported");
        for (size_t i = 0; i <= v; ++i) {
            shape[i] = i == v ? 2 : 1;
            degs[i] = d;
        }
        return wrap(api().from_host(planes, shape, degs, v + 1));
    }
    static Poly zero() { return from(T::zero()); }
    static Poly one() { return from(T::one()); }
    static Poly zero_with(const Dims& d) { return wrap(api().zero_with(d.data(), d.size())); }        // mt:208-216
    static Poly var(size_t v, const T& x, size_t len) { double b[2]; x.store(b); return wrap(api().var(v, b, len)); }  // mt:239-248
    static Poly var_at_zero(size_t v, size_t len) { return wrap(api().var_at_zero(v, len)); }          // mt:228-237
    static Poly var_with_degrees_p1(size_t v, const T& x, const Dims& d) {                               // mt:250-259
        double b[2]; x.store(b);
        return wrap(api().var_with_degrees_p1(v, b, d.data(), d.size()));
    }

    size_t num_vars() const { return api().num_vars(h()); }
    Dims shape() const { Dims d(num_vars()); api().degrees_p1(h(), d.data()); return d; }               // mt:53-56
    Dims coeffs_shape() const { Dims d(num_vars()); api().shape(h(), d.data()); return d; }
    bool is_constant() const { return api().numel(h()) == 1; }
    bool is_zero() const { return api().is_zero(h()) == 1; }
    bool is_one() const { return api().is_one(h()) == 1; }
    T constant_term() const { double b[2] = {0, 0}; api().constant_term(h(), b); return T::load(b); }   // mt:296-299
    bool extract_constant(T& out) const {                                                                // mt:262-269
        double b[2] = {0, 0};
        if (api().extract_constant(h(), b) != 1) return false;
        out = T::load(b);
        return true;
    }
    T coefficient(const Dims& idx) const {                                                               // mt:314-339
        double b[2] = {0, 0};
        if (api().coefficient(h(), idx.data(), idx.size(), b) != 0)
            throw std::runtime_error(std::string("coefficient: ") + api().last_error());
        return T::load(b);
    }
    // into_array(): host copy of the stored coefficients (plane-major -> T)
    std::vector<T> to_vector(Dims* shape_out = nullptr) const {
        size_t n = api().numel(h());
        std::vector<double> planes(n * T::WIDTH);
        api().to_host(h(), planes.data());
        std::vector<T> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = T::load_plane(planes.data(), n, i);
        if (shape_out) *shape_out = coeffs_shape();
        return out;
    }

    Poly operator+(const Poly& o) const { return traced("add", std::max(nel(*this), nel(o)), wrap(api().add(h(), o.h()))); }
    Poly operator-(const Poly& o) const { return traced("sub", std::max(nel(*this), nel(o)), wrap(api().sub(h(), o.h()))); }
    Poly operator*(const Poly& o) const { return traced("mul", std::max(nel(*this), nel(o)), wrap(api().mul(h(), o.h()))); }
    Poly operator/(const Poly& o) const { return traced("div", std::max(nel(*this), nel(o)), wrap(api().div(h(), o.h()))); }
    Poly operator-() const { return traced("neg", nel(*this), wrap(api().neg(h()))); }
    Poly exp() const { return traced("exp", nel(*this), wrap(api().exp(h()))); }
    Poly log() const { return traced("log", nel(*this), wrap(api().log(h()))); }
    Poly pow(uint32_t e) const { return traced("pow", nel(*this), wrap(api().pow(h(), e))); }
    Poly derivative(size_t v, size_t n) const { return traced("derivative", nel(*this), wrap(api().derivative(h(), v, n))); }
    Poly derivative_truncated(size_t v, size_t n, size_t d) const { return traced("derivative_truncated", nel(*this), wrap(api().derivative_truncated(h(), v, n, d))); }
    Poly taylor_expansion_of_coeff(size_t v, size_t n) const { return traced("taylor_expansion_of_coeff", nel(*this), wrap(api().taylor_expansion_of_coeff(h(), v, n))); }
    Poly shift_down(size_t v, size_t n) const { return traced("shift_down", nel(*this), wrap(api().shift_down(h(), v, n))); }
    Poly subst_var(size_t v, const Poly& s) const { return traced("subst_var", std::max(nel(*this), nel(s)), wrap(api().subst_var(h(), v, s.h()))); }
    Poly observe_step(size_t v, const T& x, const T& c, size_t d) const {
        double xb[2], cb[2];
        x.store(xb);
        c.store(cb);
        return wrap(api().observe_step(h(), v, xb, cb, d));
    }
    template <class Vec>
    Poly observe_chain(size_t v, const T& x, const Vec& cs, size_t d) const {
        double xb[2];
        x.store(xb);
        std::vector<double, gft_small::Alloc<double>> cb(cs.size() * T::WIDTH + 2);
        for (size_t i = 0; i < cs.size(); ++i) cs[i].store(cb.data() + i * T::WIDTH);
        return traced("observe_chain", nel(*this), wrap(api().observe_chain(h(), v, xb, cb.data(), cs.size(), d)));
    }
    Poly add_scaled(const Poly& o, const T& c) const {  // *this + o * from(c)
        double cb[2];
        c.store(cb);
        return traced("add_scaled", std::max(nel(*this), nel(o)), wrap(api().add_scaled(h(), o.h(), cb)));
    }
    Poly derive_scale(size_t v, const T& c, size_t d) const {
        double cb[2];
        c.store(cb);
        return traced("derive_scale", nel(*this), wrap(api().derive_scale(h(), v, cb, d)));
    }
    Poly coefficients_of_term(size_t v, size_t o) const { return traced("coefficients_of_term", nel(*this), wrap(api().coefficients_of_term(h(), v, o))); }
    Poly taylor_polynomial_terms(size_t v, const Dims& orders) const {
        return wrap(api().taylor_polynomial_terms(h(), v, orders.data(), orders.size()));
    }
    Poly truncate_to_degree_p1(size_t d) const { return traced("truncate", nel(*this), wrap(api().truncate_to_degree_p1(h(), d))); }
    Poly remove_last_variable() const { return wrap(api().remove_last_variable(h())); }
    Poly extend_to_dim(size_t nd, size_t d) const { return wrap(api().extend_to_dim(h(), nd, d)); }
};

}  // namespace gfh
