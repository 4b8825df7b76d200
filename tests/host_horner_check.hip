// CPU-only check of the host tier's Horner step (gft_host.hpp): the finite regime in RUNS of equal terms (round 6) against the
// element-by-element form it replaces, bit for bit, over random shapes, substitution axes, boxes and data classes (error
// intervals around zero, positive data, exact zeros / ones, an infinity).  Built with hipcc (the element functors are one
// source for both passes) and run on the host: no HIP call is made.  tests/test_host_horner_runs.py drives it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include "../genfer_amd/csrc/gft_host.hpp"

namespace gft {
unsigned long long g_host_horner_stats[16] = {0};
bool g_host_horner_runs = true;
int g_host_simd = -1;
}  // namespace gft
using namespace gft;

static unsigned long long rng_state = 88172645463325252ull;
static double urand() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0);
}
static unsigned irand(unsigned lo, unsigned hi) { return lo + (unsigned)(urand() * (hi - lo + 1)); }

int main() {
    typedef EIv E;
    size_t cases = 0, run_steps_before = 0;
    for (int trial = 0; trial < 12000; ++trial) {
        HornerArgs g;
        std::memset(&g, 0, sizeof(g));
        const int nd = (int)irand(1, 3);
        g.out.nd = nd;
        g.w = (int)irand(0, (unsigned)nd - 1);
        unsigned rs[3], sh[3], os[3], oc[3];
        const bool coeff_scalar = urand() < 0.3;
        for (int ax = 0; ax < nd; ++ax) {
            rs[ax] = irand(1, trial % 4 == 0 ? 41 : 7);  // (long lines: the vector bodies and their remainders)
            sh[ax] = rs[ax] + (ax == g.w ? irand(0, 1) : 0);
            oc[ax] = coeff_scalar ? 1 : irand(1, trial % 4 == 0 ? 43 : 8);
            os[ax] = coeff_scalar ? sh[ax] : (sh[ax] > oc[ax] ? sh[ax] : oc[ax]);
        }
        size_t stride = 1, astride = 1, nres = 1, nout = 1, na = 1;
        unsigned ashape[3];
        const int v = (int)irand(0, (unsigned)nd - 1);  // the substituted axis of the coefficient tensor
        for (int ax = 0; ax < nd; ++ax) ashape[ax] = ax == v ? irand(1, 4) : oc[ax];
        for (int ax = nd - 1; ax >= 0; --ax) {
            g.out.d[ax] = os[ax];
            g.rs[ax] = rs[ax];
            g.sh[ax] = sh[ax];
            g.oc[ax] = ax == v ? 1 : oc[ax];
            g.rstr[ax] = stride;
            stride *= rs[ax];
            g.astr[ax] = ax == v ? 0 : astride;
            astride *= ashape[ax];
            nres *= rs[ax];
            nout *= os[ax];
            na *= ashape[ax];
        }
        if (coeff_scalar) {
            na = ashape[v];
            for (int ax = 0; ax < nd; ++ax) g.astr[ax] = 0;
            g.a_base = irand(0, ashape[v] - 1);
        } else {
            size_t vs = 1;
            for (int ax = nd - 1; ax > v; --ax) vs *= ashape[ax];
            g.a_base = (size_t)irand(0, ashape[v] - 1) * vs;
        }
        g.coeff_scalar = coeff_scalar ? 1 : 0;
        g.upper = sh[g.w] - 1 < rs[g.w] ? sh[g.w] - 1 : rs[g.w];
        const int cls = (int)irand(0, 7);  // 4-6: positive data under a sign-known c (the `semi` regime), 5 with values that underflow, 6 with stray non-positive elements, 7 a positive head and a tail of error intervals around zero (switchpoint's lines)
        auto mk = [&](double centre, double width) {
            Iv r;
            r.lo = centre - width * urand();
            r.hi = centre + width * urand();
            return r;
        };
        Iv c = cls == 1 ? mk(0.3, 1e-3) : mk(0.0, 1e-12);
        if (cls >= 4 && urand() < 0.5) c = urand() < 0.5 ? mk(-0.3, 1e-3) : Iv{-1e-13 * (1.0 + urand()), 0.25 + urand()};
        const Iv m = urand() < 0.5 ? Iv{1.0, 1.0} : mk(0.9, 1e-6);
        g.c = Scalar2{c.lo, c.hi};
        g.m = Scalar2{m.lo, m.hi};
        std::vector<double> res(2 * nres), a(2 * na), o1(2 * nout, -7.0), o2(2 * nout, -7.0);
        auto fill = [&](std::vector<double>& t, size_t n) {
            for (size_t i = 0; i < n; ++i) {
                Iv x = (cls == 1 || cls >= 4) ? mk(0.5, 0.4) : mk(0.0, cls == 2 ? 1e-300 : 1e-20);
                if (cls == 7 && 2 * i >= n) x = mk(0.0, 1e-20);
                if (cls == 5 && urand() < 0.05) x = Iv{4.9e-324 * (double)irand(1, 3), 1e-320};
                if (cls == 6 && urand() < 0.05) x = urand() < 0.5 ? mk(0.0, 1e-3) : (urand() < 0.5 ? Iv{1.0, 1.0} : Iv{0.0, 0.0});
                if (cls == 3 && urand() < 0.1) x = urand() < 0.5 ? Iv{0.0, 0.0} : Iv{1.0, 1.0};
                if (cls == 3 && urand() < 0.02) x.hi = __builtin_inf();
                t[i] = x.lo;
                t[n + i] = x.hi;
            }
        };
        fill(res, nres);
        fill(a, na);
        g_host_horner_runs = true;
        g_host_simd = -1;
        HK<E>::horner_linear(res.data(), nres, a.data(), na, o1.data(), nout, g);
        for (int simd = 0; simd <= 1; ++simd) {  // the narrower builds of the same runs (AVX2 at most; baseline x86-64)
            std::vector<double> o3(2 * nout, -7.0);
            g_host_simd = simd;
            HK<E>::horner_linear(res.data(), nres, a.data(), na, o3.data(), nout, g);
            g_host_simd = -1;
            if (std::memcmp(o1.data(), o3.data(), sizeof(double) * 2 * nout) != 0) {
                std::printf("MISMATCH (widest vs g_host_simd = %d) trial %d\n", simd, trial);
                return 1;
            }
        }
        g_host_horner_runs = false;
        HK<E>::horner_linear(res.data(), nres, a.data(), na, o2.data(), nout, g);
        if (std::memcmp(o1.data(), o2.data(), sizeof(double) * 2 * nout) != 0) {
            std::printf("MISMATCH trial %d nd=%d w=%d cls=%d coeff_scalar=%d\n", trial, nd, g.w, cls, (int)coeff_scalar);
            return 1;
        }
        ++cases;
        (void)run_steps_before;
    }
    if (std::getenv("HOST_HORNER_TIME")) {  // ns per element of the two forms on a switchpoint-like step (positive data, c around zero)
        for (int w = 0; w < 2; ++w) {
            HornerArgs g;
            std::memset(&g, 0, sizeof(g));
            g.out.nd = 2;
            g.w = w;
            const unsigned rs[2] = {(unsigned)(std::getenv("HH_R0") ? atoi(std::getenv("HH_R0")) : 24), (unsigned)(std::getenv("HH_R1") ? atoi(std::getenv("HH_R1")) : 12)};
            unsigned sh[2] = {rs[0], rs[1]};
            sh[w] += 1;
            size_t nres = rs[0] * rs[1], nout = sh[0] * sh[1];
            g.out.d[0] = sh[0]; g.out.d[1] = sh[1];
            g.rs[0] = rs[0]; g.rs[1] = rs[1];
            g.sh[0] = sh[0]; g.sh[1] = sh[1];
            g.oc[0] = w == 0 ? 1 : rs[0]; g.oc[1] = w == 1 ? 1 : rs[1];
            g.rstr[0] = rs[1]; g.rstr[1] = 1;
            g.astr[0] = w == 0 ? 0 : rs[1]; g.astr[1] = w == 1 ? 0 : 1;
            g.upper = rs[w];
            g.c = Scalar2{-1e-13, 2e-13};
            g.m = Scalar2{0.9, 0.9000001};
            std::vector<double> res(2 * nres), a(2 * nres), o(2 * nout);
            for (size_t i = 0; i < nres; ++i) {
                res[i] = 0.1 + 0.5 * urand(); res[nres + i] = res[i] + 0.1;
                a[i] = 0.1 + 0.5 * urand(); a[nres + i] = a[i] + 0.1;
            }
            for (int runs = 3; runs >= 0; --runs) {  // 3: runs, widest vectors; 2: AVX2; 1: baseline; 0: element form
                g_host_horner_runs = runs != 0;
                g_host_simd = runs == 3 ? -1 : (runs == 2 ? 1 : 0);
                timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                const int reps = 20000;
                for (int r = 0; r < reps; ++r) HK<E>::horner_linear(res.data(), nres, a.data(), nres, o.data(), nout, g);
                clock_gettime(CLOCK_MONOTONIC, &t1);
                const double ns = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / ((double)reps * nout);
                std::printf("w=%d runs=%d: %.2f ns per element (%zu elements per step, checksum %g)\n", w, runs, ns, nout, o[nout / 2]);
            }
        }
    }
    std::printf("host_horner ok: %zu cases\n", cases);
    return 0;
}
