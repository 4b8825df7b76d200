// CPU-only check of the host tier's Horner step (gft_host.hpp): the finite regime in RUNS of equal terms (round 6) against the
// element-by-element form it replaces, bit for bit, over random shapes, substitution axes, boxes and data classes (error
// intervals around zero, positive data, exact zeros / ones, an infinity).  Built with hipcc (the element functors are one
// source for both passes) and run on the host: no HIP call is made.  tests/test_host_horner_runs.py drives it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../genfer_amd/csrc/gft_host.hpp"

namespace gft {
unsigned long long g_host_horner_stats[4] = {0, 0, 0, 0};
bool g_host_horner_runs = true;
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
    for (int trial = 0; trial < 4000; ++trial) {
        HornerArgs g;
        std::memset(&g, 0, sizeof(g));
        const int nd = (int)irand(1, 3);
        g.out.nd = nd;
        g.w = (int)irand(0, (unsigned)nd - 1);
        unsigned rs[3], sh[3], os[3], oc[3];
        const bool coeff_scalar = urand() < 0.3;
        for (int ax = 0; ax < nd; ++ax) {
            rs[ax] = irand(1, 7);
            sh[ax] = rs[ax] + (ax == g.w ? irand(0, 1) : 0);
            oc[ax] = coeff_scalar ? 1 : irand(1, 8);
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
        const int cls = (int)irand(0, 3);
        auto mk = [&](double centre, double width) {
            Iv r;
            r.lo = centre - width * urand();
            r.hi = centre + width * urand();
            return r;
        };
        const Iv c = cls == 1 ? mk(0.3, 1e-3) : mk(0.0, 1e-12), m = urand() < 0.5 ? Iv{1.0, 1.0} : mk(0.9, 1e-6);
        g.c = Scalar2{c.lo, c.hi};
        g.m = Scalar2{m.lo, m.hi};
        std::vector<double> res(2 * nres), a(2 * na), o1(2 * nout, -7.0), o2(2 * nout, -7.0);
        auto fill = [&](std::vector<double>& t, size_t n) {
            for (size_t i = 0; i < n; ++i) {
                Iv x = cls == 1 ? mk(0.5, 0.4) : mk(0.0, cls == 2 ? 1e-300 : 1e-20);
                if (cls == 3 && urand() < 0.1) x = urand() < 0.5 ? Iv{0.0, 0.0} : Iv{1.0, 1.0};
                if (cls == 3 && urand() < 0.02) x.hi = __builtin_inf();
                t[i] = x.lo;
                t[n + i] = x.hi;
            }
        };
        fill(res, nres);
        fill(a, na);
        g_host_horner_runs = true;
        HK<E>::horner_linear(res.data(), nres, a.data(), na, o1.data(), nout, g);
        g_host_horner_runs = false;
        HK<E>::horner_linear(res.data(), nres, a.data(), na, o2.data(), nout, g);
        if (std::memcmp(o1.data(), o2.data(), sizeof(double) * 2 * nout) != 0) {
            std::printf("MISMATCH trial %d nd=%d w=%d cls=%d coeff_scalar=%d\n", trial, nd, g.w, cls, (int)coeff_scalar);
            return 1;
        }
        ++cases;
        (void)run_steps_before;
    }
    std::printf("host_horner ok: %zu cases\n", cases);
    return 0;
}
