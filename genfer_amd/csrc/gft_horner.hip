// The Horner kernels of subst_var with a linear substitution (round 6: split out of gft_kernels.hip): one fused step
// (k_horner_linear), the whole loop per line through LDS (k_horner_linear_loop), the wave pipelines (k_horner_linear_pipe,
// k_horner_pipe_point and its batch form), and the witnesses of the speculative loop.  DESIGN 3.6, 5.1.
#include "gft_kernels.hpp"
#include "gft_kernels_int.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace gft {

template <class E>
__global__ void __launch_bounds__(256) k_horner_linear(const double* __restrict__ res, size_t rp, const double* __restrict__ a,
                                                       size_t ap, double* __restrict__ out, size_t op, HornerArgs g,
                                                       size_t total) {
    typedef typename E::V V;
    if (g.guard && *g.guard != 0u) return;  // the scan queued before this launch found the accumulator linear
    const V cv = E::from(g.c), mv = E::from(g.m);
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        size_t r = lin, roff = 0, aoff = g.a_base;
        unsigned kw = 0;
        bool in_p = true, in_r = true, in_c = true;
#pragma unroll 1
        for (int ax = g.out.nd - 1; ax >= 0; --ax) {
            unsigned d = g.out.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            if (k >= g.sh[ax]) in_p = false;
            if (k >= g.rs[ax]) in_r = false;
            if (k >= g.oc[ax]) in_c = false;
            if (ax == g.w) kw = k;
            roff += (size_t)k * g.rstr[ax];
            aoff += (size_t)k * g.astr[ax];
        }
        V p = E::zero();
        if (in_p) {
            // A = mul_var(res, m, w): res[k - e_w] * m inside the shifted source box, zero elsewhere
            bool in_src = kw >= 1 && kw - 1 < g.upper;
            if (in_src) {  // the other axes of the source box are res's own extents == sh's
                V A = E::mulw(E::ld(res, rp, roff - g.rstr[g.w]), mv);
                p = A;
            }
            if (!g.c_zero) {
                p = E::add0(p);
                if (in_r) {
                    V x = E::ld(res, rp, roff);
                    V B = g.c_one ? x : E::mulw(cv, x);
                    p = E::addw(p, B);
                }
            }
        }
        V v;
        if (g.coeff_scalar) {
            v = p;
            if (lin == 0) v = E::add(p, E::ld(a, ap, g.a_base));
        } else {
            v = E::zero();
            if (in_p) v = E::add0(p);
            if (in_c) v = E::addw(v, E::ld(a, ap, aoff));
        }
        E::st(out, op, lin, v);
    }
}
template <class E>
void K<E>::horner_linear(hipStream_t st, const double* res, size_t res_plane, const double* a, size_t a_plane, double* out,
                         size_t out_plane, const HornerArgs& args) {
    size_t total = 1;
    for (int i = 0; i < args.out.nd; ++i) total *= args.out.d[i];
    if (total == 0) return;
    GFT_LAUNCH(k_horner_linear<E>, dim3(grid_for(total)), dim3(256), 0, st, res, res_plane, a, a_plane, out, out_plane,
                       args, total);
}

// value of lane l - 1 in lane l (lane 0: zero) — DPP wave_shr:1, no LDS
__device__ inline double wave_shr1_d(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <class E>
__device__ inline typename E::V wave_shr1_any(typename E::V v);
template <>
__device__ inline double wave_shr1_any<EF64>(double v) { return wave_shr1_d(v); }
template <>
__device__ inline Iv wave_shr1_any<EIv>(Iv v) { return Iv{wave_shr1_d(v.lo), wave_shr1_d(v.hi)}; }

// LDS rings between the waves of a workgroup (the Horner pipelines below): one slot per step, written once by the wave
// below, read once by the wave above.  Slots start out EMPTY (a quiet-NaN payload no arithmetic produces); the reader
// REQUESTS a slot a step ahead — a relaxed workgroup-scope atomic load: a plain ds_read the compiler may neither hoist
// nor merge and whose result register it tracks like any other load's (round 3 first wrote these as inline-asm ds_read
// with a later s_waitcnt: the compiler, thinking the register defined at the asm statement, was free to copy it before
// the data arrived — wrong bounds in a fraction of the runs) — and looks at it a step later; an EMPTY slot is polled.
// Should a genuine value ever carry the EMPTY pattern (an input NaN with exactly that payload) the reader accepts it
// after 2^24 polls (ring_receive: the bound counts co-resident polls, so the writer is long done by then): slow, never wrong.
constexpr unsigned long long RING_EMPTY = 0x7ff8dead5a5a0badull;
__device__ inline unsigned long long ring_load_bits(const double* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline void ring_store_bits(double* p, unsigned long long b) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
struct RingWord {  // a requested slot: the raw bits of its planes
    unsigned long long lo, hi;
};
template <class E>
__device__ inline RingWord ring_request(const double* slot, size_t plane) {
    RingWord w;
    w.lo = ring_load_bits(slot);
    w.hi = E::W == 2 ? ring_load_bits(slot + plane) : 0ull;
    return w;
}
template <class E>
__device__ inline bool ring_ready(const RingWord& w) { return w.lo != RING_EMPTY && (E::W == 1 || w.hi != RING_EMPTY); }
template <class E>
__device__ inline typename E::V ring_value(const RingWord& w);
template <>
__device__ inline double ring_value<EF64>(const RingWord& w) { return bits_f64((long long)w.lo); }
template <>
__device__ inline Iv ring_value<EIv>(const RingWord& w) { return Iv{bits_f64((long long)w.lo), bits_f64((long long)w.hi)}; }
// the value of `slot`, requested earlier as `w`: polls while it is still EMPTY
template <class E>
__device__ inline typename E::V ring_receive(const double* slot, size_t plane, RingWord w) {
    // The bound counts POLLS OF THIS WAVE, not wall-clock time: producer and consumer are waves of one workgroup, which is
    // dispatched, preempted and resumed as a whole, so the count only advances while the producer is resident too — a
    // pre-empted / context-saved queue stops both.  2^24 polls (seconds of co-resident time; the producer needs one step,
    // ~1 us) are only exhausted if a genuine value carries the EMPTY pattern: it is then accepted, late, never wrong.
    for (unsigned spins = 0; !ring_ready<E>(w) && spins < (1u << 24); ++spins) {
        if (spins < 4096) __builtin_amdgcn_s_sleep(1);
        else __builtin_amdgcn_s_sleep(4);
        w = ring_request<E>(slot, plane);
    }
    return ring_value<E>(w);
}
__device__ inline void ring_put(double* slot, size_t, double v) { ring_store_bits(slot, (unsigned long long)f64_bits(v)); }
__device__ inline void ring_put(double* slot, size_t plane, Iv v) {
    ring_store_bits(slot, (unsigned long long)f64_bits(v.lo));
    ring_store_bits(slot + plane, (unsigned long long)f64_bits(v.hi));
}

// One element of one Horner step with a linear substitution (HornerArgs' operation sequence), shared by the in-kernel
// loops below so that they produce the same bits:  xm1 = res[k - e_w] (valid iff t1), x = res[k] (valid iff t2),
// coef = the step's coefficient at k (valid iff t3); in_p: k lies inside P = res * s.  Called by every lane that is
// inside the step's output box (the wave-level ballots inside cover exactly those lanes).
template <class E>
struct HornerConsts {
    typename E::V cv, mv;
    bool c_zero, c_one, coeff_scalar, pos_consts, semi_consts;
};
template <class E>
__device__ __forceinline__ HornerConsts<E> horner_consts(const HornerLoopArgs& g) {
    HornerConsts<E> h;
    h.cv = E::from(g.c);
    h.mv = E::from(g.m);
    h.c_zero = g.c_zero != 0;
    h.c_one = g.c_one != 0;
    h.coeff_scalar = g.coeff_scalar != 0;
    h.pos_consts = false;
    h.semi_consts = false;
    if constexpr (E::HAS_POS) {
        h.pos_consts = E::pos_ok(h.mv) && (h.c_zero || h.c_one || E::pos_ok(h.cv));
        h.semi_consts = !(g.diag & 32) && !h.pos_consts && E::pos_ok(h.mv) &&
                        (h.c_zero || h.c_one || (E::is_finite(h.cv) && h.cv.lo <= h.cv.hi && !E::maybe_special(h.cv)));
    }
    return h;
}
template <class E>
__device__ __forceinline__ typename E::V horner_elem(const HornerConsts<E>& h, bool in_p, bool t1, bool t2, bool t3,
                                                     typename E::V xm1, typename E::V x, typename E::V coef) {
    typedef typename E::V V;
    if constexpr (E::HAS_POS) {
        // Positive regime (gft_elem.hpp): probability-like accumulators, coefficients and substitution — every
        // operand of this step a proper positive interval.  The step is then ~20 instructions instead of ~130
        // (no sign cases, no short-circuit selects, next_down / next_up as integer steps), the same operations
        // on the same values; whether the regime held is settled per wave, before (operands) and after
        // (no underflow to zero, no overflow) — otherwise the general code below computes the step.
        if (h.pos_consts) {
            const bool ok = (!t1 || E::pos_ok(xm1)) && (!t2 || E::pos_ok(x)) && (!t3 || E::pos_ok(coef));
            if (!any_lane(!ok)) {
                const V p1 = E::mul_pos(xm1, h.mv);
                const V p2 = h.c_one ? x : E::mul_pos(h.cv, x);
                bool bad = (t1 && !E::pos_first_ok(p1)) || (t2 && !h.c_one && !E::pos_first_ok(p2));
                const V p12 = E::add_pos(p1, p2);
                V p = t1 ? (t2 ? p12 : p1) : (t2 ? p2 : E::zero());
                const bool has_p = t1 || t2;
                const V pc = E::add_pos(p, coef);
                V v = t3 ? (has_p ? pc : coef) : p;
                if (!h.coeff_scalar && !in_p) v = t3 ? coef : E::zero();
                bad = bad || ((has_p || t3) && !E::pos_result_ok(v));
                if (!any_lane(bad)) return v;
            }
        }
        // SEMI-positive regime: accumulator and coefficients positive, m positive, the constant c any finite
        // interval that is not one of the points 0 / +-1 — what the `--bounds` programs bring (their
        // c = subst - constant_term(subst) is a few ulps around zero).  With x > 0 the reference's min / max
        // of the four products of c * x (iv:164-190) are known from the signs of c's bounds, and for
        // positive operands they are lo * lo and hi * hi; the outward steps stay the general next_down /
        // next_up, so every intermediate is exactly the reference's whatever under- or overflows, and no
        // operation can short-circuit (no operand or partial result is [0,0] or a +-1 point: widened
        // intervals are never points).  ~60 instead of ~120 instructions per element, no fallback needed.
        if (h.semi_consts) {
            const bool ok = (!t1 || E::pos_ok(xm1)) && (!t2 || E::pos_ok(x)) && (!t3 || E::pos_ok(coef));
            if (!any_lane(!ok)) {
                const V p1 = E::widen(xm1.lo * h.mv.lo, xm1.hi * h.mv.hi);
                V p2 = x;
                if (!h.c_one) p2 = E::widen(h.cv.lo * (h.cv.lo >= 0.0 ? x.lo : x.hi), h.cv.hi * (h.cv.hi >= 0.0 ? x.hi : x.lo));
                const V p12 = E::widen(p1.lo + p2.lo, p1.hi + p2.hi);
                const V p = t1 ? (t2 ? p12 : p1) : (t2 ? p2 : E::zero());
                const bool has_p = t1 || t2;
                const V pc = E::widen(p.lo + coef.lo, p.hi + coef.hi);
                V v = t3 ? (has_p ? pc : coef) : p;
                if (!h.coeff_scalar && !in_p) v = t3 ? coef : E::zero();
                return v;
            }
        }
    }
    V p = E::zero();
    if (in_p) {
        if (t1) p = E::mulw(xm1, h.mv);
        if (!h.c_zero) {
            p = E::add0(p);
            if (t2) p = E::addw(p, h.c_one ? x : E::mulw(h.cv, x));
        }
    }
    V v;
    if (h.coeff_scalar) {
        v = p;
        if (t3) v = E::add(p, coef);
    } else {
        v = E::zero();
        if (in_p) v = E::add0(p);
        if (t3) v = E::addw(v, coef);
    }
    return v;
}

// The Horner recursion out[k] = f(res[k], res[k - e_w], coeff[k]) couples positions along the substitution axis w
// only, so every LINE along w (one position on all other axes) runs ALL steps on its own: one workgroup per line,
// no synchronisation between workgroups, intermediates ping-pong in LDS, the next step's coefficient is prefetched
// while the current one is computed, the last step writes to global memory.  A step only changes which positions
// along w are inside the current boxes (off the axis the boxes are rs0 at step 0 and the final extents afterwards).
constexpr int HL_EPT_MAX = 2;  // positions along w per thread: lines up to 2048 long
// HL_EPT positions along w per thread; coefficient prefetch depth HL_PF steps (registers: HL_PF * HL_EPT elements)
template <class E, int HL_EPT, int HL_PF>
__global__ void __launch_bounds__(1024) k_horner_linear_loop(const double* __restrict__ res0, size_t rp0,
                                                             const double* __restrict__ a, size_t ap,
                                                             double* __restrict__ out, size_t plane, HornerLoopArgs g,
                                                             unsigned* __restrict__ wit) {
    typedef typename E::V V;
    extern __shared__ double hl_lds[];  // [buffer][plane][lw_pad]
    if (g.guard && *g.guard != 0u) return;  // the scan queued before this launch found the accumulator linear: the speculation failed
    const unsigned lw = g.fs[g.w], lw_pad = g.lw_pad;
    // this block's line: position on the axes other than w
    size_t foff_b = 0, aoff_b = 0, roff0_b = 0;
    bool off_p0 = true, off_o0 = true, in_c_b = true;  // step 0 / coefficient box, axes other than w
    // Non-linearity witnesses (K<E>::witness): on this line a non-zero at position kw is one if the index has two
    // non-zero coordinates or a coordinate >= 2 — from position wit_from on, given the line's other coordinates.
    unsigned wit_from = 2;
    {
        size_t r = blockIdx.x;
        unsigned nz_coords = 0;
        bool big = false;
#pragma unroll
        for (int ax = MAXD - 1; ax >= 0; --ax) {
            if (ax < g.nd && ax != g.w) {
                unsigned d = g.fs[ax];
                unsigned k = (unsigned)(r % d);
                r /= d;
                foff_b += (size_t)k * g.fstr[ax];
                aoff_b += (size_t)k * g.astr[ax];
                roff0_b += (size_t)k * g.rstr0[ax];
                unsigned r0 = g.rs0[ax], o0 = (!g.coeff_scalar && g.oc[ax] > r0) ? g.oc[ax] : r0;
                if (k >= r0) off_p0 = false;
                if (k >= o0) off_o0 = false;
                if (k >= g.oc[ax]) in_c_b = false;
                if (k) nz_coords++;
                if (k >= 2) big = true;
            }
        }
        if (big || nz_coords >= 2) wit_from = 0;
        else if (nz_coords == 1) wit_from = 1;
    }
    const size_t wstr_f = g.fstr[g.w], wstr_0 = g.rstr0[g.w], wstr_a = g.astr[g.w];
    const unsigned degw = g.deg[g.w], ocw = g.coeff_scalar ? 0u : g.oc[g.w];
    unsigned kw[HL_EPT];
    bool have[HL_EPT], takes_c[HL_EPT];
    // Coefficient slabs are read HL_PF steps ahead into a register ring: a step is a few hundred cycles of arithmetic,
    // a global load ~2000 under this kernel's occupancy (one workgroup per line) — one step of lead left every step
    // waiting for its coefficient (1.4 us per step on mixture --bounds, 2.5 of its 3.6 s).
    V ring[HL_PF][HL_EPT];
#pragma unroll
    for (int e = 0; e < HL_EPT; ++e) {
        kw[e] = threadIdx.x + e * blockDim.x;
        have[e] = kw[e] < lw;
        takes_c[e] = have[e] && (g.coeff_scalar ? (blockIdx.x == 0 && kw[e] == 0) : (in_c_b && kw[e] < g.oc[g.w]));
    }
    // The ring's loads are UNCONDITIONAL, on clamped addresses (a position outside the coefficient box reads the
    // slab's first element, a step beyond the last reads the last step's slab), and the box is applied when the value is
    // used: a conditional load is a phi of {zero, loaded}, which hipcc materialises in a temporary and copies into the
    // ring's register at the end of the step — after waiting for the load it issued a moment ago.
    size_t c_off[HL_EPT];
#pragma unroll
    for (int e = 0; e < HL_EPT; ++e) c_off[e] = takes_c[e] ? aoff_b + (size_t)kw[e] * wstr_a : 0;
    const unsigned last_step = g.nsteps - 1;
#pragma unroll
    for (int d = 0; d < HL_PF; ++d)
#pragma unroll
        for (int e = 0; e < HL_EPT; ++e)
            ring[d][e] = E::ld(a, ap, (size_t)(g.first_i - ((unsigned)d < last_step ? (unsigned)d : last_step)) * g.a_vstride + c_off[e]);
    const HornerConsts<E> hc = horner_consts<E>(g);
    unsigned rsw = g.rs0[g.w];
    for (unsigned t0 = 0; t0 < g.nsteps; t0 += HL_PF) {
#pragma unroll
        for (int d = 0; d < HL_PF; ++d) {
            const unsigned t = t0 + (unsigned)d;
            // ring slot d: hand over step t's coefficient and request step t + HL_PF's — outside the `t < nsteps` test, so
            // that the slot's registers have ONE definition per unrolled copy (no phi, no end-of-step copy that would wait
            // for the load just issued); the clamped address makes the surplus loads harmless
            V coef[HL_EPT];
            {
                const unsigned tn = t + HL_PF;
                const size_t a_base = (size_t)(g.first_i - (tn < last_step ? tn : last_step)) * g.a_vstride;
#pragma unroll
                for (int e = 0; e < HL_EPT; ++e) {
                    coef[e] = takes_c[e] ? ring[d][e] : E::zero();
                    ring[d][e] = E::ld(a, ap, a_base + c_off[e]);
                }
            }
            if (t >= g.nsteps) continue;
            const unsigned shw = rsw + 1 < degw ? rsw + 1 : degw;
            const unsigned upper = shw - 1 < rsw ? shw - 1 : rsw;
            const unsigned osw = ocw > shw ? ocw : shw;
            const double* src_l = hl_lds + (size_t)((t + 1) & 1u) * E::W * lw_pad;  // written by step t-1
            double* dst_l = hl_lds + (size_t)(t & 1u) * E::W * lw_pad;
            const bool last = t + 1 == g.nsteps, first = t == 0;
            int witness = 0;
#pragma unroll
            for (int e = 0; e < HL_EPT; ++e) {
                if (!have[e]) continue;
                const bool in_o = (first ? off_o0 : true) && kw[e] < osw;
                if (!in_o) continue;
                const bool in_p = (first ? off_p0 : true) && kw[e] < shw;
                const bool in_r = (first ? off_p0 : true) && kw[e] < rsw;
                const bool t1 = in_p && kw[e] >= 1 && kw[e] - 1 < upper;  // res[k - 1] * m
                const bool t2 = in_p && !g.c_zero && in_r;               // c * res[k]
                V xm1 = E::one(), x = E::one();
                if (t1) xm1 = first ? E::ld(res0, rp0, roff0_b + (size_t)(kw[e] - 1) * wstr_0) : E::ld(src_l, lw_pad, kw[e] - 1);
                if (t2) x = first ? E::ld(res0, rp0, roff0_b + (size_t)kw[e] * wstr_0) : E::ld(src_l, lw_pad, kw[e]);
                const V v = horner_elem<E>(hc, in_p, t1, t2, takes_c[e], xm1, x, coef[e]);
                if (last) E::st(out, plane, foff_b + (size_t)kw[e] * wstr_f, v);
                else E::st(dst_l, lw_pad, kw[e], v);
                if (kw[e] >= wit_from && !E::is_zero(v)) witness = 1;
            }
            rsw = osw;
            // the witness word is raised per WAVE (ballot + one lane), not through a block-wide OR: __syncthreads_or is a
            // shared-memory reduction with several barriers of its own, paid on every step of this latency chain
            // (a plain store, no load-and-test first: a load would have to be waited for, and with it the whole prefetch ring)
            if (wit && !last && any_lane(witness != 0) && (threadIdx.x & 63u) == 0)
                wit_raise(&wit[t]);
            if (!(g.diag & 2)) lds_barrier();  // the line passes from step to step through LDS; global loads (the ring) stay in flight
        }
    }
}
// The same loop as a WAVE PIPELINE (round 3).  The kernel above passes the line from step to step through LDS and a
// workgroup barrier: ~0.7 us per interval step on a lone workgroup, all of it latency (LDS round trip, barrier, the
// dependent arithmetic at one wave per SIMD).  But position k of step t depends on positions k - 1 and k of step t - 1
// only — data flows UP the line and never down.  So: one position per lane, the line's values stay in REGISTERS from
// step to step, res[k - 1] arrives by a DPP wave shift, and the only value that crosses a wave boundary — the last
// position of wave b, needed by lane 0 of wave b + 1 one step later — is published in an LDS ring slot per step with a
// step counter.  Nothing flows back, so wave b never waits for wave b + 1: the waves of a line run as a pipeline,
// wave b + 1 a step or two behind wave b, with no barrier anywhere; the consumer requests its boundary value one step
// ahead (counter and value are read in order, the producer writes them in order, both through the CU's in-order LDS
// queue), so in steady state nothing on the step's dependent chain touches memory.  A line of 180 intervals is three
// waves on three SIMDs; the time per step is one wave's arithmetic chain.  Same horner_elem per element => same bits.
// POINT: the coefficient box is a single position along w (always when the substituted variable is w itself — `--bounds`
// runs of v -> c + m*v — or when the coefficient slab is one element): only position 0 of the line takes a coefficient,
// one per step.  All of them are fetched into LDS before the pipeline starts and wave 0 requests step t + 1's during
// step t, so no step of the loop touches global memory (a register ring of global loads, the general case, leaves the
// compiler's conservative vmcnt waits on the dependent chain of every few steps).
template <class E, int HL_PF, bool POINT>
__global__ void __launch_bounds__(1024) k_horner_linear_pipe(const double* __restrict__ res0, size_t rp0,
                                                             const double* __restrict__ a, size_t ap,
                                                             double* __restrict__ out, size_t plane, HornerLoopArgs g,
                                                             unsigned* __restrict__ wit) {
    typedef typename E::V V;
    extern __shared__ double hp_lds[];  // [boundary b][plane][nsteps] ring (one slot per step), then the counters
    if (g.guard && *g.guard != 0u) return;  // the scan queued before this launch found the accumulator linear: the speculation failed
    const unsigned lw = g.fs[g.w];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
    const unsigned nslots = g.nsteps;
    const double* ring_b = hp_lds + (size_t)(wave ? wave - 1 : 0) * E::W * nslots;  // the boundary BELOW this wave
    double* ring_a = hp_lds + (size_t)wave * E::W * nslots;                          // the boundary ABOVE (this wave writes)
    double* coef_l = hp_lds + (size_t)(nw - 1) * E::W * nslots;  // POINT: [plane][nsteps] coefficients of this line
    for (size_t i = threadIdx.x; i < (size_t)(nw - 1) * E::W * nslots; i += blockDim.x) ring_store_bits(hp_lds + i, RING_EMPTY);
    size_t foff_b = 0, aoff_b = 0, roff0_b = 0;
    bool off_p0 = true, off_o0 = true, in_c_b = true;
    unsigned wit_from = 2;
    {
        size_t r = blockIdx.x;
        unsigned nz_coords = 0;
        bool big = false;
#pragma unroll
        for (int ax = MAXD - 1; ax >= 0; --ax) {
            if (ax < g.nd && ax != g.w) {
                unsigned d = g.fs[ax];
                unsigned k = (unsigned)(r % d);
                r /= d;
                foff_b += (size_t)k * g.fstr[ax];
                aoff_b += (size_t)k * g.astr[ax];
                roff0_b += (size_t)k * g.rstr0[ax];
                unsigned r0 = g.rs0[ax], o0 = (!g.coeff_scalar && g.oc[ax] > r0) ? g.oc[ax] : r0;
                if (k >= r0) off_p0 = false;
                if (k >= o0) off_o0 = false;
                if (k >= g.oc[ax]) in_c_b = false;
                if (k) nz_coords++;
                if (k >= 2) big = true;
            }
        }
        if (big || nz_coords >= 2) wit_from = 0;
        else if (nz_coords == 1) wit_from = 1;
    }
    const size_t wstr_f = g.fstr[g.w], wstr_0 = g.rstr0[g.w], wstr_a = g.astr[g.w];
    const unsigned degw = g.deg[g.w], ocw = g.coeff_scalar ? 0u : g.oc[g.w];
    const unsigned kw = threadIdx.x;
    const bool have = kw < lw;
    const bool takes_c = have && (g.coeff_scalar ? (blockIdx.x == 0 && kw == 0) : (in_c_b && kw < g.oc[g.w]));
    const size_t c_off = takes_c ? aoff_b + (size_t)kw * wstr_a : 0;
    const unsigned last_step = g.nsteps - 1;
    V ring[HL_PF];  // coefficient slabs HL_PF steps ahead, unconditional clamped loads (see k_horner_linear_loop)
    if constexpr (POINT) {
        const bool line_takes = g.coeff_scalar ? blockIdx.x == 0 : in_c_b;  // position 0 of this line has coefficients at all
        for (unsigned i = threadIdx.x; i < g.nsteps; i += blockDim.x)
            E::st(coef_l, nslots, i, line_takes ? E::ld(a, ap, (size_t)(g.first_i - i) * g.a_vstride + aoff_b) : E::zero());
    } else {
#pragma unroll
        for (int d = 0; d < HL_PF; ++d) ring[d] = E::ld(a, ap, (size_t)(g.first_i - ((unsigned)d < last_step ? (unsigned)d : last_step)) * g.a_vstride + c_off);
    }
    __syncthreads();  // the only barrier: counters zeroed, coefficients staged — then the pipeline runs free
    V c_cur = E::zero(), c_nxt = E::zero();
    if (POINT && wave == 0) c_cur = E::ld(coef_l, nslots, 0);
    const HornerConsts<E> hc = horner_consts<E>(g);
    unsigned rsw = g.rs0[g.w];
    // The incoming accumulator enters the registers here, so that no step has a global load on its path (inside the
    // unrolled loop a `first ? global : register` operand makes every HL_PF-th step wait for ALL outstanding loads —
    // the coefficient ring's included).  Step 0 reads res0[k] / res0[k - 1] only inside res0's box, which is what the
    // guards below load; everything else is never looked at.
    V cur = E::zero();     // this position's value after the previous step
    V below = E::zero();   // lane 0, wave > 0: the value of position kw - 1 after the previous step (from the ring)
    if (off_p0 && kw < rsw) cur = E::ld(res0, rp0, roff0_b + (size_t)kw * wstr_0);
    if (lane == 0 && wave && off_p0 && kw - 1 < rsw) below = E::ld(res0, rp0, roff0_b + (size_t)(kw - 1) * wstr_0);
    RingWord bw{0, 0};     // the requested ring slot: the boundary value for the NEXT step
    asm volatile("; loop-invariant scalars are in their registers" : : "s"(rsw), "s"(degw), "s"(ocw), "s"(nslots));
    for (unsigned t0 = 0; t0 < g.nsteps; t0 += HL_PF) {
#pragma unroll
        for (int d = 0; d < HL_PF; ++d) {
            const unsigned t = t0 + (unsigned)d;
            V coef;
            if constexpr (!POINT) {
                const unsigned tn = t + HL_PF;
                const size_t a_base = (size_t)(g.first_i - (tn < last_step ? tn : last_step)) * g.a_vstride;
                coef = takes_c ? ring[d] : E::zero();
                ring[d] = E::ld(a, ap, a_base + c_off);
            }
            if (t >= g.nsteps) continue;
            const unsigned shw = rsw + 1 < degw ? rsw + 1 : degw;
            const unsigned upper = shw - 1 < rsw ? shw - 1 : rsw;
            const unsigned osw = ocw > shw ? ocw : shw;
            const bool last = t + 1 == g.nsteps, first = t == 0;
            if constexpr (POINT) {
                coef = takes_c ? c_cur : E::zero();
                if (wave == 0 && !last) c_nxt = E::ld(coef_l, nslots, t + 1);  // consumed after this step
            }
            // request the boundary value of THIS step's output from the wave below (needed at step t + 1): counter, then
            // value, in order; consumed after this step's arithmetic
            if (wave && !last) bw = ring_request<E>(ring_b + t, nslots);
            // res[k - 1] after the previous step: the neighbouring lane, or the ring for lane 0
            V shifted = wave_shr1_any<E>(cur);
            if (lane == 0) shifted = below;
            int witness = 0;
            const bool in_o = have && (first ? off_o0 : true) && kw < osw;
            if (in_o) {
                const bool in_p = (first ? off_p0 : true) && kw < shw;
                const bool in_r = (first ? off_p0 : true) && kw < rsw;
                const bool t1 = in_p && kw >= 1 && kw - 1 < upper;
                const bool t2 = in_p && !g.c_zero && in_r;
                V xm1 = E::one(), x = E::one();
                if (t1) xm1 = shifted;
                if (t2) x = cur;
                const V v = horner_elem<E>(hc, in_p, t1, t2, takes_c, xm1, x, coef);
                if (last) E::st(out, plane, foff_b + (size_t)kw * wstr_f, v);
                cur = v;
                if (kw >= wit_from && !E::is_zero(v)) witness = 1;
            }
            rsw = osw;
            if (last) continue;
            // publish this wave's last position for the wave above: value, then counter (in-order LDS queue)
            if (lane == 63 && wave + 1 < nw) ring_put(ring_a + t, nslots, cur);
            if (wit && any_lane(witness != 0) && lane == 0) wit_raise(&wit[t]);
            if constexpr (POINT) {
                if (wave == 0) c_cur = c_nxt;
            }
            // the slot requested at the top of the step (polled if the wave below had not published it yet)
            if (wave) below = ring_receive<E>(ring_b + t, nslots, bw);
        }
    }
}

// ---- the POINT pipeline, lean ------------------------------------------------------------------------------------------
// A lone wave issues roughly one instruction every five cycles, whatever the instruction: the time of a step IS its
// instruction count (measured: the generic step above is ~330 instructions for intervals, ~130 for f64 — flag
// arithmetic, exec-mask branches, three regimes — and costs 0.78 / 0.36 us a step; profiles/r03/horner_loop.txt).  This
// kernel is the same pipeline with a step written for instruction count, for the case the `--bounds` programs live in:
// coefficients at position 0 only (POINT), substitution constant c neither 0 nor 1, steps after the first.
//   * the boxes of a step follow from ONE scalar: A = extent of the accumulator along w, S = min(A + 1, deg);
//     position k multiplies res[k - 1] by m iff k - 1 < A and k < S (t1), adds c * res[k] iff k < A (t2);
//   * straight-line code, selects instead of branches; lanes outside the box compute garbage nobody reads; the wave index
//     is a scalar, the publishing lane writes through a per-lane address (everyone else into a dummy area), witnesses
//     are collected in a scalar bit mask and stored once per 64 steps;
//   * intervals: every operand of a lean step is a positive finite interval (own outputs are tested when they are
//     produced, the boundary value when it arrives, the line's coefficients when they are staged), m positive, c a
//     finite interval with non-zero bounds.  Then the reference's products and sums (iv:126-190) take no short-circuit
//     and the SIGN of every bound is known before it is computed — x * m and (q + c x) positive, c.lo * x and c.hi * x
//     with the signs of c's bounds — so each outward step next_down / next_up (f64.rs:127-171) is the integer step
//     `bits -/+ 1` in the direction that sign dictates: 2 instructions instead of 9.  Where an assumption fails (a
//     product underflows to zero, a sum is not positive, a bound reaches infinity) the integer step produces a NaN
//     pattern or a non-positive bound, which survives to the step's output: ONE test of the output (lo > 0, hi < inf)
//     validates the whole step, and a wave whose test fails recomputes that step with horner_elem from the operands it
//     still holds.  Same operations on the same values => same bits (tests: GFT_HORNER_LEAN=0 / GFT_HORNER_PIPE=0 A/B
//     against the oracle);
//   * `p1 + p2` (inner positions) and `coef + p2` (position 0) are one addition with a selected first operand (IEEE
//     addition commutes bit for bit).
template <class E>
struct LeanConsts;
template <>
struct LeanConsts<EF64> {
    double c, m;
    bool scalar_coef;
    __device__ explicit LeanConsts(const HornerConsts<EF64>& h) : c(h.cv), m(h.mv), scalar_coef(h.coeff_scalar) {}
    __device__ static bool usable(const HornerConsts<EF64>& h) { return !h.c_zero && !h.c_one; }
    __device__ static bool operand_ok(double) { return true; }
    __device__ static bool result_ok(double) { return true; }
    static constexpr bool CHECKED = false;
    // t1 / t2 / t3 as in horner_elem; unused operands may hold anything
    __device__ __forceinline__ double step(bool t1, bool t2, bool t3, double xm1, double x, double coef) const {
        double p = t1 ? xm1 * m : 0.0;
        p = 0.0 + p;
        const double p2 = p + c * x;
        p = t2 ? p2 : p;
        const double v = scalar_coef ? p : 0.0 + p;
        const double vc = v + coef;
        return t3 ? vc : v;
    }
};
template <>
struct LeanConsts<EIv> {
    Iv c, m;
    bool lo_uses_hi, hi_uses_lo;   // c.lo < 0: c.lo * x is smallest at x.hi; c.hi < 0: c.hi * x is largest at x.lo
    long long dlo, dhi;            // integer steps of next_down(c.lo * x) / next_up(c.hi * x): by the sign of the product
    __device__ explicit LeanConsts(const HornerConsts<EIv>& h) : c(h.cv), m(h.mv) {
        lo_uses_hi = !(c.lo >= 0.0);
        hi_uses_lo = !(c.hi >= 0.0);
        dlo = c.lo < 0.0 ? 1 : -1;   // a negative bound moves away from zero, a positive one towards it
        dhi = c.hi > 0.0 ? 1 : -1;
    }
    // m a positive interval; c finite, ordered, no 0 / +-1 point and no zero bound (the signs of c.lo * x and c.hi * x
    // must be known): the constants of horner_elem's positive AND semi-positive regimes
    __device__ static bool usable(const HornerConsts<EIv>& h) {
        return !h.c_zero && !h.c_one && EIv::pos_ok(h.mv) && EIv::is_finite(h.cv) && h.cv.lo <= h.cv.hi && !EIv::maybe_special(h.cv) &&
               h.cv.lo != 0.0 && h.cv.hi != 0.0;
    }
    __device__ static bool operand_ok(Iv v) { return EIv::pos_ok(v); }
    // a lean output: widened (never a point), so positive and finite is all there is to test; NaN fails both compares
    __device__ static bool result_ok(Iv v) { return v.lo > 0.0 && v.hi < bits_f64(0x7ff0000000000000LL); }
    static constexpr bool CHECKED = true;
    __device__ __forceinline__ Iv step(bool t1, bool t2, bool t3, Iv xm1, Iv x, Iv coef) const {
        const Iv p1 = EIv::mul_pos(xm1, m);
        Iv p2;
        p2.lo = bits_f64(f64_bits(c.lo * (lo_uses_hi ? x.hi : x.lo)) + dlo);
        p2.hi = bits_f64(f64_bits(c.hi * (hi_uses_lo ? x.lo : x.hi)) + dhi);
        Iv q;
        q.lo = t1 ? p1.lo : coef.lo;
        q.hi = t1 ? p1.hi : coef.hi;
        const Iv s = EIv::add_pos(q, p2);
        const bool both = t2 && (t1 || t3);
        Iv v;  // t2 only: p2; t1 only: p1 (= q)
        v.lo = both ? s.lo : (t2 ? p2.lo : q.lo);
        v.hi = both ? s.hi : (t2 ? p2.hi : q.hi);
        return v;
    }
};

// One line of the loop: `line` = the line's index among the loop's lines, `nw` = the waves the line needs (the workgroup may
// have more when two loops share a launch: the surplus waves only attend the one barrier).
template <class E>
__device__ __forceinline__ void horner_pipe_point_line(const double* __restrict__ res0, size_t rp0, const double* __restrict__ a, size_t ap,
                                                       double* __restrict__ out, size_t plane, const HornerLoopArgs& g,
                                                       unsigned* __restrict__ wit, const unsigned line, const unsigned nw, double* hp_lds) {
    typedef typename E::V V;
    typedef LeanConsts<E> LC;
    // hp_lds: [boundary b][plane][nsteps] rings, [plane][nsteps] coefficients, counters, dummy area
    if (g.guard && *g.guard != 0u) return;  // the scan queued before this launch found the accumulator linear: the speculation failed
    const unsigned lw = g.fs[g.w];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const unsigned nthreads = nw * 64u;
    if (wave >= nw) {  // (a surplus wave of a shared launch)
        (void)__syncthreads_and(1);
        return;
    }
    const unsigned nslots = g.nsteps;
    const double* ring_b = hp_lds + (size_t)(wave ? wave - 1 : 0) * E::W * nslots;
    double* ring_a = hp_lds + (size_t)wave * E::W * nslots;
    double* coef_l = hp_lds + (size_t)(nw - 1) * E::W * nslots;
    double* dummy = coef_l + (size_t)E::W * nslots;  // 64 x value planes nobody reads (where the non-publishing lanes write)
    for (size_t i = threadIdx.x; i < (size_t)(nw - 1) * E::W * nslots; i += nthreads) ring_store_bits(hp_lds + i, RING_EMPTY);
    size_t foff_b = 0, aoff_b = 0, roff0_b = 0;
    bool off_p0 = true, off_o0 = true, in_c_b = true;
    unsigned wit_from = 2;
    {
        size_t r = line;
        unsigned nz_coords = 0;
        bool big = false;
#pragma unroll
        for (int ax = MAXD - 1; ax >= 0; --ax) {
            if (ax < g.nd && ax != g.w) {
                unsigned d = g.fs[ax];
                unsigned k = (unsigned)(r % d);
                r /= d;
                foff_b += (size_t)k * g.fstr[ax];
                aoff_b += (size_t)k * g.astr[ax];
                roff0_b += (size_t)k * g.rstr0[ax];
                unsigned r0 = g.rs0[ax], o0 = (!g.coeff_scalar && g.oc[ax] > r0) ? g.oc[ax] : r0;
                if (k >= r0) off_p0 = false;
                if (k >= o0) off_o0 = false;
                if (k >= g.oc[ax]) in_c_b = false;
                if (k) nz_coords++;
                if (k >= 2) big = true;
            }
        }
        if (big || nz_coords >= 2) wit_from = 0;
        else if (nz_coords == 1) wit_from = 1;
    }
    const size_t wstr_f = g.fstr[g.w], wstr_0 = g.rstr0[g.w];
    const unsigned degw = g.deg[g.w], ocw = g.coeff_scalar ? 0u : g.oc[g.w];
    const unsigned kw = threadIdx.x, kwm1 = kw - 1u;  // position 0: kwm1 = 2^32 - 1, below no extent
    const bool have = kw < lw;
    const bool line_takes = g.coeff_scalar ? line == 0 : in_c_b;
    const bool takes_c = have && kw == 0 && line_takes;
    // stage the line's coefficients; are they all usable as lean operands?
    int coefs_ok = 1;
    for (unsigned i = threadIdx.x; i < g.nsteps; i += nthreads) {
        const V cf = line_takes ? E::ld(a, ap, (size_t)(g.first_i - i) * g.a_vstride + aoff_b) : E::zero();
        E::st(coef_l, nslots, i, cf);
        if (line_takes && !LC::operand_ok(cf)) coefs_ok = 0;
    }
    coefs_ok = __syncthreads_and(coefs_ok);  // the only barrier: rings emptied, coefficients staged — then the pipeline runs free
    V c_cur = E::zero();
    if (wave == 0) c_cur = E::ld(coef_l, nslots, 0);
    const HornerConsts<E> hc = horner_consts<E>(g);
    const LC lc(hc);
    unsigned rsw = g.rs0[g.w];
    V cur = E::zero(), below = E::zero();
    if (off_p0 && kw < rsw) cur = E::ld(res0, rp0, roff0_b + (size_t)kw * wstr_0);
    if (lane == 0 && wave && off_p0 && kwm1 < rsw) below = E::ld(res0, rp0, roff0_b + (size_t)kwm1 * wstr_0);
    // where this lane publishes: lane 63 of a wave with a wave above it into the ring / counter, everyone else into the dummy area
    const bool publisher = lane == 63 && wave + 1 < nw;
    double* const post_v = publisher ? ring_a : dummy + lane;
    const size_t post_vplane = publisher ? nslots : 64u;
    const unsigned post_vstep = publisher ? 1u : 0u;
    // What a wave requests at the top of step t and looks at after it: wave 0 the coefficient of step t + 1, the others
    // slot t of the ring below.  ONE unconditional load site (clamped index at the last step): a request under a
    // condition is a phi of {old, loaded}, which the compiler resolves with a copy — and a wait — right at the request.
    const double* const req_base = wave ? ring_b : coef_l;
    const bool lean_possible = LC::usable(hc) && (wave != 0 || coefs_ok);
    bool lean = false;           // decided at the end of every step from the values the next step reads
    bool in_o_last = false;
    unsigned n_lean = 0;
    unsigned long long wmask = 0;  // bit (t & 63): step t raised a witness in this wave
    asm volatile("; loop-invariant scalars are in their registers" : : "s"(rsw), "s"(degw), "s"(ocw), "s"(nslots));
    const unsigned nloop = g.nsteps - 1;  // steps [0, nloop) publish and request; the last step does neither
    // the witness words of steps [t0, t0 + 64) that lie before the last step
    auto flush_witnesses = [&](unsigned t0) {
        if (((wmask >> lane) & 1ull) && t0 + lane < nloop) wit_raise(&wit[t0 + lane]);
        wmask = 0;
    };
    // STEADY STATE: lean steps [t, t_end), all before the last step, written as one tight loop per role (W0: the wave that
    // holds position 0 reads coefficients; the others read the ring below) — no role, witness-mode or last-step decision
    // inside.  Leaves early, BEFORE executing step t, if that step's output fails the regime test, and AFTER it if the
    // boundary value for the next step is no lean operand; the generic step below takes over from there.
    auto lean_run = [&](auto w0_tag, auto wit_tag, unsigned& t, unsigned t_end) {
        constexpr bool W0 = decltype(w0_tag)::value, WIT = decltype(wit_tag)::value;
        const double* const req = W0 ? coef_l + 1 : ring_b;
        for (; t < t_end; ++t) {
            const unsigned shw = rsw + 1 < degw ? rsw + 1 : degw;
            const RingWord rw = ring_request<E>(req + t, nslots);
            V shifted = wave_shr1_any<E>(cur);
            if (!W0 && lane == 0) shifted = below;
            const bool act = kw < shw, t2 = kw < rsw, t1 = kwm1 < rsw && act;
            const V v = lc.step(t1, t2, takes_c, shifted, cur, c_cur);
            if (LC::CHECKED && any_lane(act && !LC::result_ok(v))) {
                lean = false;
                return;
            }
            cur = v;
            rsw = shw;
            in_o_last = act;
            n_lean++;
            if (WIT && any_lane(act && kw >= wit_from && !E::is_zero(v))) wmask |= 1ull << (t & 63u);
            ring_put(post_v + (size_t)t * post_vstep, post_vplane, v);
            if (W0) {
                c_cur = ring_value<E>(rw);
            } else {
                below = ring_receive<E>(ring_b + t, nslots, rw);
                // position 64 * wave reads `below` in the next step iff 64 * wave - 1 < rsw: only then must it be usable
                if (LC::CHECKED && any_lane(lane == 0 && kwm1 < rsw && !LC::operand_ok(below))) {
                    lean = false;
                    ++t;
                    return;
                }
            }
        }
    };
    // (round 6) A wave joins the loop at the first step that reaches its positions.  The accumulator's extent along w grows by one
    // per step (rsw(t) = min(rs0 + t, deg): a POINT coefficient box adds nothing beyond position 0), so until step
    // t_s = 64 * wave - rs0 every position of the wave lies outside every box of the step — nothing computed, nothing stored, and
    // what it would publish is read by a wave that joins later still.  Step t_s reads position 64 * wave - 1 as it is after step
    // t_s - 1: slot t_s - 1 of the ring below.  A 101-wide line's second wave idles through 64 of its 101 steps — a third of the
    // launch's instructions (mixture --bounds: k_horner_pipe_point_batch is 79 % of the device time).
    unsigned t_join = 0;
    if (wave) {
        const unsigned first_pos = wave * 64u, rs0w = g.rs0[g.w];
        if (first_pos > rs0w && first_pos < degw && first_pos - rs0w <= nloop && !(g.diag & 32)) {
            t_join = first_pos - rs0w;
            rsw = first_pos;  // = min(rs0 + t_join, deg)
            below = ring_receive<E>(ring_b + (t_join - 1u), nslots, ring_request<E>(ring_b + (t_join - 1u), nslots));
        }
    }
    for (unsigned t = t_join; t <= nloop;) {
        if (lean && t < nloop && !(g.diag & 16)) {
            const unsigned chunk_end = (t | 63u) + 1u;  // witness words are flushed per 64 steps
            const unsigned t_end = (wit && chunk_end < nloop) ? chunk_end : nloop;
            if (wave == 0) {
                if (wit) lean_run(std::true_type{}, std::true_type{}, t, t_end);
                else lean_run(std::true_type{}, std::false_type{}, t, t_end);
            } else {
                if (wit) lean_run(std::false_type{}, std::true_type{}, t, t_end);
                else lean_run(std::false_type{}, std::false_type{}, t, t_end);
            }
            if (wit && (t & 63u) == 0u && t > 0) flush_witnesses(t - 64u);  // ran up to a chunk boundary
            continue;
        }
        // ---- one generic step: step 0, the last step, lines outside the lean regime, a lean step whose output test failed
        const bool last = t == nloop, first = t == 0;
        const unsigned shw = rsw + 1 < degw ? rsw + 1 : degw;
        const RingWord rw = ring_request<E>(req_base + (wave ? t : (t < nloop ? t + 1 : nloop)), nslots);
        V shifted = wave_shr1_any<E>(cur);
        if (lane == 0) shifted = below;
        bool witness = false, val_ok = true;
        {
            const unsigned upper = shw - 1 < rsw ? shw - 1 : rsw;
            const unsigned osw = ocw > shw ? ocw : shw;
            const bool in_o = have && (first ? off_o0 : true) && kw < osw;
            if (in_o) {
                const bool in_p = (first ? off_p0 : true) && kw < shw;
                const bool in_r = (first ? off_p0 : true) && kw < rsw;
                const bool t1 = in_p && kw >= 1 && kwm1 < upper;
                const bool t2 = in_p && !g.c_zero && in_r;
                V xm1 = E::one(), x = E::one();
                if (t1) xm1 = shifted;
                if (t2) x = cur;
                const V coef = takes_c ? c_cur : E::zero();
                const V v = horner_elem<E>(hc, in_p, t1, t2, takes_c, xm1, x, coef);
                cur = v;
                witness = kw >= wit_from && !E::is_zero(v);
                val_ok = LC::operand_ok(v);
            }
            in_o_last = in_o;
            rsw = osw;
        }
        if (wit && any_lane(witness)) wmask |= 1ull << (t & 63u);
        if (wit && ((t & 63u) == 63u || last)) flush_witnesses(t & ~63u);
        if (last) break;
        ring_put(post_v + (size_t)t * post_vstep, post_vplane, cur);
        if (wave == 0) {
            c_cur = ring_value<E>(rw);
        } else {
            below = ring_receive<E>(ring_b + t, nslots, rw);
            if (LC::CHECKED && lane == 0 && kwm1 < rsw) val_ok = val_ok && LC::operand_ok(below);
        }
        // the next step is lean iff every value it will read is a lean operand
        lean = lean_possible && (!LC::CHECKED || !any_lane(!val_ok));
        ++t;
    }
    if (have && in_o_last) E::st(out, plane, foff_b + (size_t)kw * wstr_f, cur);
    if (g.stat && lane == 0) {  // GFT_HORNER_DIAG & 64: how many of the steps ran lean (per wave)
        atomicAdd(&g.stat[0], (unsigned long long)n_lean);
        atomicAdd(&g.stat[1], (unsigned long long)g.nsteps);
    }
}

template <class E>
__global__ void __launch_bounds__(1024) k_horner_pipe_point(const double* __restrict__ res0, size_t rp0,
                                                            const double* __restrict__ a, size_t ap,
                                                            double* __restrict__ out, size_t plane, HornerLoopArgs g,
                                                            unsigned* __restrict__ wit) {
    extern __shared__ double hp_lds[];
    horner_pipe_point_line<E>(res0, rp0, a, ap, out, plane, g, wit, blockIdx.x, blockDim.x >> 6, hp_lds);
}
// the LDS a loop needs on the POINT pipeline, or 0 if it does not run there (lines > 1024, non-point coefficient boxes, rings
// beyond 60 KB, GFT_HORNER_PIPE / GFT_HORNER_LEAN = 0)
template <class E>
static size_t horner_pipe_point_lds(const HornerLoopArgs& args) {
    static const bool pipe_on = true;
    static const bool lean_on = true;
    const unsigned lw = args.fs[args.w];
    if (!pipe_on || !lean_on || lw > 1024) return 0;
    const bool point = args.coeff_scalar || args.oc[args.w] == 1;
    if (!point) return 0;
    const unsigned nwv = (lw + 63) / 64;
    const size_t lds = (size_t)nwv * E::W * args.nsteps * sizeof(double) + (size_t)128 * 8 + 16;
    return lds <= 60 * 1024 ? lds : 0;
}
template <class E>
bool K<E>::horner_can_ride(const HornerLoopArgs& args) {
    return args.nsteps != 0 && horner_pipe_point_lds<E>(args) != 0 && args.guard == nullptr;
}
// a batch of whole loops on the POINT pipeline (each item = what a rider is)
template <class E>
__global__ void __launch_bounds__(1024) k_horner_pipe_point_batch(const HornerRider* __restrict__ items) {
    extern __shared__ double hp_lds[];
    __shared__ __align__(16) unsigned char s_args[sizeof(HornerRider)];
    const HornerRider& A = item_to_lds<HornerRider>(items, s_args);
    if (blockIdx.x >= A.lines) return;
    horner_pipe_point_line<E>(A.res0, A.rp0, A.a, A.ap, A.out, A.plane, A.g, nullptr, blockIdx.x, (A.g.fs[A.g.w] + 63u) >> 6, hp_lds);
}
template <class E>
typename K<E>::Geometry K<E>::horner_geometry(const HornerLoopArgs& a, unsigned lines) {
    Geometry g;
    g.ok = lines != 0 && horner_can_ride(a);
    g.gx = lines;
    g.threads = (a.fs[a.w] + 63) / 64 * 64;
    g.lds = horner_pipe_point_lds<E>(a);
    return g;
}
template <class E>
void K<E>::horner_batch(hipStream_t st, const HornerRider* items, unsigned n, const Geometry& g) {
    GFT_LAUNCH((k_horner_pipe_point_batch<E>), dim3(g.gx, n), dim3(g.threads), g.lds, st, items);
}
template <class E>
void K<E>::horner_linear_loop(hipStream_t st, const double* res0, size_t res0_plane, const double* a, size_t a_plane, double* out,
                              size_t plane, const HornerLoopArgs& args, unsigned lines, unsigned* wit) {
    if (args.nsteps == 0 || lines == 0) return;
    const unsigned lw = args.fs[args.w];
    // wave pipeline (k_horner_linear_pipe): lines up to 1024 whose boundary rings fit LDS
    static const bool pipe_on = true;
    if (const size_t lds0 = horner_pipe_point_lds<E>(args)) {
        GFT_LAUNCH((k_horner_pipe_point<E>), dim3(lines), dim3((lw + 63) / 64 * 64), lds0, st, res0, res0_plane, a, a_plane, out, plane, args, wit);
        return;
    }
    if (pipe_on && lw <= 1024) {
        const unsigned nwv = (lw + 63) / 64;
        const bool point = args.coeff_scalar || args.oc[args.w] == 1;
        const size_t lds = (size_t)(nwv - 1 + (point ? 1 : 0)) * E::W * args.nsteps * sizeof(double) + (point ? (size_t)128 * 8 : 0) + 16;
        if (lds <= 60 * 1024) {
            if (point)
                GFT_LAUNCH((k_horner_linear_pipe<E, 8, true>), dim3(lines), dim3(nwv * 64), lds, st, res0, res0_plane, a, a_plane, out, plane, args, wit);
            else
                GFT_LAUNCH((k_horner_linear_pipe<E, 8, false>), dim3(lines), dim3(nwv * 64), lds, st, res0, res0_plane, a, a_plane, out, plane, args, wit);
            return;
        }
    }
    // one position per thread up to 1024-long lines (measured: two per thread is 10 % slower — the element chains
    // are not interleaved by the compiler, more waves hide the latency better)
    static const unsigned per_thread = 1;
    unsigned threads = std::min<unsigned>(1024, ((lw + per_thread - 1) / per_thread + 63) / 64 * 64);
    size_t lds = (size_t)2 * E::W * args.lw_pad * sizeof(double);
    if (lw <= threads)
        GFT_LAUNCH((k_horner_linear_loop<E, 1, 8>), dim3(lines), dim3(threads), lds, st, res0, res0_plane, a, a_plane, out, plane, args, wit);
    else
        GFT_LAUNCH((k_horner_linear_loop<E, HL_EPT_MAX, 4>), dim3(lines), dim3(threads), lds, st, res0, res0_plane, a, a_plane, out, plane, args, wit);
}

template <class E>
__global__ void __launch_bounds__(256) k_witness(DView t, unsigned* flag, size_t total) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // another block was faster
    int found = 0;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total && !found;
         lin += (size_t)gridDim.x * blockDim.x) {
        if (E::is_zero(E::ld(t.p, t.plane, lin))) continue;
        size_t r = lin;
        int nz = 0;
        bool big = false;
#pragma unroll 1
        for (int ax = t.sh.nd - 1; ax >= 0; --ax) {
            unsigned d = t.sh.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            if (k) nz++;
            if (k >= 2) big = true;
        }
        if (big || nz >= 2) found = 1;
    }
    if (__syncthreads_or(found) && threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class E>
void K<E>::witness(hipStream_t st, const DView& t, unsigned* flag) {
    size_t total = 1;
    for (int i = 0; i < t.sh.nd; ++i) total *= t.sh.d[i];
    if (total == 0) return;
    unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 64);  // dense tensors are settled by the first elements
    GFT_LAUNCH(k_witness<E>, dim3(blocks), dim3(256), 0, st, t, flag, total);
}

__global__ void __launch_bounds__(256) k_witness_verdict(const unsigned* __restrict__ flags, unsigned n, Mailbox mb) {
    int missing = 0;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
        if (__hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) missing = 1;
    const int any = __syncthreads_or(missing);
    if (threadIdx.x == 0) {
        mb.payload[0] = any ? 1.0 : 0.0;
        mailbox_publish(mb);
    }
}
void witness_verdict(hipStream_t st, const unsigned* flags, unsigned n, const Mailbox& mb) {
    GFT_LAUNCH(k_witness_verdict, dim3(1), dim3(256), 0, st, flags, n, mb);
}

template void K<EF64>::horner_linear(hipStream_t, const double*, size_t, const double*, size_t, double*, size_t, const HornerArgs&);
template void K<EIv>::horner_linear(hipStream_t, const double*, size_t, const double*, size_t, double*, size_t, const HornerArgs&);
template void K<EF64>::horner_linear_loop(hipStream_t, const double*, size_t, const double*, size_t, double*, size_t, const HornerLoopArgs&, unsigned, unsigned*);
template void K<EIv>::horner_linear_loop(hipStream_t, const double*, size_t, const double*, size_t, double*, size_t, const HornerLoopArgs&, unsigned, unsigned*);
template bool K<EF64>::horner_can_ride(const HornerLoopArgs&);
template bool K<EIv>::horner_can_ride(const HornerLoopArgs&);
template K<EF64>::Geometry K<EF64>::horner_geometry(const HornerLoopArgs&, unsigned);
template K<EIv>::Geometry K<EIv>::horner_geometry(const HornerLoopArgs&, unsigned);
template void K<EF64>::horner_batch(hipStream_t, const HornerRider*, unsigned, const K<EF64>::Geometry&);
template void K<EIv>::horner_batch(hipStream_t, const HornerRider*, unsigned, const K<EIv>::Geometry&);
template void K<EF64>::witness(hipStream_t, const DView&, unsigned*);
template void K<EIv>::witness(hipStream_t, const DView&, unsigned*);

}  // namespace gft
