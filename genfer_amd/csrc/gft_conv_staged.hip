// LDS-staged reference-order convolution (f64 and interval).
//
// Same arithmetic as k_conv_naive (gft_kernels.hip): every output element is accumulated by ONE
// thread in exactly the reference's loop order (mt:971-1012: outer axes lexicographic ascending,
// last axis' partial sum from zero, separate multiply and add), so the result is bit-identical to
// the CPU algorithm — only the data movement differs.  The naive kernel streams both operands from
// L1/L2 for every MAC (4 global loads per interval MAC); here a workgroup owns up to CH outputs that
// share their "outer" multi-index K_o (all axes but the last S in {1,2}), walks the outer j_o in
// reference order, and for every j_o stages the x sub-tensor x[j_o, ...] and the y sub-tensor
// y[K_o - j_o, ...] (the last S axes, contiguous in HBM) into LDS once for all its threads.
//
// LDS layout: x sub-tensor dense; y sub-tensor with row pitch nb (= the output's last-axis length), so
// that a thread's y address is (its linear output index) - (j_a * nb + j_b): consecutive lanes read
// consecutive LDS words for every (j_a, j_b) => conflict-free ds_read_b64, while the x address is
// wave-uniform (LDS broadcast).  Only the rows/columns the chunk can reach are staged.
//
// Work decomposition: blocks = (outer K_o within the slab) x (chunks of the staged subspace), issued
// heaviest-first (largest K first), 256..1024 threads by LDS footprint so that a CU always holds
// >= 4 waves per SIMD.  HBM traffic per block step is (|x sub| + |y sub|) * 8 * W bytes for
// CH * (MACs per output) work: the kernel is VALU/LDS-bound (2 ds_read + mul + add per f64 MAC).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <map>

#include "gft_kernels.hpp"

namespace gft {

namespace {

constexpr int MAXO = MAXD - 1;  // max number of outer axes
static const size_t R_BATCH_MAX = (size_t)(8);

struct StagedArgs {
    int S;                   // staged axes (1 or 2)
    int no;                  // outer axes = nd - S
    unsigned chunks;         // chunks of the staged subspace per outer K
    unsigned long long sub_lo, sub_hi;  // linear output range inside the staged subspace
    unsigned sxa, sxb, sya, syb, na, nb;  // staged extents (S == 1: sxa = sya = na = 1)
    unsigned xcap, ycap;     // LDS plane sizes (doubles) of the x / y regions
    unsigned batch;          // S == 1: consecutive rows of the LAST outer axis staged per barrier pair (>= 1)
    unsigned rw;             // S == 1 with inner-from-zero sums: thread groups that split the rows of a batch (>= 1)
};

// One row product  sum_{j = lo}^{hi - 1} x[xb + j] * y[yb - j]  formed from zero in ascending j (mul_1d, mt:971-982).
// `pos`: every staged element is in the interval functor's positive regime — the sum is then formed with mac_pos and
// recomputed with the general mac if some lane of the wave hit a case mac_pos cannot represent.
template <class E>
__device__ inline typename E::V inner_sum(const double* xl, size_t xcap, const double* yl, size_t ycap, unsigned xb, unsigned yb,
                                          unsigned lo, unsigned hi, int regime) {
    typedef typename E::V V;
    const bool pos = regime == 1;
    if constexpr (E::HAS_POS) {
        if (regime == 2) {
            // finite regime (gft_elem.hpp): no staged element can short-circuit; unguarded outward steps, one test of the
            // finished sum, the general mac below for a sum that fails it
            V inner = E::zero();
            bool bad = false;
            if (lo < hi) {
                inner = E::mul_fin(E::ld(xl, xcap, xb + lo), E::ld(yl, ycap, yb - lo));  // [0,0] + m returns m unchanged
#pragma unroll 4
                for (unsigned j = lo + 1; j < hi; ++j) inner = E::mac_fin(inner, E::ld(xl, xcap, xb + j), E::ld(yl, ycap, yb - j));
                bad = !E::fin_result_ok(inner);
            }
            if (!any_lane(bad)) return inner;
        }
        if (pos) {
            // No per-term checks: the two ways a term can leave the regime poison the running sum — a product that
            // underflows to zero makes dec_pos produce a NaN pattern, an upper bound that reaches inf makes inc_pos
            // produce one — and NaNs stay NaNs through the following adds and integer steps.  One test of the finished
            // sum (lower bound still positive, upper bound finite) therefore covers every term; a sum that fails it is
            // recomputed with the general mac below.  (The loop was issue-bound: 15 -> 10 instructions per interval MAC.)
            bool bad = false;
            V inner = E::zero();
            if (lo < hi) {
                inner = E::mul_pos(E::ld(xl, xcap, xb + lo), E::ld(yl, ycap, yb - lo));  // [0,0] + m returns m unchanged
#pragma unroll 4
                for (unsigned j = lo + 1; j < hi; ++j)
                    inner = E::mac_pos_unchecked(inner, E::ld(xl, xcap, xb + j), E::ld(yl, ycap, yb - j));
                bad = !E::pos_first_ok(inner) || !E::pos_result_ok(inner);
            }
            if (!any_lane(bad)) return inner;
        }
    }
    V inner = E::zero();
#pragma unroll 4
    for (unsigned j = lo; j < hi; ++j) inner = E::mac(inner, E::ld(xl, xcap, xb + j), E::ld(yl, ycap, yb - j));
    return inner;
}

template <class E, bool INNER0, int S>
__global__ void __launch_bounds__(1024)
k_conv_staged(const double* __restrict__ x, size_t xp, const double* __restrict__ y, size_t yp,
              double* __restrict__ z, size_t zp, ConvArgs a, StagedArgs g) {
    typedef typename E::V V;
    extern __shared__ double smem[];
    if (a.guard && *a.guard != a.guard_epoch) return;
    double* xl = smem;                         // [W][xcap]
    double* yl = smem + (size_t)E::W * g.xcap;  // [W][ycap]
    // The block is NT threads: CH outputs x rw "row groups".  The reference forms every row's partial sum from zero
    // and adds the sums in row order (mt:971-982), so the row sums of a batch are independent: group q computes the
    // rows t = q, q + rw, .. of the batch for the block's CH outputs, the sums go through LDS and group 0 adds them
    // in ascending row order — same operations per output, 1/rw of the serial chain (a recurrence row step has 64
    // outputs and thousands of MACs each: nothing else is parallel).
    const unsigned NT = blockDim.x, tid = threadIdx.x;
    const unsigned rw = INNER0 ? g.rw : 1;
    const unsigned CH = NT / rw, q = tid / CH, otid = tid - q * CH;

    // ---- which outputs --------------------------------------------------------------------------
    const unsigned long long b = (unsigned long long)(gridDim.x - 1 - blockIdx.x);  // heaviest first
    const unsigned chunk = (unsigned)(b % g.chunks);
    unsigned long long ko_lin = b / g.chunks;
    unsigned K[MAXO], lo[MAXO], cnt[MAXO], pos[MAXO];
    unsigned long long steps = 1;
    size_t zoff = 0;
#pragma unroll
    for (int ax = MAXO - 1; ax >= 0; --ax) {
        K[ax] = lo[ax] = pos[ax] = 0;
        cnt[ax] = 1;
        if (ax < g.no) {
            unsigned d = a.zs[ax], base = 0;
            if (ax == 0) {
                d = a.slab_hi - a.slab_lo;
                base = a.slab_lo;
            }
            // (the fastest outer index rotates by the slower ones: consecutive workgroups go to consecutive XCDs, and with a
            // power-of-two extent an XCD would otherwise only see the indices k = x (mod 8) — see k_conv_rows_rb)
            const bool rot = ax == g.no - 1 && ax > 0;
            K[ax] = base + (unsigned)(rot ? (ko_lin % d + ko_lin / d) % d : ko_lin % d);
            ko_lin /= d;
            zoff += (size_t)K[ax] * a.zstr[ax];
            unsigned l = (K[ax] + 1 > a.ys[ax]) ? (K[ax] + 1 - a.ys[ax]) : 0;
            unsigned h = (K[ax] + 1 < a.xs[ax]) ? (K[ax] + 1) : a.xs[ax];
            if (ax == 0) {
                if (l < (unsigned)a.j0_min) l = (unsigned)a.j0_min;
                if (a.j0_excl && h > K[ax]) h = K[ax];
            }
            lo[ax] = l;
            cnt[ax] = h > l ? h - l : 0;
            steps *= cnt[ax];
        }
    }
    // S == 1: the rows x[.., j, :] for consecutive j of the last outer axis are contiguous in HBM (and so are the
    // rows y[.., K - j, :]), so `batch` of them are staged per barrier pair — the global-load latency of a step is
    // paid once per batch instead of once per row (a recurrence slab step is one wave per block: nothing else
    // hides it).  The rows of a batch are consumed in ascending j, i.e. in the reference's order.
    const int lastax = (S == 1) ? g.no - 1 : -1;
    unsigned nbatch = 1, cnt_last = 1, lo_last = 0, K_last = 0;
#pragma unroll
    for (int ax = 0; ax < MAXO; ++ax)
        if (ax == lastax) {
            cnt_last = cnt[ax];
            lo_last = lo[ax];
            K_last = K[ax];
        }
    if (lastax >= 0 && cnt_last > 0) {
        nbatch = (cnt_last + g.batch - 1) / g.batch;
        steps = steps / cnt_last * nbatch;
    }
    const unsigned long long first = g.sub_lo + (unsigned long long)chunk * CH;
    const unsigned long long lin = first + otid;
    const bool active = lin < g.sub_hi;
    unsigned long long last = first + CH - 1;
    if (last > g.sub_hi - 1) last = g.sub_hi - 1;
    const unsigned ka = S == 2 ? (unsigned)(lin / g.nb) : 0;
    const unsigned kb = (unsigned)(lin % g.nb);
    // staged extents this chunk can reach
    unsigned xrows = 1, yrows = 1, xcols, ycols;
    if (S == 2) {
        unsigned ka_hi = (unsigned)(last / g.nb);
        xrows = ka_hi + 1 < g.sxa ? ka_hi + 1 : g.sxa;
        yrows = ka_hi + 1 < g.sya ? ka_hi + 1 : g.sya;
        xcols = g.sxb;
        ycols = g.syb;
    } else {
        unsigned kb_hi = (unsigned)last;
        xcols = kb_hi + 1 < g.sxb ? kb_hi + 1 : g.sxb;
        ycols = kb_hi + 1 < g.syb ? kb_hi + 1 : g.syb;
    }
    const unsigned nx = xrows * xcols, ny = yrows * ycols;

    // ---- per-thread loop bounds on the staged axes (constant over the outer steps) ----------------
    const bool a_is_axis0 = (S == 2) && g.no == 0;
    const bool b_is_axis0 = (S == 1) && g.no == 0;
    unsigned lo_a = 0, hi_a = 1;
    if (S == 2) {
        lo_a = (ka + 1 > g.sya) ? (ka + 1 - g.sya) : 0;
        hi_a = (ka + 1 < g.sxa) ? (ka + 1) : g.sxa;
        if (a_is_axis0) {
            if (lo_a < (unsigned)a.j0_min) lo_a = (unsigned)a.j0_min;
            if (a.j0_excl && hi_a > ka) hi_a = ka;
        }
    }
    unsigned lo_b = (kb + 1 > g.syb) ? (kb + 1 - g.syb) : 0;
    unsigned hi_b = (kb + 1 < g.sxb) ? (kb + 1) : g.sxb;
    if (b_is_axis0) {
        if (lo_b < (unsigned)a.j0_min) lo_b = (unsigned)a.j0_min;
        if (a.j0_excl && hi_b > kb) hi_b = kb;
    }
    const bool desc_a = a_is_axis0 && a.j0_desc;
    const bool desc_b = b_is_axis0 && a.j0_desc;
    const bool desc_0 = g.no > 0 && a.j0_desc;

    const size_t zlin = zoff + (size_t)lin;
    V acc = E::zero();
    if (active && a.accumulate) acc = E::ld(z, zp, zlin);

    for (unsigned long long step = 0; step < steps; ++step) {
        // offsets of the staged sub-tensors for the current outer j
        size_t xoff = 0, yoff = 0, xpitch = 0, ypitch = 0;
        unsigned rows = 1;  // rows of this batch (S == 1)
#pragma unroll
        for (int ax = 0; ax < MAXO; ++ax) {
            if (ax < g.no) {
                if (ax == lastax) {
                    // descending j0 (log): batches are taken from the top and their rows consumed downwards
                    rows = cnt_last - pos[ax] * g.batch < g.batch ? cnt_last - pos[ax] * g.batch : g.batch;
                    const unsigned jlo = (ax == 0 && desc_0) ? (lo_last + cnt_last - pos[ax] * g.batch - rows)
                                                            : (lo_last + pos[ax] * g.batch);
                    xoff += (size_t)jlo * a.xstr[ax];
                    yoff += (size_t)(K_last - (jlo + rows - 1)) * a.ystr[ax];  // lowest y row of the batch
                    xpitch = a.xstr[ax];
                    ypitch = a.ystr[ax];
                } else {
                    unsigned j = (ax == 0 && desc_0) ? (lo[ax] + cnt[ax] - 1 - pos[ax]) : (lo[ax] + pos[ax]);
                    xoff += (size_t)j * a.xstr[ax];
                    yoff += (size_t)(K[ax] - j) * a.ystr[ax];
                }
            }
        }
        const bool rows_desc = lastax == 0 && desc_0;
        __syncthreads();  // everyone is done with the previous sub-tensors
        // Interval tensors: while staging, note whether every staged element is in the positive regime
        // (gft_elem.hpp EIv::pos_ok); the block then runs its inner sums with mac_pos (same bits, ~1/4 of the
        // instructions) and falls back to the general mac for the sums mac_pos cannot represent.
        int not_pos = 0, not_fin = 0;
        if (S == 1 && rows > 1) {
            for (unsigned i = tid; i < rows * xcols; i += NT) {
                unsigned r = i / xcols, c = i - r * xcols;
                V v = E::ld(x, xp, xoff + (size_t)r * xpitch + c);
                if (E::HAS_POS && !E::pos_ok(v)) not_pos = 1;
                if (E::HAS_POS && !E::fin_ok(v)) not_fin = 1;
                E::st(xl, g.xcap, i, v);
            }
            for (unsigned i = tid; i < rows * ycols; i += NT) {
                unsigned r = i / ycols, c = i - r * ycols;
                V v = E::ld(y, yp, yoff + (size_t)r * ypitch + c);
                if (E::HAS_POS && !E::pos_ok(v)) not_pos = 1;
                if (E::HAS_POS && !E::fin_ok(v)) not_fin = 1;
                E::st(yl, g.ycap, i, v);
            }
        } else {
        for (unsigned i = tid; i < nx; i += NT) {
            V v = E::ld(x, xp, xoff + i);
            if (E::HAS_POS && !E::pos_ok(v)) not_pos = 1;
                if (E::HAS_POS && !E::fin_ok(v)) not_fin = 1;
            E::st(xl, g.xcap, i, v);
        }
        if (S == 2 && g.syb != g.nb) {
            for (unsigned i = tid; i < ny; i += NT) {
                unsigned r = i / g.syb, c = i - r * g.syb;
                V v = E::ld(y, yp, yoff + i);
                if (E::HAS_POS && !E::pos_ok(v)) not_pos = 1;
                if (E::HAS_POS && !E::fin_ok(v)) not_fin = 1;
                E::st(yl, g.ycap, (size_t)r * g.nb + c, v);
            }
        } else {
            for (unsigned i = tid; i < ny; i += NT) {
                V v = E::ld(y, yp, yoff + i);
                if (E::HAS_POS && !E::pos_ok(v)) not_pos = 1;
                if (E::HAS_POS && !E::fin_ok(v)) not_fin = 1;
                E::st(yl, g.ycap, i, v);
            }
        }
        }
        // regime of the block's inner sums: 1 positive, 2 finite (no operand can short-circuit), 0 general
        int regime_pos = 0;
        if (E::HAS_POS) {
            if (__syncthreads_or(not_pos) == 0) regime_pos = 1;            // (block-uniform branch)
            else if (__syncthreads_or(not_fin) == 0) regime_pos = 2;
        } else {
            __syncthreads();
        }

        if (S == 2 && INNER0 && rw > 1) {
            // plane mode: the rows j_a of the staged plane play the role of the batch rows, 8 at a time
            double* sums = yl + (size_t)E::W * g.ycap;  // [plane][8][CH]
            const unsigned RB = 8;
            const size_t splane = (size_t)RB * CH;
            const bool nonempty = hi_b > lo_b;
            const unsigned cnt_a = hi_a > lo_a ? hi_a - lo_a : 0;
            for (unsigned c0 = 0; c0 < xrows; c0 += RB) {  // xrows bounds every thread's cnt_a (uniform loop)
                const unsigned cend = c0 + RB < cnt_a ? c0 + RB : cnt_a;
                if (active && nonempty) {
                    for (unsigned t = c0 + q; t < cend; t += rw) {
                        const unsigned ja = desc_a ? (hi_a - 1 - t) : (lo_a + t);
                        const unsigned xb = ja * g.sxb, yb = (ka - ja) * g.nb + kb;
                        V inner = inner_sum<E>(xl, g.xcap, yl, g.ycap, xb, yb, lo_b, hi_b, regime_pos);
                        E::st(sums, splane, (size_t)(t - c0) * CH + otid, inner);
                    }
                }
                __syncthreads();
                if (active && nonempty && q == 0)
                    for (unsigned t = c0; t < cend; ++t) acc = E::add(acc, E::ld(sums, splane, (size_t)(t - c0) * CH + otid));
                __syncthreads();
            }
        } else
        if (S == 1 && INNER0 && rw > 1) {
            double* sums = yl + (size_t)E::W * g.ycap;  // [plane][batch][CH]
            const size_t splane = (size_t)g.batch * CH;
            const bool nonempty = hi_b > lo_b;
            if (active && nonempty) {
                for (unsigned t = q; t < rows; t += rw) {  // t-th row in consumption order
                    const unsigned xr = rows_desc ? rows - 1 - t : t;
                    const unsigned xb = xr * xcols, yb = (rows - 1 - xr) * ycols + kb;
                    V inner = inner_sum<E>(xl, g.xcap, yl, g.ycap, xb, yb, lo_b, hi_b, regime_pos);
                    E::st(sums, splane, (size_t)t * CH + otid, inner);
                }
            }
            __syncthreads();
            if (active && nonempty && q == 0)
                for (unsigned t = 0; t < rows; ++t) acc = E::add(acc, E::ld(sums, splane, (size_t)t * CH + otid));
        } else
        if (active) {
            // S == 2: t walks the staged plane's rows j_a; S == 1: t walks the rows of the batch (ascending outer j)
            const unsigned cnt_a = S == 2 ? (hi_a > lo_a ? hi_a - lo_a : 0) : rows;
            for (unsigned t = 0; t < cnt_a; ++t) {
                const unsigned ja = desc_a ? (hi_a - 1 - t) : (lo_a + t);
                const unsigned xr = rows_desc ? rows - 1 - t : t;     // S == 1: staged row consumed t-th
                const unsigned xb = S == 2 ? ja * g.sxb : xr * xcols;  // wave-uniform when the wave shares ja
                const unsigned yb = (S == 2 ? (ka - ja) * g.nb : (rows - 1 - xr) * ycols) + kb;
                if (hi_b > lo_b) {
                    if (INNER0) {
                        V inner = inner_sum<E>(xl, g.xcap, yl, g.ycap, xb, yb, lo_b, hi_b, regime_pos);
                        acc = E::add(acc, inner);
                    } else {
                        const unsigned cnt_b = hi_b - lo_b;
                        for (unsigned u = 0; u < cnt_b; ++u) {
                            const unsigned j = desc_b ? (hi_b - 1 - u) : (lo_b + u);
                            acc = E::mac(acc, E::ld(xl, g.xcap, xb + j), E::ld(yl, g.ycap, yb - j));
                        }
                    }
                }
            }
        }

        // advance the outer odometer (last outer axis fastest)
        bool carry = true;
#pragma unroll
        for (int ax = MAXO - 1; ax >= 0; --ax) {
            if (ax < g.no && carry) {
                if (++pos[ax] == (ax == lastax ? nbatch : cnt[ax])) pos[ax] = 0;
                else carry = false;
            }
        }
    }
    if (active && q == 0) E::st(z, zp, zlin, acc);
}

template <class E, bool INNER0, int S>
bool launch(hipStream_t st, const double* x, size_t xp, const double* y, size_t yp, double* z, size_t zp,
            const ConvArgs& a, const StagedArgs& g, unsigned blocks, unsigned threads, size_t lds) {
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_staged<E, INNER0, S>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    GFT_LAUNCH((k_conv_staged<E, INNER0, S>), dim3(blocks), dim3(threads), lds, st, x, xp, y, yp, z, zp, a, g);
    return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Register-blocked rows (round 3): the large interval product
// ------------------------------------------------------------------------------------------
// k_conv_staged spends, per interval multiply-add, four 8-byte LDS reads (x and y, two planes each) beside ~10 VALU
// instructions, and a wave that owns 64 consecutive outputs c of a row runs j up to its HIGHEST c: a third of its lane
// slots lie outside the triangle j <= c.  At 128^3 it reaches 23 % of the VALU-issue roof (tools/bench_interval.py).  Here
//   * a lane owns TWO outputs, (.., k1, c) and (.., k1 + 1, c) — neighbours on the last outer axis P.  Their terms pair up
//     on the SAME y row: x[j_lead, k1 - t, :] (*) y[K - j_lead, t, :] goes to the first, x[j_lead, k1 + 1 - t, :] (*) the same y
//     row to the second; walking t downwards is ascending j1 for both, so each output still receives its row sums in
//     the reference's order (mt:984-1012), each formed from zero in ascending j (mt:971-982).  One y read (two planes)
//     serves two multiply-adds;
//   * a wave is 4 row pairs x 16 columns (a workgroup: 8 output rows, all columns): the steps j < c0 below a 16-column
//     tile are full, only the 16 steps of its own triangle are masked — 11 % of the lane slots outside the triangle
//     instead of 33 %.  A wave takes the tiles ct and ntiles - 1 - ct one after the other, so all waves of a workgroup do
//     the same number of steps;
//   * x rows are staged with their two planes interleaved: the coefficient (lo, hi) of a step is ONE 16-byte read, four
//     distinct addresses per wave (one per row pair).
// Regimes (positive / finite / general, gft_elem.hpp) are chosen per y row from per-row flags computed by one pass over the
// operands (k_row_flags) — worst case over the x rows the workgroup pairs with it.  A sum that fails its regime's test is
// recomputed with the general multiply-add, as in k_conv_staged.  Same operations on the same values in the same order =>
// same bits (tests: conv_rb_min_macs A/B, the oracle).
struct RbArgs {
    int no;               // outer axes = nd - 1 (1..3); the last of them is the paired axis P
    unsigned n_pg;        // groups of 8 output rows along P
    unsigned ntiles;      // 16-column tiles of a row
    unsigned n2, nx2;     // output (= y) / x row lengths
    unsigned tb;          // y rows staged per barrier pair = wave groups of a workgroup (one row each)
    unsigned ntw;         // waves of a group (two tiles each)
    size_t xrs[MAXO], yrs[MAXO];  // strides of the outer axes in ROWS
    const unsigned char* xflags;  // per row: bit 0 = some element is not pos_ok, bit 1 = some element is not fin_ok
    const unsigned char* yflags;
};
constexpr unsigned RB_ROWS = 8;  // output rows of a workgroup (4 lane groups x 2 outputs per lane)

// (both operands in one launch: the x rows first, then the y rows, flags likewise)
template <class E>
__global__ void __launch_bounds__(256) k_row_flags(const double* x, size_t xplane, size_t xrows, unsigned xlen, const double* y, size_t yplane,
                                                  size_t yrows, unsigned ylen, unsigned char* flags) {
    size_t row = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= xrows + yrows) return;
    flags += row;
    const bool isy = row >= xrows;  // (wave-uniform: a wave is one row)
    const double* p = isy ? y : x;
    const size_t plane = isy ? yplane : xplane;
    const unsigned len = isy ? ylen : xlen;
    if (isy) row -= xrows;
    const unsigned lane = threadIdx.x & 63u;
    bool np = false, nf = false;
    for (unsigned i = lane; i < len; i += 64) {
        const typename E::V v = E::ld(p, plane, row * len + i);
        np = np || !E::pos_ok(v);
        nf = nf || !E::fin_ok(v);
    }
    const bool anp = any_lane(np), anf = any_lane(nf);
    if (lane == 0) *flags = (unsigned char)((anp ? 1 : 0) | (anf ? 2 : 0));
}

// the finished row sums of one y row against the lane's two x rows, for the 16-column tile at c0; REG 1 positive, 2
// finite, 0 general.  x0 / x1: staged x rows, (lo, hi) interleaved; yl: the lane's position c in the staged y row.
// (Measured and dropped: the x coefficient through scalar loads — it is not wave-uniform with four row pairs per wave, and
// with one pair per wave the triangle is back; through one LDS read per 16 steps and a DPP row broadcast per step —
// 8 extra VALU slots per step cost more than the LDS reads they replace, 506 vs 400 ms at 128^3.)
template <class E, int REG>
__device__ __forceinline__ void rb_sums(const double* x0, const double* x1, const double* yl, unsigned ypl, unsigned c, unsigned c0,
                                        unsigned nx2, typename E::V& s0, typename E::V& s1) {
    typedef typename E::V V;
    auto first = [](V xv, V yv) {
        if (REG == 1) return E::mul_pos(xv, yv);
        if (REG == 2) return E::mul_fin(xv, yv);
        return E::mac(E::zero(), xv, yv);
    };
    auto next = [](V acc, V xv, V yv) {
        if (REG == 1) return E::mac_pos_unchecked(acc, xv, yv);
        if (REG == 2) return E::mac_fin(acc, xv, yv);
        return E::mac(acc, xv, yv);
    };
    const unsigned jend = nx2 < c0 + 16 ? nx2 : c0 + 16;  // no lane of this tile has a term beyond
    {   // j = 0: every lane's first term
        const V yv = Iv{yl[0], yl[ypl]};
        s0 = first(Iv{x0[0], x0[1]}, yv);
        s1 = first(Iv{x1[0], x1[1]}, yv);
    }
    unsigned j = 1;
    // steps every lane of the tile takes part in: j <= c0 (eight at a time; the long general multiply-add gains nothing
    // from unrolling)
    const unsigned jfull = c0 + 1 < jend ? c0 + 1 : jend;
    constexpr int UN = REG == 0 ? 1 : 8;
    for (; j + UN <= jfull; j += UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const V yv = Iv{yl[-(int)(j + u)], yl[(int)ypl - (int)(j + u)]};
            s0 = next(s0, Iv{x0[2 * (j + u)], x0[2 * (j + u) + 1]}, yv);
            s1 = next(s1, Iv{x1[2 * (j + u)], x1[2 * (j + u) + 1]}, yv);
        }
    }
    for (; j < jfull; ++j) {
        const V yv = Iv{yl[-(int)j], yl[(int)ypl - (int)j]};
        s0 = next(s0, Iv{x0[2 * j], x0[2 * j + 1]}, yv);
        s1 = next(s1, Iv{x1[2 * j], x1[2 * j + 1]}, yv);
    }
    // the tile's own triangle: lane c takes part while j <= c
#pragma unroll 5
    for (; j < jend; ++j) {
        if (j <= c) {
            const V yv = Iv{yl[-(int)j], yl[(int)ypl - (int)j]};
            s0 = next(s0, Iv{x0[2 * j], x0[2 * j + 1]}, yv);
            s1 = next(s1, Iv{x1[2 * j], x1[2 * j + 1]}, yv);
        }
    }
}

// one y row against one tile: both outputs of the lane.  v0 / v1: the lane's outputs take this row (else what it computes
// — on a clamped, staged x row — is discarded)
template <class E>
__device__ __forceinline__ void rb_row(int regime, bool v0, bool v1, const double* x0, const double* x1, const double* yl, unsigned ypl,
                                       unsigned c, unsigned c0, unsigned nx2, typename E::V& s0, typename E::V& s1) {
    bool redo = true;
    if (regime == 1) {
        rb_sums<E, 1>(x0, x1, yl, ypl, c, c0, nx2, s0, s1);
        const bool bad = (v0 && (!E::pos_first_ok(s0) || !E::pos_result_ok(s0))) || (v1 && (!E::pos_first_ok(s1) || !E::pos_result_ok(s1)));
        redo = any_lane(bad);
    } else if (regime == 2) {
        rb_sums<E, 2>(x0, x1, yl, ypl, c, c0, nx2, s0, s1);
        const bool bad = (v0 && !E::fin_result_ok(s0)) || (v1 && !E::fin_result_ok(s1));
        redo = any_lane(bad);
    }
    if (redo) rb_sums<E, 0>(x0, x1, yl, ypl, c, c0, nx2, s0, s1);
}

template <class E>
__global__ void __launch_bounds__(1024) k_conv_rows_rb(const double* __restrict__ x, size_t xp, const double* __restrict__ y, size_t yp,
                                                      double* __restrict__ z, size_t zp, ConvArgs a, RbArgs g) {
    typedef typename E::V V;
    // [tb][2][n2] staged y rows (+ 16 doubles of slack), [tb + 7][nx2][2] staged x rows, the rows' flags, then the row sums
    // of the wave groups 1 .. tb - 1: [group - 1][wave of the group][4 sums][2 planes][64 lanes]
    // The y rows of a batch are independent until their sums are ADDED (each sum is formed from zero): wave group r forms
    // the sums of row r, group 0 adds them in row order.  The longest chain of a workgroup — (k0 + 1)(k1 + 1) row sums for
    // the last outputs, as long as the whole job's share of one SIMD at 128^3 — is cut by the number of groups.
    extern __shared__ __align__(16) double smem[];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned grp = wave_all / g.ntw, wave = wave_all - grp * g.ntw;
    const unsigned q = lane >> 4, cl = lane & 15u;
    const int P = g.no - 1;
    double* const ys_l = smem;
    double* const xs_l = smem + (size_t)g.tb * 2 * g.n2 + 16;
    const unsigned xpitch = 2 * g.nx2 + 2;  // doubles per staged x row: the rows of a wave's four pairs start in different banks
    unsigned char* const fl_l = reinterpret_cast<unsigned char*>(xs_l + (size_t)(g.tb + RB_ROWS - 1) * xpitch);
    double* const sums_l = xs_l + (size_t)(g.tb + RB_ROWS - 1) * xpitch + (2 * g.tb + RB_ROWS + 8 + 7) / 8;  // [tb] y flags, [tb + 7] x flags
    // the wave's two column tiles
    const unsigned ctA = wave, ctB = g.ntiles - 1 - wave;
    const bool twoB = ctB != ctA;
    const unsigned c0A = ctA * 16, c0B = ctB * 16, cA = c0A + cl, cB = c0B + cl;
    // ---- which outputs: heaviest first, the row group fastest
    unsigned long long b = (unsigned long long)(gridDim.x - 1 - blockIdx.x);
    // (consecutive workgroups go to consecutive XCDs: with the row group simply the fastest index and a power-of-two number
    // of groups, an XCD would only ever see the groups pg = x (mod 8) — up to 2.4x the work of its neighbour's; the rotation
    // by the slower index spreads every residue over all XCDs: 128^3 424 -> see interval_product.txt)
    const unsigned pg = (unsigned)((b % g.n_pg + b / g.n_pg) % g.n_pg);
    b /= g.n_pg;
    const unsigned k1g = RB_ROWS * pg;  // the group's first output row on P
    unsigned K[MAXO], lo[MAXO], cnt[MAXO], pos[MAXO];
    unsigned long long lead_steps = 1;
    size_t zoff = 0;
#pragma unroll
    for (int ax = MAXO - 1; ax >= 0; --ax) {
        K[ax] = lo[ax] = pos[ax] = 0;
        cnt[ax] = 1;
        if (ax < P) {
            unsigned d = a.zs[ax], base = 0;
            if (ax == 0) {
                d = a.slab_hi - a.slab_lo;
                base = a.slab_lo;
            }
            K[ax] = base + (unsigned)(b % d);
            b /= d;
            zoff += (size_t)K[ax] * a.zstr[ax];
            lo[ax] = K[ax] + 1 > a.ys[ax] ? K[ax] + 1 - a.ys[ax] : 0;
            const unsigned h = K[ax] + 1 < a.xs[ax] ? K[ax] + 1 : a.xs[ax];
            cnt[ax] = h > lo[ax] ? h - lo[ax] : 0;
            lead_steps *= cnt[ax];
        }
    }
    unsigned nP = 1, nxP = 1, nyP = 1;
    size_t zstrP = 0;
#pragma unroll
    for (int ax = 0; ax < MAXO; ++ax)
        if (ax == P) {
            nP = a.zs[ax];
            nxP = a.xs[ax];
            nyP = a.ys[ax];
            zstrP = a.zstr[ax];
        }
    // the group's y rows: t from t_hi down to t_lo (the group's first output row reaches lowest)
    const unsigned ktop = k1g + RB_ROWS - 1 < nP - 1 ? k1g + RB_ROWS - 1 : nP - 1;
    const int t_hi = (int)(ktop < nyP - 1 ? ktop : nyP - 1);
    const int t_lo = k1g + 1 > nxP ? (int)(k1g + 1 - nxP) : 0;
    // this lane's pair of output rows
    const unsigned k1 = k1g + 2 * q;
    const bool has0 = k1 < nP, has1 = k1 + 1 < nP;
    V accA0 = E::zero(), accA1 = E::zero(), accB0 = E::zero(), accB1 = E::zero();
    const unsigned ypl = g.n2;
    for (unsigned long long step = 0; step < lead_steps; ++step) {
        size_t xrow_lead = 0, yrow_lead = 0;
#pragma unroll
        for (int ax = 0; ax < MAXO; ++ax)
            if (ax < P) {
                const unsigned j = lo[ax] + pos[ax];
                xrow_lead += (size_t)j * g.xrs[ax];
                yrow_lead += (size_t)(K[ax] - j) * g.yrs[ax];
            }
        for (int tt = t_hi; tt >= t_lo; tt -= (int)g.tb) {
            const int rows = tt - t_lo + 1 < (int)g.tb ? tt - t_lo + 1 : (int)g.tb;  // y rows tt, tt - 1, .., tt - rows + 1
            const int tbot = tt - rows + 1;
            // x rows the batch touches: j1 = k' - t over the group's output rows k' and the batch's rows t
            const int jlo = (int)k1g - tt > 0 ? (int)k1g - tt : 0;
            int jhi = (int)ktop - tbot;
            if (jhi > (int)nxP - 1) jhi = (int)nxP - 1;
            const int xrows = jhi >= jlo ? jhi - jlo + 1 : 0;
            __syncthreads();  // everyone is done with the previous batch
            // y rows (contiguous: P is the last outer axis): slot r holds row tt - r
            for (unsigned i = tid; i < (unsigned)rows * g.n2; i += blockDim.x) {
                const unsigned rr = i / g.n2, cc = i - rr * g.n2;
                const size_t src = (yrow_lead + (size_t)(tbot + (int)rr)) * g.n2 + cc;
                double* dst = ys_l + (size_t)((unsigned)rows - 1 - rr) * 2 * g.n2 + cc;
                dst[0] = y[src];
                dst[g.n2] = y[yp + src];
            }
            // x rows jlo .. jhi (contiguous), planes interleaved
            for (unsigned i = tid; i < (unsigned)xrows * g.nx2; i += blockDim.x) {
                const size_t src = (xrow_lead + (size_t)jlo) * g.nx2 + i;
                const unsigned xr = i / g.nx2, xc = i - xr * g.nx2;
                reinterpret_cast<double2*>(xs_l + (size_t)xr * xpitch)[xc] = double2{x[src], x[xp + src]};
            }
            if (tid < (unsigned)rows) fl_l[tid] = g.yflags[yrow_lead + (size_t)(tt - (int)tid)];
            if (tid >= 64 && tid - 64 < (unsigned)xrows) fl_l[g.tb + tid - 64] = g.xflags[xrow_lead + (size_t)jlo + (tid - 64)];
            __syncthreads();
            // this group's row r = grp of the batch: its sums (for the lane's two outputs in the wave's two tiles)
            V sA0 = E::zero(), sA1 = E::zero(), sB0 = E::zero(), sB1 = E::zero();
            auto row_terms = [&](int r, bool& v0, bool& v1, int& ja, int& jb) {  // which of the lane's outputs take row r
                const int t = tt - r;
                ja = (int)k1g - t;
                jb = (int)ktop - t;
                if (ja < 0) ja = 0;
                if (jb > (int)nxP - 1) jb = (int)nxP - 1;
                v0 = jb >= ja && has0 && t <= (int)k1 && k1 - (unsigned)t < nxP;
                v1 = jb >= ja && has1 && t <= (int)k1 + 1 && k1 + 1 - (unsigned)t < nxP;
            };
            if ((int)grp < rows && xrows > 0) {
                const int r = (int)grp, t = tt - r;
                bool v0, v1;
                int ja, jb;
                row_terms(r, v0, v1, ja, jb);
                if (jb >= ja) {  // (uniform)
                    // the x rows the workgroup pairs with this y row: k' - t for k' = k1g .. ktop, inside x's box
                    unsigned f = fl_l[r];
                    for (int jj = ja; jj <= jb; ++jj) f |= fl_l[g.tb + (unsigned)(jj - jlo)];
                    const int regime = (f & 1u) == 0u ? 1 : ((f & 2u) == 0u ? 2 : 0);
                    // (a lane without a term computes on a clamped row of [ja, jb]; its sums are not added)
                    int j10 = (int)k1 - t, j11 = (int)k1 + 1 - t;
                    j10 = j10 < ja ? ja : (j10 > jb ? jb : j10);
                    j11 = j11 < ja ? ja : (j11 > jb ? jb : j11);
                    const double* x0 = xs_l + (size_t)(j10 - jlo) * xpitch;
                    const double* x1 = xs_l + (size_t)(j11 - jlo) * xpitch;
                    const double* yrow = ys_l + (size_t)r * 2 * g.n2;
                    if (cA < g.n2) rb_row<E>(regime, v0, v1, x0, x1, yrow + cA, ypl, cA, c0A, g.nx2, sA0, sA1);
                    if (twoB && cB < g.n2) rb_row<E>(regime, v0, v1, x0, x1, yrow + cB, ypl, cB, c0B, g.nx2, sB0, sB1);
                }
                if (grp > 0) {
                    double* sl = sums_l + ((size_t)(grp - 1) * g.ntw + wave) * 512 + lane;
                    E::st(sl, 64, 0, sA0);
                    E::st(sl + 128, 64, 0, sA1);
                    E::st(sl + 256, 64, 0, sB0);
                    E::st(sl + 384, 64, 0, sB1);
                }
            }
            if (g.tb > 1) __syncthreads();
            if (grp == 0 && xrows > 0) {
                for (int r = 0; r < rows; ++r) {
                    bool v0, v1;
                    int ja, jb;
                    row_terms(r, v0, v1, ja, jb);
                    if (r > 0) {
                        const double* sl = sums_l + ((size_t)(r - 1) * g.ntw + wave) * 512 + lane;
                        sA0 = E::ld(sl, 64, 0);
                        sA1 = E::ld(sl + 128, 64, 0);
                        sB0 = E::ld(sl + 256, 64, 0);
                        sB1 = E::ld(sl + 384, 64, 0);
                    }
                    if (v0) {
                        accA0 = E::add(accA0, sA0);
                        accB0 = E::add(accB0, sB0);
                    }
                    if (v1) {
                        accA1 = E::add(accA1, sA1);
                        accB1 = E::add(accB1, sB1);
                    }
                }
            }
        }
        // advance the lead odometer (last lead axis fastest)
        bool carry = true;
#pragma unroll
        for (int ax = MAXO - 1; ax >= 0; --ax)
            if (ax < P && carry) {
                if (++pos[ax] == cnt[ax]) pos[ax] = 0;
                else carry = false;
            }
    }
    if (grp != 0) return;
    if (has0 && cA < g.n2) E::st(z, zp, zoff + (size_t)k1 * zstrP + cA, accA0);
    if (has1 && cA < g.n2) E::st(z, zp, zoff + (size_t)(k1 + 1) * zstrP + cA, accA1);
    if (twoB && has0 && cB < g.n2) E::st(z, zp, zoff + (size_t)k1 * zstrP + cB, accB0);
    if (twoB && has1 && cB < g.n2) E::st(z, zp, zoff + (size_t)(k1 + 1) * zstrP + cB, accB1);
}

// ------------------------------------------------------------------------------------------
// Row-pair sums (round 4): every row sum of the product as an independent task
// ------------------------------------------------------------------------------------------
// The reference orders only the ADDITIONS of the row sums; a row sum  S[a][b][c] = sum_j x[a][j] * y[b][c - j]  (formed from
// zero in ascending j, mt:971-982) depends on one x row a and one y row b, and goes to the single output row a + b.  So
// phase 1 is a batch of independent 1-d products, all (a, b) with a + b inside the result's box, and can take the layout
// of the fast f64 kernel (gft_conv_tiled.hip) that the fused kernels above cannot — there the x value of a step comes
// from LDS for every multiply-add because a wave spans four output rows:
//   lanes       64 y rows b (a T0 x T1 tile of the last two outer axes), staged in LDS once per workgroup and reused by
//               every x row the workgroup walks; a lane keeps 4 consecutive outputs c of ITS row sum in registers and
//               slides a 4-wide window of its y row (one 16-byte LDS read per 4 multiply-adds instead of three per two);
//   x           one row per task, wave-uniform: scalar loads, SGPR operands — no LDS, no VGPRs;
//   tasks       (x row, pair of column blocks p and nbw - 1 - p: nbw + 1 chunk products whatever p), a contiguous share per
//               wave (the row's regime and slots once per row); 8 x rows per workgroup (a tile's x rows are cut into chunks: blockIdx.x, so that
//               a tile's chunks spread over the XCDs);
//   registers   <= 128 VGPRs, 16 waves per CU: one wave issues a 64-bit VALU operation every 8 cycles, four per SIMD
//               every 4.9 (profiles/r03/microbench_int64.txt) — the first, 8-wide version of this kernel (168 VGPRs, two
//               waves per SIMD) stood at 23-38 % of the issue roof;
//   order       chunks q ascending, s ascending inside: ascending j for every output, first term a plain product —
//               the same operations on the same values as rb_sums / inner_sum => the same bits;
//   regimes     positive / finite / general from the operands (the x row | the tile's y rows, tested while they are read),
//               validated on the finished sums and recomputed with the general multiply-add when a lane fails, as rb_row.
// Phase 2 (k_pair_collect) is one workgroup per output row, a thread per column: it adds the row's terms in the
// reference's order (outer axes lexicographic ascending, mt:984-1012) with sixteen loads in flight — a contiguous stream:
// the slots are ordered by OUTPUT row (row-major) and inside a row by the reference's term order; phase 1 computes a lane's
// slot from closed-form per-axis prefix sums of the term counts.  A slot is the row sum's n2 intervals, (lo, hi)
// interleaved: 4.4 GB at 64^3, written and read once (1 ms each way against the 5 ms saved).
// MI355X, positive / mixed-sign data (profiles/r04/interval_product.txt): 32^3 1.23 / 2.24 -> 0.22 / 0.39 ms, 48^3 3.03 / 5.44
// -> 1.19 / 2.50, 64^3 10.5 / 19.3 -> 5.0 / 10.6, 72^3 24.6 / 49.1 -> 9.0 / 19.6, 80^3 28.5 / 59.3 -> 15.7 / 36.3.
struct PairArgs {
    unsigned xU, x0, x1, yU, y0, y1, zU, z0, z1;  // canonical outer extents (rank 2: U = axis 0 = 1; rank 3: U = 1)
    unsigned n2, nx2, n8, nb, nxc;                // row lengths, padded row length, column blocks, x chunks
    unsigned tsh;                                 // lane tile: T1 = 1 << tsh rows along axis 1, T0 = 64 >> tsh along axis 0
    unsigned NW, xch;                             // waves and x rows of a phase-1 workgroup
    unsigned tiles0, tiles1;                      // y tiles along the two lane axes
    unsigned pitch;                               // LDS row pitch in doubles ((lo, hi) interleaved; pitch / 2 odd)
    unsigned long long S0, S1;                    // terms summed over all k0 / all k1
    // Bounded workspace (round 5): a launch covers the output slabs [klo, khi) of the LEADING outer axis only, and its slots
    // start at slot_base.  band 2: the leading axis is U (rank 4) — the x rows' ju is restricted; band 1: it is axis 0 (rank 3),
    // a lane axis — the y tile is the WINDOW d0 = klo - j0 + l0 of T0 = khi - klo rows under one j0 (every lane of it pairs with
    // an output slab of the range), a workgroup's x rows share that j0.  band 0: the whole product (klo = 0).
    unsigned band, klo, khi;
    unsigned long long slot_base;
    // Phase-1 grid: one workgroup per VALID (x rows, y tile) combination, in one dimension (see k_pair_sums): c1tot chunks of
    // x rows over all tiles of axis 1, s0tot (j0, tile of axis 0) combinations (band 1: j0 from at_lo up, one window each)
    unsigned c1tot, s0tot, at_lo;
    unsigned kuf;  // band 1: the slab of U the range lies in (rank 3: 0) — the x rows' ju = kuf - ud for every y slab ud that has one
};
// terms of output index k on one axis: j in [max(0, k + 1 - ny), min(k + 1, nx)), and the number of terms of all k' < k
__host__ __device__ inline unsigned pair_lo(unsigned k, unsigned ny) { return k + 1 > ny ? k + 1 - ny : 0u; }
__host__ __device__ inline unsigned pair_cnt(unsigned k, unsigned nx, unsigned ny) {
    const unsigned lo = pair_lo(k, ny), hi = k + 1 < nx ? k + 1 : nx;
    return hi > lo ? hi - lo : 0u;
}
__host__ __device__ inline unsigned long long pair_pre(unsigned k, unsigned nx, unsigned ny) {  // sum_{i < k} pair_cnt(i) (for x, y <= z: no empty k)
    const unsigned long long kk = k;
    const unsigned long long A = k <= nx ? kk * (kk + 1) / 2 : (unsigned long long)nx * (nx + 1) / 2 + (kk - nx) * nx;  // sum min(i + 1, nx)
    const unsigned long long B = k > ny ? (kk - ny) * (kk - ny + 1) / 2 : 0ull;                                   // sum max(0, i + 1 - ny)
    return A - B;
}
// slot of the row sum x row (ju, j0, j1) (*) y row (k - j): output rows in row-major order, each row's terms in the reference's
// order (outer axes lexicographic ascending j) — phase 2 reads every output row's slots as ONE contiguous stream
__device__ __forceinline__ unsigned long long pair_slot(const PairArgs& g, unsigned ju, unsigned j0, unsigned j1, unsigned ku, unsigned k0,
                                                        unsigned k1) {
    const unsigned cU = pair_cnt(ku, g.xU, g.yU), c0 = pair_cnt(k0, g.x0, g.y0), c1 = pair_cnt(k1, g.x1, g.y1);
    const unsigned long long base = pair_pre(ku, g.xU, g.yU) * g.S0 * g.S1 + (unsigned long long)cU * (pair_pre(k0, g.x0, g.y0) * g.S1 + (unsigned long long)c0 * pair_pre(k1, g.x1, g.y1));
    return base + ((unsigned long long)(ju - pair_lo(ku, g.yU)) * c0 + (j0 - pair_lo(k0, g.y0))) * c1 + (j1 - pair_lo(k1, g.y1));
}

// first output row (row-major over (ku, k0, k1)) of a launch's slab range
__device__ __forceinline__ unsigned long long pair_row_base(const PairArgs& g) {
    return g.band == 2 ? (unsigned long long)g.klo * g.z0 * g.z1 : (g.band == 1 ? ((unsigned long long)g.kuf * g.z0 + g.klo) * g.z1 : 0ull);
}

typedef const double __attribute__((address_space(4))) * pair_cptr_t;  // wave-uniform, read-only: scalar loads

// One CW x CW chunk product: x[CW q + s] (lo / hi from SGPRs) against the window (cur: y chunk cb - q, prev: the chunk below).
// FIRST: chunk 0, whose s = 0 term starts every sum; TRI: the diagonal chunk (terms with s <= r only); PART: the x row may end
// inside this chunk (slim = elements left; uniform tests per s).  REG 1: positive regime, 2: finite regime (gft_elem.hpp).
// CW = 4: a lane holds 4 outputs and two 4-wide windows — ~60 VGPRs, so that 16 waves fit a CU: a single wave issues a 64-bit
// VALU operation only every 8 cycles, two waves per SIMD reach 5.6, four 4.9 (profiles/r03/microbench_int64.txt), and the
// 8-wide form of this kernel (168 VGPRs, 2 waves per SIMD) stood at 23-38 % of the issue roof.
template <class E>
__device__ __forceinline__ typename E::V pair_mk(double lo, double hi) {  // an element from its plane values (f64: one plane)
    if constexpr (E::W == 2) return Iv{lo, hi};
    else return lo;
}
// REG 1: positive regime, 2: finite regime (gft_elem.hpp), 3: plain separate multiply and add (f64: E::mac; the first term is
// 0 + x y as the reference's sum starts from zero, which turns a -0 product into +0)
template <class E, int REG, int CW, bool FIRST, bool TRI, bool PART>
__device__ __forceinline__ void pair_chunk(typename E::V (&acc)[CW], const double (&xlo)[CW], const double (&xhi)[CW], const typename E::V (&cur)[CW],
                                           const typename E::V (&prev)[CW], unsigned slim) {
    typedef typename E::V V;
#pragma unroll
    for (int s = 0; s < CW; ++s) {
        if (!PART || (unsigned)s < slim) {
            const V xv = pair_mk<E>(xlo[s], xhi[s]);
#pragma unroll
            for (int r = 0; r < CW; ++r) {
                if (TRI && r < s) continue;
                const V yv = (r - s >= 0) ? cur[(r - s) % CW] : prev[(CW + r - s) % CW];
                if constexpr (REG == 1) acc[r] = (FIRST && s == 0) ? E::mul_pos(xv, yv) : E::mac_pos_unchecked(acc[r], xv, yv);
                else if constexpr (REG == 2) acc[r] = (FIRST && s == 0) ? E::mul_fin(xv, yv) : E::mac_fin(acc[r], xv, yv);
                else acc[r] = E::mac((FIRST && s == 0) ? E::zero() : acc[r], xv, yv);
            }
        }
    }
}
template <class E, int CW>
__device__ __forceinline__ void pair_ldx(double (&lo)[CW], double (&hi)[CW], pair_cptr_t xl, pair_cptr_t xh, unsigned q) {
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        lo[i] = xl[CW * q + i];
        hi[i] = E::W == 2 ? xh[CW * q + i] : 0.0;
    }
}
template <class E, int CW>
__device__ __forceinline__ void pair_ldw(typename E::V (&w)[CW], const double* yrow, unsigned chunk) {  // CW elements, 16-byte reads
    const double2* q = reinterpret_cast<const double2*>(__builtin_assume_aligned(yrow + E::W * CW * chunk, 16));
    if constexpr (E::W == 2) {
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            const double2 v = q[i];
            w[i] = Iv{v.x, v.y};
        }
    } else {
#pragma unroll
        for (int i = 0; i < CW / 2; ++i) {
            const double2 v = q[i];
            w[2 * i] = v.x;
            w[2 * i + 1] = v.y;
        }
    }
}
// the CW outputs of column block cb of one row sum: x chunks q = 0 .. min(cb, nxc) - 1 against full windows (the window
// alternates between (A, B) and (B, A): it slides without register moves), then the diagonal chunk if x reaches it
template <class E, int REG, int CW>
__device__ __forceinline__ void pair_block(typename E::V (&acc)[CW], unsigned cb, pair_cptr_t xl, pair_cptr_t xh, const double* yrow, unsigned nxc,
                                           unsigned nx2) {
    typedef typename E::V V;
    V A[CW], B[CW];
    double xlo[CW], xhi[CW];
    pair_ldw<E, CW>(A, yrow, cb);
    const unsigned qfull = cb < nxc ? cb : nxc;
    const unsigned qwhole = nx2 / CW < qfull ? nx2 / CW : qfull;  // chunks x spans completely
    if (qfull == 0) {  // cb == 0: the diagonal chunk is chunk 0
        pair_ldx<E, CW>(xlo, xhi, xl, xh, 0);
        pair_chunk<E, REG, CW, true, true, true>(acc, xlo, xhi, A, A, nx2);
        return;
    }
    pair_ldw<E, CW>(B, yrow, cb - 1);
    pair_ldx<E, CW>(xlo, xhi, xl, xh, 0);
    pair_chunk<E, REG, CW, true, false, true>(acc, xlo, xhi, A, B, nx2);
    unsigned q = 1;
    for (; q + 2 <= qwhole; q += 2) {  // (the current window is in B here)
        pair_ldw<E, CW>(A, yrow, cb - q - 1);
        pair_ldx<E, CW>(xlo, xhi, xl, xh, q);
        pair_chunk<E, REG, CW, false, false, false>(acc, xlo, xhi, B, A, CW);
        pair_ldw<E, CW>(B, yrow, cb - q - 2);
        pair_ldx<E, CW>(xlo, xhi, xl, xh, q + 1);
        pair_chunk<E, REG, CW, false, false, false>(acc, xlo, xhi, A, B, CW);
    }
    for (; q < qfull; ++q) {  // at most one whole chunk and x's partial one: the window moves by copies here
#pragma unroll
        for (int i = 0; i < CW; ++i) A[i] = B[i];
        pair_ldw<E, CW>(B, yrow, cb - q - 1);
        pair_ldx<E, CW>(xlo, xhi, xl, xh, q);
        pair_chunk<E, REG, CW, false, false, true>(acc, xlo, xhi, A, B, nx2 - CW * q);
    }
    if (cb < nxc) {  // the diagonal chunk: the current window is y chunk 0, in B
        pair_ldx<E, CW>(xlo, xhi, xl, xh, cb);
        pair_chunk<E, REG, CW, false, true, true>(acc, xlo, xhi, B, B, nx2 - CW * cb);
    }
}
// the same CW outputs with the general multiply-add, as a plain loop per output (rare: operands with exact 0 / 1 / inf / NaN,
// sums that left their regime — and the general multiply-add unrolled would be most of the kernel's code)
template <class E, int CW>
__device__ __forceinline__ void pair_block_general(typename E::V (&acc)[CW], unsigned cb, pair_cptr_t xl, pair_cptr_t xh, const double* yrow, unsigned nx2) {
    typedef typename E::V V;
#pragma unroll
    for (int r = 0; r < CW; ++r) {
        const unsigned c = CW * cb + r;
        V s = E::mac(E::zero(), Iv{xl[0], xh[0]}, Iv{yrow[2 * c], yrow[2 * c + 1]});
        const unsigned jend = c + 1 < nx2 ? c + 1 : nx2;
#pragma unroll 1
        for (unsigned j = 1; j < jend; ++j) s = E::mac(s, Iv{xl[j], xh[j]}, Iv{yrow[2 * (c - j)], yrow[2 * (c - j) + 1]});
        acc[r] = s;
    }
}

// column block width: intervals 4 (see pair_chunk), f64 8 (half the registers per element: fewer window / x loads per multiply-add)
template <class E>
struct PairCW { static constexpr int value = E::W == 2 ? 4 : 8; };

template <class E>
__global__ void __launch_bounds__(1024) k_pair_sums(const double* __restrict__ x, size_t xp, const double* __restrict__ y, size_t yp,
                                                   double* __restrict__ ws, PairArgs g) {
    typedef typename E::V V;
    constexpr int CW = PairCW<E>::value;
    constexpr unsigned W = E::W;
    extern __shared__ __align__(16) double smem[];
    __shared__ unsigned s_tileflag;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned nwaves = blockDim.x >> 6;
    const unsigned T1 = 1u << g.tsh, T0 = 64u >> g.tsh;
    // The grid is ONE-dimensional and holds only the valid (x rows, y tile) combinations, every one the same work: workgroup L
    // is chunk r of the x rows (ju, j0, xch consecutive j1) against tile (ud, at, bt).  (Round 4 launched chunks x tiles and let
    // the combinations outside the triangle return at once — 44 % of the grid; the dispatcher deals workgroups to the XCDs and
    // their shader engines by index, so where valid and invalid ones alternate with a short period — the slab ranges: 8 chunks
    // per tile, the first 8 - bt valid — some engines got only valid ones and the launch waited for them: 1.5x at 64^3,
    // profiles/r05/interval_pairs_bounded.txt.)
    const unsigned long long L = blockIdx.x;
    unsigned r = (unsigned)(L % g.c1tot);
    const unsigned long long q = L / g.c1tot;
    unsigned bt = 0, n1 = 0;
    for (;; ++bt) {  // (uniform; r < c1tot = the sum of the tiles' chunk counts — the loops below end at their last tile whatever the counts say)
        n1 = g.z1 - T1 * bt < g.x1 ? g.z1 - T1 * bt : g.x1;
        const unsigned nc = (n1 + g.xch - 1) / g.xch;
        if (r < nc || bt + 1 >= g.tiles1) break;
        r -= nc;
    }
    unsigned at = 0, ud = 0, j0 = 0, ju = 0;
    if (g.band == 1) {  // `at` is the x rows' j0; one y slab ud per x slab ju = kuf - ud
        const unsigned nat = g.tiles0 - g.at_lo;
        at = g.at_lo + (unsigned)(q % nat);
        j0 = at;
        ud = (g.kuf + 1 > g.xU ? g.kuf + 1 - g.xU : 0u) + (unsigned)(q / nat);
        ju = g.kuf - ud;
    } else {
        unsigned q0 = (unsigned)(q % g.s0tot);
        unsigned long long qu = q / g.s0tot;
        for (;; ++at) {
            const unsigned n0 = g.z0 - T0 * at < g.x0 ? g.z0 - T0 * at : g.x0;
            if (q0 < n0 || at + 1 >= g.tiles0) break;
            q0 -= n0;
        }
        j0 = q0;
        for (;; ++ud) {  // x terms ju of the leading axis this tile's ud pairs with (band 2: ju + ud in [klo, khi))
            unsigned nU = g.zU - ud < g.xU ? g.zU - ud : g.xU, ju_lo = 0;
            if (g.band == 2) {
                ju_lo = g.klo > ud ? g.klo - ud : 0u;
                const unsigned hi = g.khi > ud ? (g.khi - ud < nU ? g.khi - ud : nU) : 0u;
                nU = hi > ju_lo ? hi - ju_lo : 0u;
            }
            if (qu < nU || ud + 1 >= g.yU) {
                ju = ju_lo + (unsigned)qu;
                break;
            }
            qu -= nU;
        }
    }
    const int d0b = g.band == 1 ? (int)g.klo - (int)at : (int)(T0 * at);
    const unsigned d1b = T1 * bt;
    const unsigned j1_lo = r * g.xch;
    const unsigned rows_here = j1_lo + g.xch < n1 ? g.xch : n1 - j1_lo;
    // ---- stage the tile's 64 y rows, (lo, hi) interleaved, zero beyond the row / for rows outside y
    if (tid == 0) s_tileflag = 0;
    __syncthreads();
    {
        unsigned fl = 0;
        for (unsigned i = tid; i < 64u * g.n8; i += blockDim.x) {
            const unsigned r = i / g.n8, cc = i - r * g.n8;
            const unsigned d0 = (unsigned)(d0b + (int)(r >> g.tsh)), d1 = d1b + (r & (T1 - 1u));  // (a window row below 0 wraps: fails the test)
            double lo = 0.0, hi = 0.0;
            if (d0 < g.y0 && d1 < g.y1 && cc < g.n2) {
                const size_t row = ((size_t)ud * g.y0 + d0) * g.y1 + d1;
                lo = y[row * g.n2 + cc];
                if constexpr (W == 2) {
                    hi = y[yp + row * g.n2 + cc];
                    const Iv e = Iv{lo, hi};  // the tile's regime: the worst of its elements (as k_row_flags would say)
                    fl |= (E::pos_ok(e) ? 0u : 1u) | (E::fin_ok(e) ? 0u : 2u);
                }
            }
            if constexpr (W == 2) reinterpret_cast<double2*>(smem + (size_t)r * g.pitch)[cc] = double2{lo, hi};
            else smem[(size_t)r * g.pitch + cc] = lo;
        }
        if (fl) atomicOr(&s_tileflag, fl);
    }
    __syncthreads();
    const unsigned tileflag = s_tileflag;
    const unsigned l0 = lane >> g.tsh, l1 = lane & (T1 - 1u);
    const unsigned d0 = (unsigned)(d0b + (int)l0), d1 = d1b + l1;
    const bool row_ok = d0 < g.y0 && d1 < g.y1 && (g.band != 1 || l0 < g.khi - g.klo);
    const double* yrow = smem + (size_t)lane * g.pitch;
    // ---- tasks (x row, pair of column blocks p and nbw - 1 - p: p + 1 and nbw - p chunk products, the same sum for every p):
    // every wave takes a contiguous share of them, so what belongs to the x row — its regime, the lanes' slots — is worked
    // out once per row and wave (with 8 rows and 8 waves: once per row), not once per task
    const unsigned nbw = g.n8 / CW, nxc = (g.nx2 + CW - 1) / CW, npairs = (nbw + 1) / 2;
    const unsigned ntasks = rows_here * npairs;
    const unsigned t_lo = (unsigned)((unsigned long long)ntasks * wave / nwaves), t_hi = (unsigned)((unsigned long long)ntasks * (wave + 1) / nwaves);
    unsigned cur_row = 0xffffffffu;
    pair_cptr_t xl = nullptr, xh = nullptr;
    bool lane_ok = false;
    int regime = 0;
    double* dst = ws;
    for (unsigned t = t_lo; t < t_hi; ++t) {
        const unsigned row = t / npairs, pr = t - row * npairs;
        if (row != cur_row) {  // (uniform)
            cur_row = row;
            const unsigned j1 = j1_lo + row;
            const size_t arow = ((size_t)ju * g.x0 + j0) * g.x1 + j1;
            xl = (pair_cptr_t)(x + arow * g.nx2);
            xh = (pair_cptr_t)(x + xp + arow * g.nx2);
            lane_ok = row_ok && j0 + d0 < g.z0 && j1 + d1 < g.z1;
            unsigned f = tileflag;
            if constexpr (E::HAS_POS) {  // the x row's regime, from vector loads of the row (it is read through the scalar cache below)
                bool np = false, nf = false;
                for (unsigned i = lane; i < g.nx2; i += 64) {
                    const Iv e = Iv{x[arow * g.nx2 + i], x[xp + arow * g.nx2 + i]};
                    np = np || !E::pos_ok(e);
                    nf = nf || !E::fin_ok(e);
                }
                f |= (any_lane(np) ? 1u : 0u) | (any_lane(nf) ? 2u : 0u);
            }
            regime = (f & 1u) == 0u ? 1 : ((f & 2u) == 0u ? 2 : 0);
            dst = ws + (size_t)(lane_ok ? pair_slot(g, ju, j0, j1, ju + ud, j0 + d0, j1 + d1) - g.slot_base : 0ull) * g.n2 * W;
        }
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const unsigned cb = half == 0 ? pr : nbw - 1 - pr;
            if (half == 1 && cb <= pr) continue;  // (uniform: the middle block of an odd count is its own pair)
            V acc[CW];
#pragma unroll
            for (int r = 0; r < CW; ++r) acc[r] = E::zero();
            if constexpr (E::HAS_POS) {
                bool redo = true;
                if (regime == 1) {
                    pair_block<E, 1, CW>(acc, cb, xl, xh, yrow, nxc, g.nx2);
                    bool bad = false;
#pragma unroll
                    for (int r = 0; r < CW; ++r) bad = bad || (CW * cb + r < g.n2 && (!E::pos_first_ok(acc[r]) || !E::pos_result_ok(acc[r])));
                    redo = any_lane(lane_ok && bad);
                } else if (regime == 2) {
                    pair_block<E, 2, CW>(acc, cb, xl, xh, yrow, nxc, g.nx2);
                    bool bad = false;
#pragma unroll
                    for (int r = 0; r < CW; ++r) bad = bad || (CW * cb + r < g.n2 && !E::fin_result_ok(acc[r]));
                    redo = any_lane(lane_ok && bad);
                }
                if (redo) pair_block_general<E, CW>(acc, cb, xl, xh, yrow, g.nx2);
            } else {
                pair_block<E, 3, CW>(acc, cb, xl, xh, yrow, nxc, g.nx2);
            }
            if (lane_ok) {
                if constexpr (W == 2) {
                    double2* d = reinterpret_cast<double2*>(dst) + CW * cb;
#pragma unroll
                    for (int r = 0; r < CW; ++r)
                        if (CW * cb + r < g.n2) d[r] = double2{acc[r].lo, acc[r].hi};
                } else {
                    double* d = dst + CW * cb;
#pragma unroll
                    for (int r = 0; r < CW; ++r)
                        if (CW * cb + r < g.n2) d[r] = acc[r];
                }
            }
        }
    }
}

// phase 2: the terms of one output row are one contiguous stream in the reference's order; thread = column, D terms in
// flight (the heaviest row of 64^3 has 4096 terms of 1 KB: its chain of loads and dependent additions is what the launch waits for).
// The ordered sum of the n terms p[0], p[pitch], .. onto acc.  REG 1 (intervals): every term is tested for 0 < lo <= hi and added
// with the positive regime's integer steps (gft_elem.hpp add_pos: two dependent instructions per bound instead of the general
// add's nine — the chain, not the bandwidth, bounds a slab range's launch); REG 2: every term is tested for [0,0] (the add's only
// short-circuit) and added with the finite regime's unguarded outward step (five), whose NaN pattern marks a sum that met an
// infinity; `bad` says that some term failed its test, the caller redoes the row with the general add (REG 0).
// (Rows of <= 32 elements with the wave's idle half loading every second term and handing it down through __shfl_down:
// measured, SLOWER — 32^4 84 -> 95 ms, 32^3 0.147 -> 0.166: the cross-lane reads wait on the same counter as the chain's loads.)
template <class E, int REG, class Raw>
__device__ __forceinline__ typename E::V pair_sum_terms(const Raw* __restrict__ p, size_t pitch, unsigned long long n, typename E::V acc, bool& bad) {
    typedef typename E::V V;
    if (n == 0) return acc;
    auto step = [&](V a, const Raw& r) -> V {
        if constexpr (E::W == 2) {
            const Iv t = Iv{r.x, r.y};
            if constexpr (REG == 1) {
                bad = bad || !(t.lo > 0.0 && t.hi >= t.lo);
                return E::add_pos(a, t);
            } else if constexpr (REG == 2) {
                bad = bad || (t.lo == 0.0 && t.hi == 0.0);
                return E::widen_fin(a.lo + t.lo, a.hi + t.hi);
            } else {
                return E::add(a, t);
            }
        } else {
            return E::add(a, r);
        }
    };
    constexpr int D = 32;
    Raw buf[D];
    // (no conditional loads in the steady state: hipcc waits for vmcnt(0) at every basic-block boundary, and a load
    // under `if` is a block of its own — one load in flight instead of D)
#pragma unroll
    for (int u = 0; u < D; ++u) buf[u] = p[(size_t)((unsigned long long)u < n ? u : n - 1) * pitch];
    unsigned long long i0 = 0;
    for (; i0 + 2 * D <= n; i0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            acc = step(acc, buf[u]);
            buf[u] = p[(size_t)(i0 + u + D) * pitch];
        }
    }
    // the last D .. 2 D - 1 terms: D in the buffers, the rest loaded with clamped indices
    Raw last[D];
#pragma unroll
    for (int u = 0; u < D; ++u) last[u] = p[(size_t)(i0 + D + u < n ? i0 + D + u : n - 1) * pitch];
#pragma unroll
    for (int u = 0; u < D; ++u)
        if (i0 + u < n) acc = step(acc, buf[u]);
#pragma unroll
    for (int u = 0; u < D; ++u)
        if (i0 + D + u < n) acc = step(acc, last[u]);
    return acc;
}

template <class E>
__global__ void __launch_bounds__(128) k_pair_collect(const double* __restrict__ ws, double* __restrict__ z, size_t zp, PairArgs g) {
    typedef typename E::V V;
    typedef typename std::conditional<E::W == 2, double2, double>::type Raw;  // a stored element
    const unsigned c = blockIdx.y * blockDim.x + threadIdx.x;
    const bool act = c < g.n2;
    const unsigned cc = act ? c : g.n2 - 1;  // (threads beyond the row follow its last column: the wave decides the regime together)
    unsigned long long rr = (unsigned long long)(gridDim.x - 1 - blockIdx.x) + pair_row_base(g);  // heaviest rows first; the launch's slab range
    const unsigned k1 = (unsigned)(rr % g.z1);
    rr /= g.z1;
    const unsigned k0 = (unsigned)(rr % g.z0), ku = (unsigned)(rr / g.z0);
    const unsigned long long n = (unsigned long long)pair_cnt(ku, g.xU, g.yU) * pair_cnt(k0, g.x0, g.y0) * pair_cnt(k1, g.x1, g.y1);
    V acc = E::zero();
    if (n > 0) {
        const Raw* p = reinterpret_cast<const Raw*>(ws + (size_t)(pair_slot(g, pair_lo(ku, g.yU), pair_lo(k0, g.y0), pair_lo(k1, g.y1), ku, k0, k1) - g.slot_base) * g.n2 * E::W) + cc;
        const size_t pitch = g.n2;  // elements per term
        bool bad = false;
        if constexpr (E::HAS_POS) {
            // the reference's sum starts from [0,0], whose add returns the first term itself (interval.rs:126-139); where that term
            // is positive in every column the row is tried in the positive regime — same operations on the same values while
            // every term stays 0 < lo <= hi and no bound overflows (inc_pos(inf) is a NaN pattern and stays one), else redone
            // (a first term that is not [0,0] in any column: the finite regime — the sum is never [0,0] again, widened intervals are not points)
            const Raw t0 = p[0];
            const bool pos0 = t0.x > 0.0 && t0.y >= t0.x, nz0 = !(t0.x == 0.0 && t0.y == 0.0);
            bool done = false;
            if (!any_lane(!pos0)) {
                acc = pair_sum_terms<E, 1>(p + pitch, pitch, n - 1, Iv{t0.x, t0.y}, bad);
                bad = bad || !E::pos_result_ok(acc);
                done = !any_lane(bad);
            } else if (!any_lane(!nz0)) {
                acc = pair_sum_terms<E, 2>(p + pitch, pitch, n - 1, Iv{t0.x, t0.y}, bad);
                bad = bad || !E::fin_result_ok(acc);
                done = !any_lane(bad);
            }
            if (!done) acc = pair_sum_terms<E, 0>(p, pitch, n, E::zero(), bad);
        } else {
            acc = pair_sum_terms<E, 0>(p, pitch, n, E::zero(), bad);
        }
    }
    if (act) E::st(z, zp, (((size_t)ku * g.z0 + k0) * g.z1 + k1) * g.n2 + c, acc);
}

// f64 rows of even length: two columns per thread (16-byte loads, as the interval form has anyway)
__global__ void __launch_bounds__(128) k_pair_collect2_f64(const double* __restrict__ ws, double* __restrict__ z, PairArgs g) {
    const unsigned c = 2 * (blockIdx.y * blockDim.x + threadIdx.x);
    unsigned long long rr = (unsigned long long)(gridDim.x - 1 - blockIdx.x) + pair_row_base(g);  // heaviest rows first; the launch's slab range
    const unsigned k1 = (unsigned)(rr % g.z1);
    rr /= g.z1;
    const unsigned k0 = (unsigned)(rr % g.z0), ku = (unsigned)(rr / g.z0);
    const unsigned long long n = (unsigned long long)pair_cnt(ku, g.xU, g.yU) * pair_cnt(k0, g.x0, g.y0) * pair_cnt(k1, g.x1, g.y1);
    double a0 = 0.0, a1 = 0.0;
    if (c >= g.n2) return;
    if (n > 0) {
        const double2* p = reinterpret_cast<const double2*>(ws + (size_t)(pair_slot(g, pair_lo(ku, g.yU), pair_lo(k0, g.y0), pair_lo(k1, g.y1), ku, k0, k1) - g.slot_base) * g.n2 + c);
        const size_t pitch = g.n2 / 2;  // double2 per term
        constexpr int D = 16;
        double2 buf[D];
#pragma unroll
        for (int u = 0; u < D; ++u) buf[u] = p[(size_t)((unsigned long long)u < n ? u : n - 1) * pitch];
        unsigned long long i0 = 0;
        for (; i0 + 2 * D <= n; i0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                a0 = a0 + buf[u].x;
                a1 = a1 + buf[u].y;
                buf[u] = p[(size_t)(i0 + u + D) * pitch];
            }
        }
        double2 last[D];
#pragma unroll
        for (int u = 0; u < D; ++u) last[u] = p[(size_t)(i0 + D + u < n ? i0 + D + u : n - 1) * pitch];
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (i0 + u < n) {
                a0 = a0 + buf[u].x;
                a1 = a1 + buf[u].y;
            }
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (i0 + D + u < n) {
                a0 = a0 + last[u].x;
                a1 = a1 + last[u].y;
            }
    }
    double* zr = z + (((size_t)ku * g.z0 + k0) * g.z1 + k1) * g.n2 + c;
    zr[0] = a0;
    zr[1] = a1;
}

// per-stream scratch for the row flags (grow-only; a stream's launches are ordered, so one buffer per stream is enough)
struct RbScratch {
    unsigned char* p = nullptr;
    size_t bytes = 0;
};
static std::map<hipStream_t, RbScratch>& rb_scratch() {
    static std::map<hipStream_t, RbScratch> m;
    return m;
}
struct PairWs {  // per-stream workspace of the row-pair form (grow-only up to the cap)
    double* p = nullptr;
    size_t bytes = 0;
};
static std::map<hipStream_t, PairWs>& pair_ws() {
    static std::map<hipStream_t, PairWs> m;
    return m;
}
// bytes of row sums the row-pair form may hold AT A TIME (2 GiB; round 4 held a whole product's: up to 24 GiB).  A product whose
// row sums exceed it runs in slab ranges of its leading outer axis (PairArgs::band), each range through both phases from the
// same workspace; per output the terms still arrive in the reference's order, so the bits do not depend on the cut.
static size_t rb_pairs_cap = (size_t)2048 << 20;
static int rb_pairs_lanes = -1;  // slab ranges on two lanes when half the cap leaves the ranges as they are (56^3 .. 72^3)
void staged_set_rb_pairs_lanes(double v) { rb_pairs_lanes = v < 0 ? -1 : (v >= 1.0 ? 1 : 0); }  // "conv_rb_pairs_lanes" (tests: both forms bit for bit)
void staged_set_rb_pairs_cap(double bytes) { rb_pairs_cap = bytes >= 1.0 ? (size_t)bytes : ((size_t)2048 << 20); }  // "conv_rb_pairs_cap"
size_t staged_scratch_bytes() {  // what the grow-only workspaces hold right now (gft_pool_stats counts it)
    size_t n = 0;
    for (auto& kv : rb_scratch()) n += kv.second.bytes;
    for (auto& kv : pair_ws()) n += kv.second.bytes;
    return n;
}
// row-pair form (k_pair_sums + k_pair_collect): 0 never, 1 products of [rb_pairs_min, ..) multiply-adds whose row sums fit the
// workspace cap, 2 whenever it applies (tests)
static int rb_pairs_mode = 1;
static const double rb_pairs_min = 3.0e5;  // rank >= 3 (profiles/r04/interval_shapes.txt: 12^3 = 5e5 multiply-adds 0.098 -> 0.036 ms, 16^3 0.20 -> 0.05)
static const double rb_pairs_min_rank2 = 1.0e6;  // (64^2 = 4e6 multiply-adds 0.064 -> 0.033 ms, 128^2 0.31 -> 0.07)
void staged_set_rb_pairs(double v) { rb_pairs_mode = v < 0.0 ? 1 : (int)v; }  // "conv_rb_pairs" (negative: back to the default)
// measured crossover against k_conv_staged between 64^3 and 80^3 (profiles/r03/interval_product.txt)
static double rb_min_macs = 1.5e10;  // (72^3 = 1.8e10: 29.4 -> 24.5 ms on this kernel; 64^3 = 9e9 stays on k_conv_staged: 10.6 vs 14.1 ms — profiles/r04/interval_rb_crossover.txt)
void staged_set_rb_min_macs(double v) { rb_min_macs = v; }  // "conv_rb_min_macs" (tests; negative = never)
void staged_release_scratch() {
    for (auto& kv : rb_scratch())
        if (kv.second.p) (void)hipFree(kv.second.p);
    rb_scratch().clear();
    for (auto& kv : pair_ws())
        if (kv.second.p) (void)hipFree(kv.second.p);
    pair_ws().clear();
}

// Two lanes for the slab ranges of a product whose row sums exceed the cap (PairArgs::band): each range is a phase-1 launch
// (compute-bound, uneven workgroups) followed by a phase-2 launch (a latency-bound stream per output row) on the SAME
// workspace — on one stream every launch waits for the slowest workgroup of the one before it, 2 x #ranges times per product
// (64^3 in 8 ranges: 7.1 ms against 5.1 ms in one piece).  With the ranges dealt to two streams, each with half the cap as its
// own workspace, one lane's tails and phase 2 run under the other lane's phase 1.  Fork and join are one event each way per
// product (HIP events cost ~20 us a pair, tools/microbench_streams.hip: nothing at these sizes).
struct PairLanes {
    hipStream_t st[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    bool ok = false, tried = false;
};
static PairLanes& pair_lanes() {
    static PairLanes l;
    if (!l.tried) {
        l.tried = true;
        l.ok = hipStreamCreateWithFlags(&l.st[0], hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&l.st[1], hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&l.join[0], hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&l.join[1], hipEventDisableTiming) == hipSuccess;
        if (!l.ok) (void)hipGetLastError();
    }
    return l;
}

// The plain full product of large contiguous interval tensors; false = not this kernel's case (nothing launched).
template <class E>
static bool conv_rows_rb(hipStream_t st, const double* x, size_t xp, const double* y, size_t yp, double* z, size_t zp, const ConvArgs& a,
                         bool pairs_only = false, double pairs_max_macs = 1e300) {
    static const int on = 1;
    const int nd = a.nd;
    if (nd < 2 || nd > 4) return false;
    if (a.accumulate || a.j0_min || a.j0_excl || a.j0_desc || !a.inner_from_zero || a.guard) return false;
    const int P = nd - 2;
    const unsigned n2 = a.zs[nd - 1], nx2 = a.xs[nd - 1];
    if (a.ys[nd - 1] != n2 || n2 < 8 || n2 > 256 || nx2 == 0) return false;
    if (P == 0 && (a.slab_lo != 0 || a.slab_hi != a.zs[0])) return false;
    // contiguous operands and result
    size_t xs_ = 1, ys_ = 1, zs_ = 1;
    for (int ax = nd - 1; ax >= 0; --ax) {
        if (a.xstr[ax] != xs_ || a.ystr[ax] != ys_ || a.zstr[ax] != zs_) return false;
        if (a.xs[ax] == 0 || a.ys[ax] == 0 || a.xs[ax] > a.zs[ax] || a.ys[ax] > a.zs[ax]) return false;
        xs_ *= a.xs[ax];
        ys_ *= a.ys[ax];
        zs_ *= a.zs[ax];
    }
    const size_t xrows = xs_ / nx2, yrows = ys_ / n2;
    // worth it from a few 10^7 multiply-adds (two passes over the operands, a block per output row pair)
    double macs = 1.0;
    for (int ax = 0; ax < nd; ++ax) macs *= 0.5 * (double)a.zs[ax] * (double)std::min(a.xs[ax], a.ys[ax]);
    // ---- row-pair form
    bool pairs_ok = (rb_pairs_mode == 2 || (rb_pairs_mode == 1 && macs >= (nd == 2 ? rb_pairs_min_rank2 : rb_pairs_min))) && macs <= pairs_max_macs && n2 <= (E::W == 2 ? 128u : 256u) && a.slab_lo == 0 && a.slab_hi == a.zs[0];
    for (int ax = 0; ax + 1 < nd; ++ax)
        if (a.zs[ax] > a.xs[ax] + a.ys[ax] - 1) pairs_ok = false;  // (an output row without terms: the slot prefix sums assume none)
    if (!a.operands_slack && nx2 % PairCW<E>::value != 0) pairs_ok = false;  // (the scalar loads of x's last chunk read to the chunk's end: a caller's raw buffer may end with the row)
    if (pairs_ok) {
        PairArgs g;
        std::memset(&g, 0, sizeof(g));
        g.xU = g.x0 = g.x1 = g.yU = g.y0 = g.y1 = g.zU = g.z0 = g.z1 = 1;
        const int no = nd - 1;
        if (no == 1) {
            g.x1 = a.xs[0]; g.y1 = a.ys[0]; g.z1 = a.zs[0];
        } else if (no == 2) {
            g.x0 = a.xs[0]; g.y0 = a.ys[0]; g.z0 = a.zs[0];
            g.x1 = a.xs[1]; g.y1 = a.ys[1]; g.z1 = a.zs[1];
        } else {
            g.xU = a.xs[0]; g.yU = a.ys[0]; g.zU = a.zs[0];
            g.x0 = a.xs[1]; g.y0 = a.ys[1]; g.z0 = a.zs[1];
            g.x1 = a.xs[2]; g.y1 = a.ys[2]; g.z1 = a.zs[2];
        }
        g.n2 = n2;
        g.nx2 = nx2;
        g.n8 = (n2 + 7) / 8 * 8;
        g.nb = g.n8 / 8;
        g.nxc = (nx2 + 7) / 8;
        {   // lane tile: the shape with the fewest lanes outside y's rows (ties: the squarest)
            double best = -1.0;
            for (unsigned tsh = 0; tsh <= 6; ++tsh) {
                const unsigned T1 = 1u << tsh, T0 = 64u >> tsh;
                const double e = (double)g.y0 / ((g.y0 + T0 - 1) / T0 * T0) * (double)g.y1 / ((g.y1 + T1 - 1) / T1 * T1);
                const double pref = e - 1e-6 * (tsh > 3 ? tsh - 3 : 3 - tsh);
                if (pref > best) {
                    best = pref;
                    g.tsh = tsh;
                }
            }
        }
        const unsigned T1 = 1u << g.tsh, T0 = 64u >> g.tsh;
        g.tiles0 = (std::min(g.y0, g.z0) + T0 - 1) / T0;  // (y rows at or beyond z's extent pair with nothing)
        g.tiles1 = (std::min(g.y1, g.z1) + T1 - 1) / T1;
        g.pitch = E::W * g.n8 + 2;
        // 16 waves per CU (the kernel holds <= 128 VGPRs): two workgroups of 8 where two tiles fit the LDS, else one of 16
        g.NW = (size_t)64 * g.pitch * sizeof(double) * 2 + 1024 <= 160 * 1024 ? 8u : 16u;
        static const unsigned nw_env = (unsigned)(0);
        if (nw_env) g.NW = nw_env;
        static const unsigned xch_env = 8;  // x rows per phase-1 workgroup (sweep in profiles/r04/interval_pairs_sweep.txt: 4 .. 12 equal, 32 loses 10 % to the last round of workgroups)
        g.xch = xch_env;
        // (small products: fewer x rows per workgroup until there are ~1000 workgroups — half of the (tile, chunk) grid is
        // outside the triangle)
        while (g.xch > 1 && (unsigned long long)g.yU * g.tiles0 * g.tiles1 * ((xrows + g.xch - 1) / g.xch) < 2048ull) g.xch /= 2;
        g.S0 = pair_pre(g.z0, g.x0, g.y0);  // terms over all k0 / all k1
        g.S1 = pair_pre(g.z1, g.x1, g.y1);
        const unsigned long long slots = pair_pre(g.zU, g.xU, g.yU) * g.S0 * g.S1;
        const unsigned long long row_bytes = (unsigned long long)n2 * E::W * sizeof(double);
        const unsigned long long need = slots * row_bytes;
        // ---- the plan: one launch pair for the whole product, or slab ranges of the leading outer axis that fit the cap
        struct Range {
            unsigned lo, hi, band, ku;  // band 0: the whole product; 2: slabs [lo, hi) of U (rank 4); 1: slabs [lo, hi) of axis 0 inside slab ku of U (rank 3: ku = 0)
            unsigned long long base, slots;
        };
        std::vector<Range> plan;
        bool plan_ok = slots > 0 && zs_ / n2 <= 0x7fffffffull, use_lanes = false;
        if (plan_ok && need <= rb_pairs_cap) {
            plan.push_back(Range{0, 0, 0, 0, 0, slots});
        } else if (plan_ok && no >= 2) {
            // Two lanes (see PairLanes) need two workspaces, so each gets half the cap: taken when that does not make the ranges
            // thinner — then one lane's phase 2 (a few hundred one-wave rows: latency, not bandwidth) runs under the other's phase 1.
            // MI355X, positive / mixed-sign: 56^3 2.88 / 5.92 -> 2.55 / 5.05 ms, 64^3 5.31 / 11.5 -> 4.92 / 10.2, 72^3 9.40 / 21.0 -> 8.84 / 20.5;
            // from 80^3 on half the cap halves the window's height and the lanes lose (88^3 29.9 -> 30.4, 96^3 49.8 -> 54.0): off there.
            const int lanes_on = rb_pairs_lanes;  // 0 never, 1 always, negative: when the ranges stay the same
            // rank 3: ranges of 1, 2, 4 or 8 slabs of axis 0 (a lane axis: the range's height is the window's).  Rank 4: ranges of slabs
            // of U where a slab fits the cap, ranges of axis 0 INSIDE a slab of U where it does not (32^4: the top slab alone is 4.6 GB)
            auto k0_ranges = [&](unsigned ku, size_t range_cap, std::vector<Range>& out) {
                const unsigned long long cU = pair_cnt(ku, g.xU, g.yU), ubase = pair_pre(ku, g.xU, g.yU) * g.S0 * g.S1;
                auto sl = [&](unsigned lo, unsigned hi) { return cU * (pair_pre(hi, g.x0, g.y0) - pair_pre(lo, g.x0, g.y0)) * g.S1; };
                for (unsigned lo = 0; lo < g.z0;) {
                    unsigned h = 0;
                    for (unsigned c = 8; c >= 1; c /= 2)
                        if (lo + c <= g.z0 && sl(lo, lo + c) * row_bytes <= range_cap) {
                            h = c;
                            break;
                        }
                    if (h == 0) return false;  // (one slab of axis 0 alone exceeds the cap: not this form's product)
                    out.push_back(Range{lo, lo + h, 1u, ku, ubase + cU * pair_pre(lo, g.x0, g.y0) * g.S1, sl(lo, lo + h)});
                    lo += h;
                }
                return true;
            };
            auto make_plan = [&](size_t range_cap, std::vector<Range>& out) {
                out.clear();
                if (no == 2) return k0_ranges(0, range_cap, out) && out.size() <= 4096;
                auto sl = [&](unsigned lo, unsigned hi) { return (pair_pre(hi, g.xU, g.yU) - pair_pre(lo, g.xU, g.yU)) * g.S0 * g.S1; };
                for (unsigned lo = 0; lo < g.zU;) {
                    unsigned h = 0;
                    while (lo + h < g.zU && sl(lo, lo + h + 1) * row_bytes <= range_cap) ++h;
                    if (h) {
                        out.push_back(Range{lo, lo + h, 2u, 0u, pair_pre(lo, g.xU, g.yU) * g.S0 * g.S1, sl(lo, lo + h)});
                        lo += h;
                    } else {
                        if (!k0_ranges(lo, range_cap, out)) return false;
                        lo += 1;
                    }
                }
                return out.size() <= 4096;
            };
            plan_ok = make_plan(lanes_on == 1 && pair_lanes().ok ? rb_pairs_cap / 2 : rb_pairs_cap, plan);
            use_lanes = plan_ok && lanes_on == 1 && pair_lanes().ok;
            if (plan_ok && lanes_on < 0 && plan.size() >= 2 && pair_lanes().ok) {
                std::vector<Range> half;
                if (make_plan(rb_pairs_cap / 2, half) && half.size() == plan.size()) {
                    plan.swap(half);
                    use_lanes = true;
                }
            }
        } else
            plan_ok = false;
        if (plan_ok) {
            unsigned long long most = 0;
            for (const Range& r : plan) most = std::max(most, r.slots * row_bytes);
            if (plan.size() < 2) use_lanes = false;
            bool ok = true;
            PairWs* wsv[2] = {nullptr, nullptr};
            {   // the cap bounds what ALL the streams' workspaces hold together: blocks of other streams go when this one's need the room
                hipStream_t mine[2] = {use_lanes ? pair_lanes().st[0] : st, use_lanes ? pair_lanes().st[1] : st};
                unsigned long long others = 0, here = 0;
                for (auto& kv : pair_ws()) {
                    if (kv.first == mine[0] || kv.first == mine[1]) here += std::max<unsigned long long>(kv.second.bytes, most);
                    else others += kv.second.bytes;
                }
                if (!pair_ws().count(mine[0])) here += most;
                if (use_lanes && !pair_ws().count(mine[1])) here += most;
                if (others && others + here > rb_pairs_cap) {
                    launch_drain();
                    for (auto& kv : pair_ws())
                        if (kv.first != mine[0] && kv.first != mine[1] && kv.second.p) {
                            (void)(hipStreamSynchronize)(kv.first);
                            (void)(hipFree)(kv.second.p);
                            kv.second.p = nullptr;
                            kv.second.bytes = 0;
                        }
                }
            }
            for (int l = 0; l < (use_lanes ? 2 : 1) && ok; ++l) {
                PairWs& w = pair_ws()[use_lanes ? pair_lanes().st[l] : st];
                if (w.bytes < most) {
                    launch_drain();  // (nothing queued may still be using the old block)
                    if (w.p) {
                        (void)(hipStreamSynchronize)(use_lanes ? pair_lanes().st[l] : st);
                        (void)(hipFree)(w.p);
                    }
                    w.p = nullptr;
                    w.bytes = 0;
                    if (hipMalloc((void**)&w.p, most) != hipSuccess) {
                        (void)hipGetLastError();
                        w.p = nullptr;
                        ok = false;
                    } else {
                        w.bytes = most;
                    }
                }
                wsv[l] = &w;
            }
            // grid limits (per launch)
            if (ok) {
                const size_t lds = (size_t)64 * g.pitch * sizeof(double);
                static bool attr = false;
                if (!attr) {
                    // (the kernel also has 4 bytes of static LDS: the full 160 KB is refused — and a refused call leaves its error
                    // in this thread's HIP state, where the CALLER's next HIP call finds it)
                    if (hipFuncSetAttribute((const void*)k_pair_sums<E>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) (void)hipGetLastError();
                    attr = true;
                }
                static const unsigned cw_env = (unsigned)(64);
                // (checked for every range BEFORE the first launch: a false return promises that nothing was launched)
                std::vector<PairArgs> gs;
                std::vector<dim3> grids;
                std::vector<unsigned long long> rows;
                for (const Range& r : plan) {
                    PairArgs q = g;
                    const unsigned band = r.band;
                    q.band = band;
                    q.kuf = r.ku;
                    q.klo = r.lo;
                    q.khi = r.hi;
                    q.slot_base = r.base;
                    unsigned long long nrows = zs_ / n2, combos = 0;  // (x rows sharing (ju, j0)) x (tiles over U and axis 0)
                    if (band == 1) {  // window tiles: T0 = the range's height, one per j0; a workgroup's x rows share j0
                        const unsigned h = r.hi - r.lo;
                        unsigned tsh = 6;
                        while ((64u >> tsh) < h) --tsh;
                        q.tsh = tsh;
                        const unsigned T1w = 1u << tsh;
                        q.tiles1 = (std::min(q.y1, q.z1) + T1w - 1) / T1w;
                        q.tiles0 = std::min(q.x0, r.hi);                       // j0 <= k0 < khi
                        q.at_lo = r.lo + 1 > q.y0 ? r.lo + 1 - q.y0 : 0u;      // (a compact y: the window of a lower j0 lies above y's last row)
                        const unsigned ud_lo = r.ku + 1 > q.xU ? r.ku + 1 - q.xU : 0u, ud_hi = std::min(q.yU - 1, r.ku);  // y slabs ud with an x slab ju = ku - ud
                        combos = q.tiles0 > q.at_lo && ud_hi >= ud_lo ? (unsigned long long)(q.tiles0 - q.at_lo) * (ud_hi - ud_lo + 1) : 0u;
                        nrows = (unsigned long long)(r.hi - r.lo) * q.z1;
                    } else {
                        const unsigned T0w = 64u >> q.tsh;
                        unsigned long long s0 = 0, sU = 0;
                        for (unsigned at = 0; at < q.tiles0; ++at) s0 += std::min(q.z0 - T0w * at, q.x0);
                        for (unsigned ud = 0; ud < std::min(q.yU, q.zU); ++ud) {
                            unsigned nU = std::min(q.zU - ud, q.xU), ju_lo = 0;
                            if (band == 2) {
                                ju_lo = r.lo > ud ? r.lo - ud : 0u;
                                const unsigned hi = r.hi > ud ? std::min(r.hi - ud, nU) : 0u;
                                nU = hi > ju_lo ? hi - ju_lo : 0u;
                            }
                            sU += nU;
                        }
                        q.s0tot = (unsigned)s0;
                        combos = s0 * sU;
                        if (band == 2) nrows = (unsigned long long)(r.hi - r.lo) * q.z0 * q.z1;
                    }
                    {
                        const unsigned T1w = 1u << q.tsh;
                        unsigned long long c1 = 0;
                        for (unsigned bt = 0; bt < q.tiles1; ++bt) c1 += (std::min(q.z1 - T1w * bt, q.x1) + q.xch - 1) / q.xch;
                        q.c1tot = (unsigned)c1;
                    }
                    const unsigned long long nwg = combos * q.c1tot;
                    if (nwg == 0 || nwg > 0x7fffffffull || nrows > 0x7fffffffull) ok = false;
                    gs.push_back(q);
                    grids.push_back(dim3((unsigned)nwg));
                    rows.push_back(nrows);
                }
                if (ok) {
                    PairLanes& L = pair_lanes();
                    if (use_lanes) {  // fork: the lanes wait for everything queued on the product's stream (the operands)
                        hipEvent_t ev = L.fork;
                        hipStream_t a0 = L.st[0], a1 = L.st[1];
                        enqueue_task([=] {
                            lq_note((hipEventRecord)(ev, st), nullptr, "hipEventRecord (row-pair lanes, fork)");
                            lq_note((hipStreamWaitEvent)(a0, ev, 0), nullptr, "hipStreamWaitEvent (row-pair lane 0)");
                            lq_note((hipStreamWaitEvent)(a1, ev, 0), nullptr, "hipStreamWaitEvent (row-pair lane 1)");
                        });
                    }
                    // heaviest ranges first (the top slabs), alternating between the lanes
                    for (size_t k = plan.size(); k-- > 0;) {
                        const size_t i = k;
                        const int l = use_lanes ? (int)((plan.size() - 1 - k) & 1u) : 0;
                        hipStream_t ls = use_lanes ? L.st[l] : st;
                        double* wp = wsv[l]->p;
                        const PairArgs& q = gs[i];
                        GFT_LAUNCH(k_pair_sums<E>, grids[i], dim3(q.NW * 64), lds, ls, x, xp, y, yp, wp, q);
                        if (E::W == 1 && n2 >= 64 && n2 % 2 == 0 && !((uintptr_t)z & 15))  // (shorter rows: too few threads per row — 32^3 0.104 -> 0.113 ms; 80^3 7.7 -> 7.1)
                            GFT_LAUNCH(k_pair_collect2_f64, dim3((unsigned)rows[i], (n2 / 2 + 63) / 64), dim3(64), 0, ls, (const double*)wp, z, q);
                        else
                            GFT_LAUNCH(k_pair_collect<E>, dim3((unsigned)rows[i], (n2 + cw_env - 1) / cw_env), dim3(cw_env), 0, ls, (const double*)wp, z, zp, q);
                    }
                    if (use_lanes) {  // join: the product's stream waits for both lanes
                        hipEvent_t e0 = L.join[0], e1 = L.join[1];
                        hipStream_t a0 = L.st[0], a1 = L.st[1];
                        enqueue_task([=] {
                            lq_note((hipEventRecord)(e0, a0), nullptr, "hipEventRecord (row-pair lane 0)");
                            lq_note((hipEventRecord)(e1, a1), nullptr, "hipEventRecord (row-pair lane 1)");
                            lq_note((hipStreamWaitEvent)(st, e0, 0), nullptr, "hipStreamWaitEvent (row-pair lanes, join)");
                            lq_note((hipStreamWaitEvent)(st, e1, 0), nullptr, "hipStreamWaitEvent (row-pair lanes, join)");
                        });
                    }
                    return true;
                }
            }
        }
    }
    if (pairs_only) return false;
    if constexpr (!E::HAS_POS) {
        return false;  // (f64: the row-pair form only)
    } else {
    if (!on || rb_min_macs < 0.0 || macs < rb_min_macs || n2 < 64) return false;  // (rows shorter than 64: the row-pair form only)
    RbArgs g;
    std::memset(&g, 0, sizeof(g));
    g.no = nd - 1;
    g.n_pg = (a.zs[P] + RB_ROWS - 1) / RB_ROWS;
    g.ntiles = (n2 + 15) / 16;
    g.n2 = n2;
    g.nx2 = nx2;
    static const unsigned rg_env = (unsigned)(0);
    g.ntw = (g.ntiles + 1) / 2;
    // groups cut the longest chain of a workgroup (what bounds mid sizes) but leave one workgroup per CU with nothing to
    // overlap its staging with; two groups once the workgroups outnumber the CUs ~6 times (128^3: 257 ms with 2, 299 with 4;
    // 96^3: 77 with 2, 68 with 4)
    unsigned long long nblk = g.n_pg;
    for (int ax = 0; ax < P; ++ax) nblk *= ax == 0 ? (a.slab_hi - a.slab_lo) : a.zs[ax];
    const unsigned want = rg_env ? rg_env : (nblk >= 1536 ? 2u : 4u);
    g.tb = std::max(1u, std::min(want, 16u / g.ntw));
    auto lds_of = [&](unsigned tb) {
        return ((size_t)tb * 2 * n2 + 16 + (size_t)(tb + RB_ROWS - 1) * (2 * nx2 + 2) + (2 * tb + RB_ROWS + 8 + 7) / 8 + (size_t)(tb - 1) * g.ntw * 512) *
               sizeof(double);
    };
    while (g.tb > 1 && lds_of(g.tb) > 150 * 1024) g.tb /= 2;
    if (lds_of(g.tb) > 150 * 1024) return false;
    for (int ax = 0; ax < g.no; ++ax) {
        g.xrs[ax] = a.xstr[ax] / nx2;
        g.yrs[ax] = a.ystr[ax] / n2;
    }
    // (every "not this kernel" exit comes before the first launch: a false return promises that nothing was launched)
    unsigned long long blocks = g.n_pg;
    for (int ax = 0; ax < P; ++ax) blocks *= ax == 0 ? (a.slab_hi - a.slab_lo) : a.zs[ax];
    if (blocks == 0) return true;
    if (blocks > 0x7fffffffULL) return false;
    RbScratch& sc = rb_scratch()[st];
    if (sc.bytes < xrows + yrows) {
        if (sc.p) (void)hipFree(sc.p);
        sc.p = nullptr;
        sc.bytes = 0;
        const size_t want = std::max<size_t>((xrows + yrows) * 2, 1 << 16);
        if (hipMalloc(&sc.p, want) != hipSuccess) {
            (void)hipGetLastError();
            sc.p = nullptr;
            return false;
        }
        sc.bytes = want;
    }
    g.xflags = sc.p;
    g.yflags = sc.p + xrows;
    GFT_LAUNCH(k_row_flags<E>, dim3((unsigned)((xrows + yrows + 3) / 4)), dim3(256), 0, st, x, xp, xrows, nx2, y, yp, yrows, n2, sc.p);
    const unsigned threads = g.ntw * g.tb * 64;
    const size_t lds = lds_of(g.tb);
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        if (hipFuncSetAttribute((const void*)k_conv_rows_rb<E>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set = true;
    }
    GFT_LAUNCH(k_conv_rows_rb<E>, dim3((unsigned)blocks), dim3(threads), lds, st, x, xp, y, yp, z, zp, a, g);
    return true;
    }
}

// The row-pair form alone (k_pair_sums + k_pair_collect) for plain products of at most max_macs multiply-adds; false: not its
// case, nothing launched.  (gft_api.hip asks it first for small f64 products: bit-exact AND faster than the tiled kernel there.)
template <class E>
bool conv_pairs(hipStream_t st, const double* x, size_t xp, const double* y, size_t yp, double* z, size_t zp, const ConvArgs& a, double max_macs) {
    if (a.slab_hi <= a.slab_lo) return false;
    return conv_rows_rb<E>(st, x, xp, y, yp, z, zp, a, true, max_macs);
}
template bool conv_pairs<EF64>(hipStream_t, const double*, size_t, const double*, size_t, double*, size_t, const ConvArgs&, double);
template bool conv_pairs<EIv>(hipStream_t, const double*, size_t, const double*, size_t, double*, size_t, const ConvArgs&, double);

// Returns false (nothing launched) when the shape does not suit the staged kernel; the caller then
// uses conv_naive.  `force`: ignore the "worth it" thresholds (tests).
template <class E>
bool conv_staged(hipStream_t st, const double* x, size_t xp, const double* y, size_t yp, double* z, size_t zp,
                 const ConvArgs& a, bool force) {
    const int nd = a.nd;
    if (nd < 1) return false;
    if (a.slab_hi <= a.slab_lo) return true;  // nothing to do
    if (conv_rows_rb<E>(st, x, xp, y, yp, z, zp, a)) return true;  // (the row-pair form, f64 and interval; the fused rows kernel, interval)
    constexpr size_t LDS_MAX = 160 * 1024;
    const size_t W = E::W;
    StagedArgs g;
    // S = 2 (planes of the last two axes) while the planes are small enough for >= 4 workgroups per CU;
    // larger planes would leave one fat workgroup per CU with a triangular load => rows (S = 1) instead,
    // unless the rows are too short to fill a wave.
    static const int force_s = 0;
    auto fits = [&](int s, size_t budget) {
        unsigned sxa = s == 2 ? a.xs[nd - 2] : 1, sya = s == 2 ? a.ys[nd - 2] : 1;
        size_t xcap = (size_t)sxa * a.xs[nd - 1], ycap = (size_t)sya * (s == 2 ? a.zs[nd - 1] : a.ys[nd - 1]);
        return (xcap + ycap) * 8 * W <= budget;
    };
    for (int i = 0; i < nd; ++i)
        if (a.xs[i] > a.zs[i] || a.ys[i] > a.zs[i]) return false;  // operands are pre-truncated; be safe
    int S = 0;
    if (force_s == 1 || force_s == 2) {
        if (force_s <= nd && fits(force_s, LDS_MAX)) S = force_s;
    } else if (nd >= 2 && fits(2, 40 * 1024)) S = 2;
    else if (a.zs[nd - 1] >= 32 && fits(1, LDS_MAX)) S = 1;
    else if (nd >= 2 && fits(2, LDS_MAX)) S = 2;
    else if (fits(1, LDS_MAX)) S = 1;
    g.batch = 1;
    if (S) {
        g.S = S;
        g.sxa = S == 2 ? a.xs[nd - 2] : 1; g.sya = S == 2 ? a.ys[nd - 2] : 1; g.na = S == 2 ? a.zs[nd - 2] : 1;
        g.sxb = a.xs[nd - 1]; g.syb = a.ys[nd - 1]; g.nb = a.zs[nd - 1];
        g.xcap = g.sxa * g.sxb;
        g.ycap = g.sya * (S == 2 ? g.nb : g.syb);
        if (S == 1 && nd >= 2) {  // rows of the last outer axis, `batch` per barrier pair
            size_t row_bytes = (size_t)(g.sxb + g.syb) * 8 * W;
            size_t b = std::min<size_t>(R_BATCH_MAX, std::max<size_t>(1, (40 * 1024) / std::max<size_t>(row_bytes, 1)));
            b = std::min<size_t>(b, std::max<unsigned>(1u, std::min(a.xs[nd - 2], a.ys[nd - 2])));
            g.batch = (unsigned)b;
            g.xcap *= g.batch;
            g.ycap *= g.batch;
        }
    }
    g.rw = 1;
    if (S == 0) return false;
    g.no = nd - S;
    unsigned long long sub = (unsigned long long)g.na * g.nb, n_outer = 1;
    if (g.no == 0) {  // axis 0 is a staged axis: the slab restricts the subspace range
        unsigned long long rest = sub / a.zs[0];
        g.sub_lo = a.slab_lo * rest;
        g.sub_hi = a.slab_hi * rest;
    } else {
        g.sub_lo = 0;
        g.sub_hi = sub;
        n_outer = a.slab_hi - a.slab_lo;
        for (int ax = 1; ax < g.no; ++ax) n_outer *= a.zs[ax];
    }
    const unsigned long long span = g.sub_hi - g.sub_lo;
    if (span == 0 || n_outer == 0) return true;
    (void)force;  // every supported shape is at least as fast here as on the one-thread-per-output kernel
    size_t lds = (size_t)(g.xcap + g.ycap) * 8 * W;
    unsigned threads = lds <= 40 * 1024 ? 256 : (lds <= 80 * 1024 ? 512 : 1024);
    // do not use more threads than one balanced chunk needs
    unsigned long long chunks = (span + threads - 1) / threads;
    unsigned per = (unsigned)((span + chunks - 1) / chunks);
    unsigned need = (per + 63) / 64 * 64;
    if (need < threads && lds <= 40 * 1024) threads = need;
    chunks = (span + threads - 1) / threads;
    g.chunks = (unsigned)chunks;
    // row groups (see the kernel): as many as the batch has rows and the block has room for
    static const unsigned rw_cap = (unsigned)(1024);
    if (S == 1 && a.inner_from_zero && g.batch > 1) {
        unsigned rw = std::min<unsigned>(std::min<unsigned>(g.batch, 1024u / threads), rw_cap);
        // Row groups shorten the serial chain of ONE block (a recurrence step has a handful of blocks and nothing else to
        // run); a product with thousands of blocks is better served by more, smaller blocks per CU — their staging,
        // summing and adding phases interleave instead of 16 waves idling through them together (interval 128^3:
        // 741 ms with 8 groups, 537 ms with 4).
        const unsigned long long nblocks = n_outer * chunks;
        if (nblocks >= 1024) rw = std::min(rw, 4u);  // (2 groups: 128^3 the same, mixed-sign 64^3 — the long general path — 28 % slower)
        if (rw > 1 && lds + (size_t)g.batch * threads * 8 * W <= 64 * 1024) {
            g.rw = rw;
            lds += (size_t)g.batch * threads * 8 * W;
        }
    }
    if (S == 2 && a.inner_from_zero && std::min(g.sxa, g.sya) >= 4) {
        unsigned rw = std::min<unsigned>(8u, 1024u / threads);
        if (rw > 1 && lds + (size_t)8 * threads * 8 * W <= 64 * 1024) {
            g.rw = rw;
            lds += (size_t)8 * threads * 8 * W;
        }
    }
    const unsigned block_threads = threads * g.rw;
    unsigned long long blocks = n_outer * chunks;
    if (blocks == 0) return true;
    if (blocks > 0x7fffffffULL) return false;
#define GFT_ST(I0, SS) launch<E, I0, SS>(st, x, xp, y, yp, z, zp, a, g, (unsigned)blocks, block_threads, lds)
    if (a.inner_from_zero) return S == 2 ? GFT_ST(true, 2) : GFT_ST(true, 1);
    return S == 2 ? GFT_ST(false, 2) : GFT_ST(false, 1);
#undef GFT_ST
}

template bool conv_staged<EF64>(hipStream_t, const double*, size_t, const double*, size_t, double*, size_t,
                                const ConvArgs&, bool);
template bool conv_staged<EIv>(hipStream_t, const double*, size_t, const double*, size_t, double*, size_t,
                               const ConvArgs&, bool);

}  // namespace gft
