// LDS-tiled f64 truncated N-d convolution for gfx950 (the hot loop of TaylorPoly * TaylorPoly,
// src/multivariate_taylor.rs:971-1012) — compute-bound on FP64 FMA, so the design goal is to keep
// the 4 SIMDs of every CU issuing v_fma_f64 with both operands already on-chip:
//
//   canonical problem   z[u][k0][k1][k2] = sum_j x[ju][j0][j1][j2] * y[u-ju][k0-j0][k1-j1][k2-j2]
//                       (rank 3: U = 1; rank 4: U = leading axis)
//   lanes               the 64 lanes of a wave own a T0 x T1 tile of (k0,k1) output ROWS (8x8 by default; 4x16, 2x32 or
//                       1x64 when k0 is short or absent: rank 2, piece-split products); each lane keeps R = 8
//                       consecutive k2 outputs of its row in registers (two such blocks per wave, c and nb-1-c, so
//                       every wave of the workgroup has the same trip count although the index space is triangular)
//   x operand           wave-uniform: x[J][j2..j2+7] comes from scalar loads (constant address space =>
//                       s_load_dwordx16) into SGPRs and is the SGPR source of v_fma_f64 — zero VGPR/LDS
//                       traffic for one of the two operands
//   y operand           a T0 x T1 window of y rows lives in LDS, row slot r at offset r*P1 doubles with P1/2 odd
//                       (conflict-free ds_read_b128 per 16-lane group); each lane slides an 8-wide register window
//                       along its row: 8 new doubles from LDS per 64 FMAs
//   window motion       the step space is walked row by row (a run of j1 steps under one (ju, j0)): stepping j1
//                       replaces one of the T1 ring slots (T0 rows, prefetched global->VGPR during the step, written
//                       to LDS after it); a new row starts with a full window load
//   triangular waste    none along k2 (the diagonal 8x8 chunk is a 36-FMA triangle); (n+8)/(n+1) per
//                       lane axis from masked lanes in diagonal tiles
//   load balance        stream-K: the linearised (tile, ju, j0, j1) step space is cut into equal
//                       contiguous ranges, one per resident workgroup (2 per CU).  Tiles covered by a
//                       single range are written straight to z; split tiles go through partial slabs
//                       in a workspace and a fixed-order reduce kernel => deterministic results.
//
// Numerics: explicit fma (one rounding per MAC) and a different summation order than the
// reference => 1e-10 relative parity, not bit-exact (the reference-order kernel in
// gft_kernels.hip is the bit-exact path).  Operands must be finite (zero padding times inf would
// create NaNs the reference does not produce); the caller checks and falls back otherwise.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "gft_kernels.hpp"

namespace gft {

namespace {

struct TileSeg {
    unsigned u, a, b;                // tile coordinates
    unsigned step_begin, step_end;   // range in the tile's linearised (ju, j0, j1) space
    int dest;                        // -1: write z directly; >= 0: workspace slot
};

struct RedTile {
    unsigned u, a, b;
    unsigned first, count;           // pieces: red_slots[first .. first+count)
    unsigned slot0;                  // red_slots == null: the pieces are the workspace slots slot0 .. slot0 + count - 1
};

struct TiledArgs {
    unsigned xU, x0, x1, xI;
    unsigned yU, y0, y1, yI;
    unsigned zU, z0, z1, zI;
    unsigned nx8, ny8;               // padded inner lengths of the packed operands
    unsigned nxc, nyc, nb;           // chunk counts (nx8/8, ny8/8) and number of 8-wide output blocks
    unsigned P1;                     // LDS row pitch in doubles
    unsigned tsh;                    // lane tile: T1 = 1 << tsh lanes along k1, T0 = 64 >> tsh along k0 (8x8 ... 1x64)
    unsigned slab_axis;              // the slab range applies to: 0 = u, 1 = k0, 2 = k1
    unsigned slab_lo, slab_hi;
    int accumulate;
    unsigned xcd_remap;              // 1: remap blockIdx so that each XCD owns a contiguous chunk of ranges
    const double* xp;
    const double* yp;
    double* z;
    double* ws;
    const TileSeg* segs;
    const unsigned* wg_begin;        // segments of workgroup w: [wg_begin[w], wg_begin[w+1])
    const RedTile* red;
    const unsigned* red_slots;
    const unsigned* guard;           // non-finite verdict word: main/reduce kernels do nothing if *guard == guard_epoch
    unsigned guard_epoch;
    unsigned char blk1[8], blk2[8];  // output blocks (8 consecutive k2) owned by wave w; 0xff = none
};

// x rows are read through the constant address space: the address is wave-uniform and x is never
// written by this kernel, so hipcc emits s_load_dwordx16 into SGPRs (scalar cache path) instead of
// 64-lane broadcast vector loads that would occupy the texture-address unit.
typedef const double __attribute__((address_space(4))) * cptr_t;

#ifndef GFT_TILED_DEFAULT_VARIANT
#define GFT_TILED_DEFAULT_VARIANT 7
#endif
// VAR bits: 1 = software-pipelined fast path for full inner extents; diagnostics (wrong results, timing
// only, built with -DGFT_TILED_DIAG): 16 = no LDS reads in the chunk loop, 32 = no scalar x loads,
// 64 = no window maintenance / barriers.

__device__ inline void loadx(double (&dst)[8], cptr_t p) {  // wave-uniform: 8 doubles -> 16 SGPRs
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = p[i];
}

__device__ inline void load8(double (&dst)[8], const double* p) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = p[i];
}

// 16-byte aligned window chunk: four ds_read_b128 (4 LDS cycles each) instead of four ds_read2_b64 (8 each)
__device__ inline void load8_b128(double (&dst)[8], const double* p) {
    const double2* q = reinterpret_cast<const double2*>(__builtin_assume_aligned(p, 16));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double2 v = q[i];
        dst[2 * i] = v.x;
        dst[2 * i + 1] = v.y;
    }
}

__device__ inline void fma_rows(double (&acc)[8], const double (&xq)[8], const double (&cur)[8],
                                const double (&prev)[8], int s_lo, int s_hi) {
    // acc[r] += x[8q+s] * y[8(c-q) + r - s]; r-s >= 0 -> cur[r-s], else prev[8 + r - s]
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < s_lo || s >= s_hi) continue;
        const double xs = xq[s];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double yv = (r - s >= 0) ? cur[(r - s) & 7] : prev[(8 + r - s) & 7];
            acc[r] = __builtin_fma(xs, yv, acc[r]);
        }
    }
}

__device__ inline void fma_full(double (&acc)[8], const double (&xq)[8], const double (&cur)[8],
                                const double (&prev)[8]) {
    fma_rows(acc, xq, cur, prev, 0, 8);
}

__device__ inline void fma_tri(double (&acc)[8], const double (&xq)[8], const double (&cur)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const double xs = xq[s];
#pragma unroll
        for (int r = s; r < 8; ++r) acc[r] = __builtin_fma(xs, cur[r - s], acc[r]);
    }
}

// General path: one output block c (8 consecutive k2) of one lane-row for one (x row, y row) pair; handles
// compact operands (fewer x / y chunks than output blocks).
template <int VAR>
__device__ inline void block_mac(double (&acc)[8], unsigned c, cptr_t xr, const double* yrow, unsigned nxc,
                                 unsigned nyc) {
    constexpr bool NO_LDS = (VAR & 16) != 0;
    constexpr bool NO_X = (VAR & 32) != 0;
    unsigned q_lo = c > nyc ? c - nyc : 0;
    unsigned q_hi = c < nxc ? c : nxc;  // full chunks q in [q_lo, q_hi)
    double A[8], B[8], X[8];
    unsigned m0 = c - q_lo;
    if (m0 < nyc) {
        load8(A, yrow + 8 * m0);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) A[i] = 0.0;
    }
    if (NO_LDS) load8(B, yrow);
    if (NO_X) loadx(X, xr);
    unsigned q = q_lo;
    // two chunks per iteration so the sliding window alternates between A and B without moves
    for (; q + 2 <= q_hi; q += 2) {
        if (!NO_LDS) load8(B, yrow + 8 * (c - q - 1));
        if (!NO_X) loadx(X, xr + 8 * q);
        fma_full(acc, X, A, B);
        if (!NO_LDS) load8(A, yrow + 8 * (c - q - 2));
        if (!NO_X) loadx(X, xr + 8 * (q + 1));
        fma_full(acc, X, B, A);
    }
    if (q < q_hi) {
        if (!NO_LDS) load8(B, yrow + 8 * (c - q - 1));
        if (!NO_X) loadx(X, xr + 8 * q);
        fma_full(acc, X, A, B);
        if (c < nxc) {
            if (!NO_X) loadx(X, xr + 8 * c);
            fma_tri(acc, X, B);
        }
    } else if (c < nxc) {
        if (!NO_X) loadx(X, xr + 8 * c);
        fma_tri(acc, X, A);
    }
}

// Fast path (x and y span every chunk the block touches: q_lo = 0, q_hi = c).  The window buffers rotate three ways
// (A, B, C: current, previous, in flight), the x buffers two ways (X0, X1 — 32 SGPRs; a third set pushed the kernel's
// scalar state out of the register file, and every per-step scalar then cost a v_readlane from a spill lane): chunk i
// uses (X_i; cur, prev) and, right after its first FMA row — i.e. after the s_waitcnt that the first use of its own
// operands triggers — requests x[q+i+1] and W[q+i+2] for the following chunks.  Every LDS / scalar-load result is
// therefore >= 56 FMAs (~250 cycles) old when first used, and the window slides without register moves.  The
// scheduling barriers pin this order (hipcc otherwise sinks the loads next to their uses and stalls on them).
// The diagonal triangle (x chunk c against y chunk 0) goes FIRST, on the buffers the pipeline fills last (X1, C), so
// the chunk loop needs no per-remainder epilogues: six phases (the rotation's period), then up to five of them again.
// W[t] = yrow[8(c-t) ..+7]; t = c+1 reads the row's 8-double front padding — loaded but never used.
#define GFT_STEP(XU, CUR, PREV, XN, TX, WN, TW)                 \
    do {                                                        \
        fma_rows(acc, XU, CUR, PREV, 0, 1);                     \
        __builtin_amdgcn_sched_barrier(0);                      \
        if (!NO_X) loadx(XN, xr + 8 * (TX));                    \
        if (!NO_LDS) { if (B128) load8_b128(WN, w0 - 8 * (int)(TW)); else load8(WN, w0 - 8 * (int)(TW)); } \
        __builtin_amdgcn_sched_barrier(0);                      \
        fma_rows(acc, XU, CUR, PREV, 1, 8);                     \
    } while (0)
#define GFT_PH0 GFT_STEP(X0, A, B, X1, q + 1, C, q + 2)
#define GFT_PH1 GFT_STEP(X1, B, C, X0, q + 2, A, q + 3)
#define GFT_PH2 GFT_STEP(X0, C, A, X1, q + 3, B, q + 4)
#define GFT_PH3 GFT_STEP(X1, A, B, X0, q + 4, C, q + 5)
#define GFT_PH4 GFT_STEP(X0, B, C, X1, q + 5, A, q + 6)
#define GFT_PH5 GFT_STEP(X1, C, A, X0, q + 6, B, q + 7)
// chunks q .. q_hi-1 with (X0; A, B) holding chunk q's operands
#define GFT_CHUNK_LOOP(Q_HI)                                    \
    do {                                                        \
        for (; q + 6 <= (Q_HI); q += 6) {                       \
            GFT_PH0; GFT_PH1; GFT_PH2; GFT_PH3; GFT_PH4; GFT_PH5; \
        }                                                       \
        const unsigned rem = (Q_HI) - q;                        \
        if (rem >= 1) {                                         \
            GFT_PH0;                                            \
            if (rem >= 2) {                                     \
                GFT_PH1;                                        \
                if (rem >= 3) {                                 \
                    GFT_PH2;                                    \
                    if (rem >= 4) {                             \
                        GFT_PH3;                                \
                        if (rem >= 5) GFT_PH4;                  \
                    }                                           \
                }                                               \
            }                                                   \
        }                                                       \
    } while (0)

template <int VAR>
__device__ inline void block_fast(double (&acc)[8], unsigned c, cptr_t xr, const double* yrow) {
    constexpr bool NO_LDS = (VAR & 16) != 0;
    constexpr bool NO_X = (VAR & 32) != 0;
    constexpr bool B128 = (VAR & 2) != 0;
    const double* w0 = yrow + 8 * c;  // W[t] = w0 - 8t
    double A[8], B[8], C[8], X0[8], X1[8];
    // everything the first two chunks need is requested up front (for c = 0 the pipeline's operands are loaded and never
    // used: x chunk 0, W[0] = the triangle's own window, W[1] = the front padding)
    loadx(X1, xr + 8 * c);
    loadx(X0, xr);
    if (B128) { load8_b128(C, yrow); load8_b128(A, w0); load8_b128(B, w0 - 8); } else { load8(C, yrow); load8(A, w0); load8(B, w0 - 8); }
    __builtin_amdgcn_sched_barrier(0);
    fma_tri(acc, X1, C);
    unsigned q = 0;
    GFT_CHUNK_LOOP(c);
}

// The same pipeline for COMPACT operands (fewer x / y chunks than output blocks: the piece-split products, Horner
// steps with a small substitution): x chunks q in [q_lo, q_hi) with q_lo = max(0, c - nyc), q_hi = min(c, nxc), the
// diagonal triangle only if c < nxc, and the first window is zero when it would lie beyond y's last chunk.
// A separate function (and instantiation, VAR bit 8) so that the full-extent kernel's code is untouched.
template <int VAR>
__device__ inline void block_fast_gen(double (&acc)[8], unsigned c, cptr_t xr, const double* yrow, unsigned nxc,
                                      unsigned nyc) {
    constexpr bool NO_LDS = (VAR & 16) != 0;
    constexpr bool NO_X = (VAR & 32) != 0;
    constexpr bool B128 = (VAR & 2) != 0;
    const unsigned q_lo = c > nyc ? c - nyc : 0, q_hi = c < nxc ? c : nxc;
    const bool tri = c < nxc;
    if (q_hi <= q_lo && !tri) return;
    const double* w0 = yrow + 8 * c;  // W[t] = w0 - 8t = y chunk c - t
    double A[8], B[8], C[8], X0[8], X1[8];
    // the scalar (x) loads are unconditional, on clamped chunk indices: a conditionally loaded SGPR array becomes a phi
    // that hipcc parks in VGPRs (16 of them, and the kernel has none to spare)
    loadx(X1, xr + 8 * (tri ? c : 0u));
    loadx(X0, xr + 8 * (q_hi > q_lo ? q_lo : 0u));
    if (B128) load8_b128(C, yrow); else load8(C, yrow);
    if (c - q_lo < nyc) {
        if (B128) load8_b128(A, w0 - 8 * (int)q_lo); else load8(A, w0 - 8 * (int)q_lo);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) A[i] = 0.0;
    }
    if (B128) load8_b128(B, w0 - 8 * (int)(q_lo + 1)); else load8(B, w0 - 8 * (int)(q_lo + 1));
    __builtin_amdgcn_sched_barrier(0);
    if (tri) fma_tri(acc, X1, C);
    unsigned q = q_lo;
    GFT_CHUNK_LOOP(q_hi);
}

struct TileGeom {
    unsigned julo, n_ju, j0lo, n_j0, j1lo, n_j1;
};

__host__ __device__ inline TileGeom tile_geom(const TiledArgs& A, unsigned tsh, unsigned u, unsigned a, unsigned b) {
    TileGeom g;
    const unsigned T0 = 64u >> tsh, T1 = 1u << tsh;
    // uniform axis: ju in [max(0,u+1-yU), min(u+1,xU))
    g.julo = (u + 1 > A.yU) ? (u + 1 - A.yU) : 0;
    unsigned juhi = (u + 1 < A.xU) ? (u + 1) : A.xU;
    g.n_ju = juhi > g.julo ? juhi - g.julo : 0;
    // lane axes: any lane of the tile valid.  k in [T a, min(T a + T - 1, z - 1)]
    unsigned k0max = (T0 * a + T0 - 1 < A.z0 - 1) ? T0 * a + T0 - 1 : A.z0 - 1;
    g.j0lo = (T0 * a + 1 > A.y0) ? (T0 * a + 1 - A.y0) : 0;
    unsigned j0hi = (k0max + 1 < A.x0) ? (k0max + 1) : A.x0;
    g.n_j0 = j0hi > g.j0lo ? j0hi - g.j0lo : 0;
    unsigned k1max = (T1 * b + T1 - 1 < A.z1 - 1) ? T1 * b + T1 - 1 : A.z1 - 1;
    g.j1lo = (T1 * b + 1 > A.y1) ? (T1 * b + 1 - A.y1) : 0;
    unsigned j1hi = (k1max + 1 < A.x1) ? (k1max + 1) : A.x1;
    g.n_j1 = j1hi > g.j1lo ? j1hi - g.j1lo : 0;
    return g;
}

constexpr unsigned YPAD = 8;  // front padding of every LDS row (doubles)

// (Round 4 tried the reduction of split tiles INSIDE this kernel once more — a two-level tree of last arrivers: the workgroup
// that stores a piece arrives at its group of 8 consecutive pieces, a group's last arriver sums the group in slab order
// and arrives at the tile, whose last arriver sums the group sums and writes z; deterministic, nobody waits.  Correct, and
// slower at every size: 32^3 48.7 -> 48.4 us, 64^3 447 -> 521 us, 378^2 310 -> 443 us.  Every arrival needs a device-scope
// release + acquire around its counter, which on this part is a write-back and an invalidate of the XCD's whole L2 —
// paid by a thousand workgroups while their neighbours are still streaming y windows through that L2 — and the call kept
// 116 bytes of scratch in the kernel.)

template <int NW, int VAR, int TSH>
__global__ void __launch_bounds__(NW * 64, 4)  // 4 waves per SIMD = 16 waves per CU => <= 128 VGPRs
k_conv_tiled(TiledArgs A) {
    constexpr bool FAST = (VAR & 1) != 0;
    constexpr bool NO_WINDOW = (VAR & 64) != 0;
    extern __shared__ double lds[];
    // an operand holds inf/NaN: the reference-order kernel (launched next, guarded the other way) owns z
    if (A.guard && *A.guard == A.guard_epoch) return;
    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // lane tile T0 x T1 over the (k0, k1) output rows: 8x8 by default (TSH = 3, compile-time); 1x64, 2x32 or 4x16 (TSH = 0:
    // A.tsh at run time) when axis k0 is short or absent — the rank-2 products and the piece-split ones, whose lanes
    // would otherwise sit masked
    const unsigned tsh = TSH ? (unsigned)TSH : A.tsh, T1m = (1u << tsh) - 1u, T0 = 64u >> tsh;
    const unsigned l0 = lane >> tsh, l1 = lane & T1m;
    constexpr unsigned NT = NW * 64;

    const unsigned c1 = A.blk1[wave];
    const unsigned c2 = A.blk2[wave];
    const bool has1 = c1 != 0xffu;
    const bool has2 = c2 != 0xffu;

    const unsigned half_row = A.ny8 >> 1;                  // 16-byte pieces per row
    const unsigned hr_magic = (1u << 24) / half_row + 1u;  // p / half_row == (p * magic) >> 24 for the p < 4096 used here
    const size_t y_row_stride = A.ny8;
    // this thread's piece of a ring-slot refill (one refill = T0 rows of half_row 16-byte pieces; at most one per thread:
    // T0 * ny8 / 2 <= 64 NW because yI <= zI and NW >= nb / 2)
    const unsigned pf_s0 = (tid * hr_magic) >> 24, pf_col = tid - pf_s0 * half_row;
    const bool pf_mine = pf_s0 < T0;
    double* const lds_lane = lds + (size_t)(l0 << tsh) * A.P1 + YPAD;
    double* const lds_pf = lds + (size_t)(pf_s0 << tsh) * A.P1 + YPAD + 2 * pf_col;

    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a CONTIGUOUS eighth of
    // the stream-K ranges — neighbouring ranges sweep overlapping y-row windows and then share one L2.
    unsigned wg = blockIdx.x;
    if (A.xcd_remap) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const unsigned seg_begin = A.wg_begin[wg], seg_end = A.wg_begin[wg + 1];
    for (unsigned si = seg_begin; si < seg_end; ++si) {
        const TileSeg seg = A.segs[si];
        const unsigned u = seg.u, a = seg.a, b = seg.b;
        const TileGeom g = tile_geom(A, tsh, u, a, b);
        const unsigned k0 = T0 * a + l0, k1 = (b << tsh) + l1;
        bool lane_in = k0 < A.z0 && k1 < A.z1;
        if (A.slab_axis == 1) lane_in = lane_in && k0 >= A.slab_lo && k0 < A.slab_hi;
        else if (A.slab_axis == 2) lane_in = lane_in && k1 >= A.slab_lo && k1 < A.slab_hi;

        double acc1[8], acc2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc1[i] = acc2[i] = 0.0;

        unsigned tj1 = seg.step_begin % g.n_j1;
        unsigned t = seg.step_begin / g.n_j1;
        unsigned tj0 = t % g.n_j0;
        unsigned ju = g.julo + t / g.n_j0, j0 = g.j0lo + tj0, j1 = g.j1lo + tj1;
        unsigned steps_left = seg.step_end - seg.step_begin;
        const unsigned j1_end = g.j1lo + g.n_j1;

        // The step space is walked row by row: a ROW is a run of j1 steps under one (ju, j0) — it starts with a full
        // window load, and inside it every per-step quantity is a running value (x row pointer, the lane's k1 - j1,
        // the refill piece's source pointer) instead of a function of (ju, j0, j1) recomputed from the plan's scalars.
        while (steps_left) {
            const unsigned row_steps = steps_left < j1_end - j1 ? steps_left : j1_end - j1;
            const double* ybase = A.yp + (size_t)(u - ju) * A.y0 * A.y1 * y_row_stride;
            if (!NO_WINDOW) {
                // (re)load the whole T0 x T1 window of y rows for (ju, j0, j1)
                // (round 4: the <= 8 pieces of a thread are loaded one after the other — each load sits under its range test, a
                // block of its own, and hipcc waits for vmcnt(0) at block boundaries.  Issuing all eight unconditionally first
                // was measured: 16 hoisted per-piece invariants spill, 64^3 410 -> 429 us, 24^4 636 -> 704 us, 128^3 -2 %: the
                // other workgroups of the CU already cover a row start)
                __syncthreads();
                for (unsigned p = tid; p < 64 * half_row; p += NT) {
                    unsigned row = (p * hr_magic) >> 24, col = p - row * half_row;
                    unsigned s0 = row >> tsh, s1 = row & T1m;
                    int R0 = (int)(T0 * a + s0) - (int)j0;
                    int R1 = (int)((b << tsh) + s1) - (int)j1;
                    if (R0 >= 0 && R0 < (int)A.y0 && R1 >= 0 && R1 < (int)A.y1) {
                        const double2 v = *reinterpret_cast<const double2*>(
                            ybase + ((size_t)R0 * A.y1 + (size_t)R1) * y_row_stride + 2 * col);
                        double* d = lds + ((s0 << tsh) + ((unsigned)R1 & T1m)) * A.P1 + YPAD + 2 * col;
                        d[0] = v.x;
                        d[1] = v.y;
                    }
                }
                __syncthreads();
            }
            // running values of the row
            cptr_t xr = (cptr_t)(A.xp + (((size_t)ju * A.x0 + j0) * A.x1 + j1) * A.nx8);  // wave-uniform
            const bool valid0 = lane_in && j0 <= k0 && (k0 - j0) < A.y0;
            int d1 = (int)k1 - (int)j1;               // this lane's y row along axis 1
            int R1n = (int)(b << tsh) - (int)j1 - 1;  // the y row the NEXT step's window gains
            const int pf_R0 = (int)(T0 * a + pf_s0) - (int)j0;
            const bool pf_row = pf_mine && pf_R0 >= 0 && pf_R0 < (int)A.y0;
            const double* pf_src = ybase + ((ptrdiff_t)pf_R0 * (ptrdiff_t)A.y1 + R1n) * (ptrdiff_t)y_row_stride + 2 * pf_col;

            for (unsigned rs = row_steps; rs > 0; --rs) {
                const bool last = rs == 1;  // last step of the row (or of the range): no refill, no barriers
                // prefetch the ring slot that the next j1 step needs: rows (.., R1n) — nothing else is new
                const bool do_pf = !NO_WINDOW && !last && R1n >= 0 && R1n < (int)A.y1;
                double2 pf;
                if (do_pf && pf_row) pf = *reinterpret_cast<const double2*>(pf_src);

                // ---- compute this step -------------------------------------------------------------
                if (valid0 && (unsigned)d1 < A.y1) {  // j1 <= k1 and k1 - j1 < y1
                    const double* yrow = lds_lane + (size_t)((unsigned)d1 & T1m) * A.P1;
                    if constexpr (FAST && (VAR & 8)) {  // pipelined path for compact operands
                        if (has1) block_fast_gen<VAR>(acc1, c1, xr, yrow, A.nxc, A.nyc);
                        if (has2) block_fast_gen<VAR>(acc2, c2, xr, yrow, A.nxc, A.nyc);
                    } else if constexpr (FAST) {  // host guarantees nxc, nyc >= nb for this instantiation
                        if (has1) block_fast<VAR>(acc1, c1, xr, yrow);
                        if (has2) block_fast<VAR>(acc2, c2, xr, yrow);
                    } else {
                        if (has1) block_mac<VAR>(acc1, c1, xr, yrow, A.nxc, A.nyc);
                        if (has2) block_mac<VAR>(acc2, c2, xr, yrow, A.nxc, A.nyc);
                    }
                }

                // ---- advance within the row ----------------------------------------------------------
                if (!last && !NO_WINDOW) {
                    __syncthreads();  // everyone is done reading the slot being replaced
                    if (do_pf && pf_row) {
                        double* d = lds_pf + (size_t)((unsigned)R1n & T1m) * A.P1;
                        d[0] = pf.x;
                        d[1] = pf.y;
                    }
                    __syncthreads();
                }
                xr += A.nx8;
                d1 -= 1;
                R1n -= 1;
                pf_src -= y_row_stride;
            }
            steps_left -= row_steps;
            // next row
            j1 = g.j1lo;
            if (j0 + 1 < g.j0lo + g.n_j0) j0++;
            else {
                j0 = g.j0lo;
                ju++;
            }
        }

        // ---- write out ---------------------------------------------------------------------------------
        if (seg.dest < 0) {
            if (lane_in) {
                double* zrow = A.z + (((size_t)u * A.z0 + k0) * A.z1 + k1) * A.zI;
                if (has1) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        unsigned k2 = 8 * c1 + r;
                        if (k2 < A.zI) zrow[k2] = A.accumulate ? zrow[k2] + acc1[r] : acc1[r];
                    }
                }
                if (has2) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        unsigned k2 = 8 * c2 + r;
                        if (k2 < A.zI) zrow[k2] = A.accumulate ? zrow[k2] + acc2[r] : acc2[r];
                    }
                }
            }
        } else {
            double* w = A.ws + (size_t)seg.dest * A.nb * 512;
            if (has1) {
                double* d = w + ((size_t)c1 * 64 + lane) * 8;
#pragma unroll
                for (int r = 0; r < 8; ++r) d[r] = acc1[r];
            }
            if (has2) {
                double* d = w + ((size_t)c2 * 64 + lane) * 8;
#pragma unroll
                for (int r = 0; r < 8; ++r) d[r] = acc2[r];
            }
        }
    }
}

// Fixed-order sum of the partial slabs of split tiles.  One workgroup per (tile, 8-wide output block); its four waves
// each sum a contiguous quarter of the tile's partial slabs and the quarters are added in order through LDS — the order
// depends on the plan only, so results are deterministic.  (A slab step of a recurrence has few tiles and many partial
// slabs: one workgroup per tile walking all of them serially took 140 us at 64^3, 2x the product itself.)
__global__ void __launch_bounds__(256) k_conv_reduce(TiledArgs A, unsigned n_red) {
    const unsigned ti = blockIdx.x, c = blockIdx.y;
    if (ti >= n_red) return;
    if (A.guard && *A.guard == A.guard_epoch) return;
    __shared__ double part[3][64][8];
    const RedTile rt = A.red[ti];
    const unsigned lane = threadIdx.x & 63u, q = threadIdx.x >> 6;
    const unsigned k0 = (64u >> A.tsh) * rt.a + (lane >> A.tsh), k1 = (rt.b << A.tsh) + (lane & ((1u << A.tsh) - 1u));
    bool lane_in = k0 < A.z0 && k1 < A.z1;
    if (A.slab_axis == 1) lane_in = lane_in && k0 >= A.slab_lo && k0 < A.slab_hi;
    else if (A.slab_axis == 2) lane_in = lane_in && k1 >= A.slab_lo && k1 < A.slab_hi;
    double v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = 0.0;
    double* zrow = A.z + (((size_t)rt.u * A.z0 + k0) * A.z1 + k1) * A.zI;
    if (lane_in) {
        if (q == 0 && A.accumulate) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                unsigned k2 = 8 * c + r;
                if (k2 < A.zI) v[r] = zrow[k2];
            }
        }
        const unsigned lo = (unsigned)((unsigned long long)rt.count * q / 4), hi = (unsigned)((unsigned long long)rt.count * (q + 1) / 4);
        // (a tile's partial slabs are consecutive workspace slots — the plan's ranges are contiguous in tile order — so
        // the slab address follows from the tile's entry alone: one dependent load less in a launch that is nothing
        // but a few memory latencies; and four slabs' loads go out before the first is added — in the same order as before)
        unsigned p = lo;
        if (!A.red_slots) {
            const size_t stride = (size_t)A.nb * 512;
            const double* w = A.ws + (size_t)(rt.slot0 + lo) * stride + ((size_t)c * 64 + lane) * 8;
            for (; p + 4 <= hi; p += 4, w += 4 * stride) {
                double t[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 8; ++r) t[u][r] = w[(size_t)u * stride + r];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += t[u][r];
            }
        }
        for (; p < hi; ++p) {
            const unsigned slot = A.red_slots ? A.red_slots[rt.first + p] : rt.slot0 + p;
            const double* w = A.ws + (size_t)slot * A.nb * 512 + ((size_t)c * 64 + lane) * 8;
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += w[r];
        }
        if (q > 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) part[q - 1][lane][r] = v[r];
        }
    }
    __syncthreads();
    if (q == 0 && lane_in) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += part[g][lane][r];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            unsigned k2 = 8 * c + r;
            if (k2 < A.zI) zrow[k2] = v[r];
        }
    }
}
// (Round 4 tried 16 waves per (tile, block) with four slabs' loads in flight: 32^3 18 -> 16 us, but 128^3 65 -> 233 us — 60 KB of
// static LDS per workgroup leaves two of them per CU to read 285 MB of slabs.  Kept as it was.)

// out[row][i] = i < len ? in[row][i] : 0, rows x n8
// Operand preparation: rows are copied into the zero-padded packed layout and, on the way, *flag is raised to
// `epoch` if any element is inf/NaN (the verdict is an epoch stamp, so the word never needs resetting).
// Both operands in one launch: x is always packed; y is packed if yp != nullptr, otherwise only scanned.
__global__ void __launch_bounds__(256) k_prep_operands(const double* __restrict__ x, double* __restrict__ xp, size_t x_rows,
                                                       unsigned xlen, unsigned nx8, const double* __restrict__ y,
                                                       double* __restrict__ yp, size_t y_rows, unsigned ylen, unsigned ny8,
                                                       unsigned* flag, unsigned epoch) {
    const size_t tx = x_rows * nx8, ty = yp ? y_rows * ny8 : y_rows * ylen;
    bool bad = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tx + ty; i += (size_t)gridDim.x * blockDim.x) {
        double v;
        if (i < tx) {
            size_t row = i / nx8;
            unsigned col = (unsigned)(i - row * nx8);
            v = col < xlen ? x[row * xlen + col] : 0.0;
            xp[i] = v;
        } else if (yp) {
            size_t k = i - tx, row = k / ny8;
            unsigned col = (unsigned)(k - row * ny8);
            v = col < ylen ? y[row * ylen + col] : 0.0;
            yp[k] = v;
        } else {
            v = y[i - tx];
        }
        if (!((v - v) == 0.0)) bad = true;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMax(flag, epoch);
}

// ---- inner-axis splitting (rank 2, or an inner axis longer than 128) ---------------------------------------------
// A row of length n is viewed as P = ceil(n / B) pieces of B: x~[p][r] = x[pB + r] (zero padded).  The product of
// the (.., Px, B) and (.., Py, B) tensors with an UNtruncated last axis (2B - 1 <= 127 coefficients) contains
// exactly the products of the original one: z[pB + r] = z~[p][r] + z~[p - 1][B + r] (overlap-add of the carries).
// That turns a rank-d product with a long last axis into a rank-(d+1) product the tiled kernel supports.
// Packed layouts put the PIECE axis first — x~[p][row][r], z~[p][row][0..2B-2] — so that the tiled kernel sees it as
// its wave-uniform axis u and the lanes tile real rows (pieces are few: as a lane axis they left most lanes masked).
__global__ void __launch_bounds__(256) k_pad_rows(const double* __restrict__ in, double* __restrict__ out, size_t rows,
                                                  unsigned len, unsigned P, unsigned B) {
    const size_t per_p = rows * B, total = per_p * P;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned p = (unsigned)(i / per_p);
        const size_t rem = i - (size_t)p * per_p, row = rem / B;
        const unsigned col = p * B + (unsigned)(rem - row * B);
        out[i] = col < len ? in[row * len + col] : 0.0;
    }
}
__global__ void __launch_bounds__(256) k_fold_rows(const double* __restrict__ zt, double* __restrict__ z, size_t rows,
                                                   size_t row_lo, size_t row_hi, unsigned Pz, unsigned B, unsigned zI,
                                                   int accumulate, const unsigned* guard, unsigned epoch) {
    if (guard && *guard == epoch) return;
    const unsigned RI = 2 * B - 1;
    size_t total = (row_hi - row_lo) * zI;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t row = row_lo + i / zI;
        unsigned k = (unsigned)(i % zI), p = k / B, r = k - p * B;
        // z~ has Pz = min(pieces of z, pieces of x + pieces of y - 1) pieces: the last output piece may consist of the
        // carry alone (a 110-long row times a 63-long one reaches k = 171 with 2 + 1 - 1 = 2 piece products)
        double v = p < Pz ? zt[((size_t)p * rows + row) * RI + r] : 0.0;
        if (p > 0 && p - 1 < Pz && r + 1 < B) v += zt[((size_t)(p - 1) * rows + row) * RI + B + r];
        double* dst = z + row * zI + k;
        *dst = accumulate ? *dst + v : v;
    }
}

// ---- host-side plan ------------------------------------------------------------------------------------

struct PlanKey {
    unsigned v[18];
    bool operator<(const PlanKey& o) const { return std::memcmp(v, o.v, sizeof(v)) < 0; }
};

struct Plan {
    TiledArgs base;  // shapes/pitches filled in; pointers to device tables filled in
    unsigned n_wg = 0, n_red = 0, n_slots = 0, NW = 0;
    size_t lds_bytes = 0;
    void* d_tables = nullptr;
};

struct TableArena {
    char* dev = nullptr;
    char* host = nullptr;  // pinned mirror
    size_t bytes = 0, head = 0;
    bool tried = false;
    bool ensure() {
        if (dev) return true;
        if (tried) return false;
        tried = true;
        const size_t n = 32u << 20;
        if (hipMalloc((void**)&dev, n) != hipSuccess) { dev = nullptr; (void)hipGetLastError(); return false; }
        if (hipHostMalloc((void**)&host, n, hipHostMallocDefault) != hipSuccess) {
            (void)hipFree(dev);
            dev = nullptr;
            (void)hipGetLastError();
            return false;
        }
        bytes = n;
        return true;
    }
};
TableArena& arena() {
    static TableArena a;
    return a;
}

std::map<PlanKey, Plan>& plan_cache() {
    static std::map<PlanKey, Plan> c;
    return c;
}

int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

static void assign_blocks(TiledArgs& T, unsigned NW);

int& tiled_force_tsh() {  // A/B and test knob (GFT_TILED_TSH / "tiled_tile"): 3..6 forces the lane tile 8x8 .. 1x64
    static int v = 0;
    return v;
}

bool build_plan(const ConvArgs& a, Plan& P, hipStream_t st) {
    TiledArgs& T = P.base;
    std::memset(&T, 0, sizeof(T));
    // canonical form z[u][k0][k1][k2]: u is wave-uniform, (k0, k1) are the lane axes, k2 the register axis
    unsigned slab_axis;  // canonical axis the caller's slab range applies to
    if (a.slab_axis < 0 || a.slab_axis >= a.nd - 1) return false;
    if (a.nd == 2) {  // rows x inner: no k0 axis — the lane tile becomes 1 x 64 rows
        T.xU = T.yU = T.zU = 1;
        T.x0 = T.y0 = T.z0 = 1;
        T.x1 = a.xs[0]; T.xI = a.xs[1];
        T.y1 = a.ys[0]; T.yI = a.ys[1];
        T.z1 = a.zs[0]; T.zI = a.zs[1];
        slab_axis = 2;
    } else if (a.nd == 3) {
        T.xU = T.yU = T.zU = 1;
        T.x0 = a.xs[0]; T.x1 = a.xs[1]; T.xI = a.xs[2];
        T.y0 = a.ys[0]; T.y1 = a.ys[1]; T.yI = a.ys[2];
        T.z0 = a.zs[0]; T.z1 = a.zs[1]; T.zI = a.zs[2];
        slab_axis = 1 + (unsigned)a.slab_axis;
    } else if (a.nd == 4) {
        T.xU = a.xs[0]; T.x0 = a.xs[1]; T.x1 = a.xs[2]; T.xI = a.xs[3];
        T.yU = a.ys[0]; T.y0 = a.ys[1]; T.y1 = a.ys[2]; T.yI = a.ys[3];
        T.zU = a.zs[0]; T.z0 = a.zs[1]; T.z1 = a.zs[2]; T.zI = a.zs[3];
        slab_axis = (unsigned)a.slab_axis;
    } else {
        return false;
    }
    T.slab_axis = slab_axis;
    if (T.zI > 128 || T.xI > T.zI || T.yI > T.zI) return false;
    if (a.j0_min != 0 || a.j0_excl != 0 || a.j0_desc != 0) return false;  // recurrence steps: reference-order kernel
    T.nx8 = (T.xI + 7) / 8 * 8;
    T.ny8 = (T.yI + 7) / 8 * 8;
    T.nxc = T.nx8 / 8;
    T.nyc = T.ny8 / 8;
    T.nb = (T.zI + 7) / 8;
    // front padding + pitch: odd (8-byte slots, bijective mod 16/32 for ds_read_b64/read2_b64) or, for the
    // ds_read_b128 variant, even with P1/2 odd (16-byte slots bijective mod 16 for the b128 lane groups)
    T.P1 = (a.variant & 2) ? T.ny8 + YPAD + 2 : T.ny8 + YPAD + 1;
    // Lane tile: the shape whose lanes are busiest.  A lane (k0, k1) works in step (j0, j1) iff j <= k and k - j < y's
    // extent on both axes; the steps a tile runs are the union over its lanes, so a tile straddling the diagonal (or a
    // short axis) carries masked lanes.  Useful / issued lane-steps factorises over the two axes.
    {
        auto axis_eff = [](unsigned T, unsigned nx, unsigned ny, unsigned nz) {
            double useful = 0, issued = 0;
            for (unsigned t = 0; t * T < nz; ++t) {
                const unsigned kmax = std::min(T * t + T - 1, nz - 1);
                const unsigned jlo = T * t + 1 > ny ? T * t + 1 - ny : 0, jhi = std::min(kmax + 1, nx);
                if (jhi > jlo) issued += (double)(jhi - jlo) * T;
                for (unsigned k = T * t; k <= kmax; ++k) {
                    const unsigned lo = k + 1 > ny ? k + 1 - ny : 0, hi = std::min(k + 1, nx);
                    if (hi > lo) useful += hi - lo;
                }
            }
            return issued > 0 ? useful / issued : 0.0;
        };
        const int force = tiled_force_tsh();
        double best = -1.0;
        unsigned best_tsh = 3;
        for (unsigned tsh = 3; tsh <= 6; ++tsh) {
            const double e = axis_eff(64u >> tsh, T.x0, T.y0, T.z0) * axis_eff(1u << tsh, T.x1, T.y1, T.z1);
            if (e > best * 1.02) {  // ties (and near ties) go to the squarer tile: fewer window reloads per step
                best = e;
                best_tsh = tsh;
            }
        }
        T.tsh = (force >= 3 && force <= 6) ? (unsigned)force : best_tsh;
    }
    const unsigned TT0 = 64u >> T.tsh, TT1 = 1u << T.tsh;
    T.slab_lo = a.slab_lo;
    T.slab_hi = a.slab_hi;
    T.accumulate = a.accumulate;
    unsigned npairs = (T.nb + 1) / 2;
    P.NW = npairs <= 1 ? 1 : (npairs <= 2 ? 2 : (npairs <= 4 ? 4 : 8));
    P.lds_bytes = (size_t)64 * T.P1 * sizeof(double);
    assign_blocks(T, P.NW);

    // tiles in (u, a, b) order
    struct TileInfo { unsigned u, a, b; unsigned long long steps; };
    std::vector<TileInfo> tiles;
    unsigned u_lo = 0, u_hi = T.zU, a_lo = 0, a_hi = (T.z0 + TT0 - 1) / TT0, b_lo = 0, b_hi = (T.z1 + TT1 - 1) / TT1;
    if (slab_axis == 0) {
        u_lo = a.slab_lo;
        u_hi = a.slab_hi;
    } else if (slab_axis == 1) {
        a_lo = a.slab_lo / TT0;
        a_hi = (a.slab_hi + TT0 - 1) / TT0;
    } else {
        b_lo = a.slab_lo / TT1;
        b_hi = (a.slab_hi + TT1 - 1) / TT1;
    }
    unsigned long long S = 0;
    for (unsigned u = u_lo; u < u_hi; ++u)
        for (unsigned aa = a_lo; aa < a_hi; ++aa)
            for (unsigned bb = b_lo; bb < b_hi; ++bb) {
                TileGeom g = tile_geom(T, T.tsh, u, aa, bb);
                unsigned long long st = (unsigned long long)g.n_ju * g.n_j0 * g.n_j1;
                if (st == 0 || st > 0xffffffffull) return false;
                tiles.push_back({u, aa, bb, st});
                S += st;
            }
    if (tiles.empty()) return false;
    unsigned wg_per_cu = std::min<unsigned>(16u / P.NW, (unsigned)(160 * 1024 / P.lds_bytes));
    if (wg_per_cu < 1) wg_per_cu = 1;
    // Ranges per resident workgroup slot: with exactly one range per slot every workgroup runs from the first to the last
    // cycle of the launch and the slowest (its CU's clock, its neighbours' traffic) sets the time; four ranges per slot let
    // the dispatcher even that out — as many (2, 4, 8) as leave a range >= ~190 steps (128^3: 2312 steps per slot, eight ranges, 27.0 -> 28.7
    // TMAC/s in three alternating runs on one box; 96^3 and 100^3 four, +3 %; 64^3, 81 steps per slot, stays at one: two
    // are neutral, four lose 4 % to the extra partial slabs).
    static const int wg_mult_env = 0;
    unsigned long long n_wg = (unsigned long long)num_cus() * wg_per_cu;
    const unsigned long long per_slot = S / n_wg;  // steps (one lane tile x one (ju, j0, j1)) per resident slot
    const unsigned long long fit = per_slot / 190;  // ranges of >= ~190 steps
    n_wg *= wg_mult_env ? (unsigned)wg_mult_env : (fit >= 8 ? 8u : (fit >= 4 ? 4u : (fit >= 2 ? 2u : 1u)));
    // Every range pays a window fill and, if it splits a tile, a 4 KB-per-block partial slab plus its share of the
    // reduction, so small products must not be cut into confetti.  A step costs ~ (chunk pairs + 3) units
    // (pairs = nb(nb+1)/2 8x8x8 chunk products per lane tile, 3 ~ barriers + refill) and the fixed part grows
    // with the row length; ranges get >= ~(64 + 20 nb) units (sweep on MI355X: 24^3 171 -> 50 us,
    // 30^3 91 -> 71 us, 10x10x100 154 -> 92 us per product).
    static const unsigned long long MIN_UNITS = 64;
    const unsigned long long step_units = (unsigned long long)T.nb * (T.nb + 1) / 2 + 3;
    const unsigned long long min_steps =
        std::max<unsigned long long>(1, (MIN_UNITS + 20ull * T.nb + step_units / 2) / step_units);
    if (n_wg > S / min_steps) n_wg = std::max<unsigned long long>(1, S / min_steps);
    P.n_wg = (unsigned)n_wg;

    std::vector<TileSeg> segs;
    std::vector<unsigned> wg_begin(P.n_wg + 1, 0);
    std::vector<std::vector<unsigned>> tile_slots(tiles.size());
    std::vector<int> tile_direct(tiles.size(), 0);
    unsigned n_slots = 0;
    size_t ti = 0;
    unsigned long long tile_start = 0;  // global step index where tile ti starts
    for (unsigned w = 0; w < P.n_wg; ++w) {
        unsigned long long lo = S * w / n_wg, hi = S * (w + 1) / n_wg;
        wg_begin[w] = (unsigned)segs.size();
        unsigned long long pos = lo;
        while (pos < hi) {
            while (pos >= tile_start + tiles[ti].steps) {
                tile_start += tiles[ti].steps;
                ti++;
            }
            unsigned long long tend = tile_start + tiles[ti].steps;
            unsigned long long e = hi < tend ? hi : tend;
            TileSeg sg;
            sg.u = tiles[ti].u;
            sg.a = tiles[ti].a;
            sg.b = tiles[ti].b;
            sg.step_begin = (unsigned)(pos - tile_start);
            sg.step_end = (unsigned)(e - tile_start);
            if (sg.step_begin == 0 && sg.step_end == tiles[ti].steps) {
                sg.dest = -1;
                tile_direct[ti] = 1;
            } else {
                sg.dest = (int)n_slots;
                tile_slots[ti].push_back(n_slots);
                n_slots++;
            }
            segs.push_back(sg);
            pos = e;
        }
    }
    wg_begin[P.n_wg] = (unsigned)segs.size();
    std::vector<RedTile> red;
    std::vector<unsigned> red_slots;
    bool slots_consecutive = true;
    for (size_t i = 0; i < tiles.size(); ++i) {
        if (tile_direct[i] || tile_slots[i].empty()) continue;
        RedTile r;
        r.u = tiles[i].u;
        r.a = tiles[i].a;
        r.b = tiles[i].b;
        r.first = (unsigned)red_slots.size();
        r.count = (unsigned)tile_slots[i].size();
        r.slot0 = tile_slots[i][0];
        for (size_t q = 0; q < tile_slots[i].size(); ++q)
            if (tile_slots[i][q] != r.slot0 + q) slots_consecutive = false;
        for (unsigned sl : tile_slots[i]) red_slots.push_back(sl);
        red.push_back(r);
    }
    P.n_red = (unsigned)red.size();
    P.n_slots = n_slots;

    // All tables of a plan live in one slice of a persistent device arena and are uploaded with ONE stream-ordered
    // copy from the pinned host mirror of that slice: building a plan for a new shape costs no hipMalloc and no
    // host synchronisation (Genfer's supports grow statement by statement, so new shapes are the common case).
    size_t b_segs = segs.size() * sizeof(TileSeg), b_wg = wg_begin.size() * sizeof(unsigned);
    size_t b_red = red.size() * sizeof(RedTile), b_rs = red_slots.size() * sizeof(unsigned);
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t total = al(b_segs) + al(b_wg) + al(b_red) + al(b_rs) + 256;
    TableArena& A = arena();
    char *base = nullptr, *hbase = nullptr;
    if (A.ensure() && total <= A.bytes / 4) {
        if (A.head + total > A.bytes) {  // wrap: every cached plan's slice may be overwritten from now on
            (void)hipStreamSynchronize(st);   // pending uploads still read the host mirror
            for (auto& kv : plan_cache())
                if (kv.second.d_tables) (void)hipFree(kv.second.d_tables);
            plan_cache().clear();
            A.head = 0;
        }
        base = A.dev + A.head;
        hbase = A.host + A.head;
        A.head += total;
    } else {
        if (hipMalloc(&P.d_tables, total) != hipSuccess) return false;
        base = (char*)P.d_tables;
    }
    size_t off = 0;
    auto put = [&](const void* src, size_t bytes) -> void* {
        void* d = base + off;
        if (bytes) {
            if (hbase) std::memcpy(hbase + off, src, bytes);
            else (void)hipMemcpy(d, src, bytes, hipMemcpyHostToDevice);
        }
        off += al(bytes);
        return d;
    };
    T.segs = (const TileSeg*)put(segs.data(), b_segs);
    T.wg_begin = (const unsigned*)put(wg_begin.data(), b_wg);
    T.red = (const RedTile*)put(red.data(), b_red);
    T.red_slots = (const unsigned*)put(red_slots.data(), b_rs);
    if (slots_consecutive) T.red_slots = nullptr;  // k_conv_reduce derives the slots from RedTile::slot0
    if (hbase && hipMemcpyAsync(base, hbase, off, hipMemcpyHostToDevice, st) != hipSuccess) return false;
    return true;
}

// Which output blocks each wave of a workgroup accumulates (at most two: 16 accumulator VGPR pairs each).
// Block c costs one 8x8x8 chunk product per (x chunk, y chunk) pair that lands in it, i.e. c + 1 for full
// operands.  The classic pairing (c, nb-1-c) gives every ACTIVE wave the same load, but when nb < 2*NW the
// active waves sit unevenly on the CU's four SIMDs (wave w runs on SIMD w % 4): nb = 10 keeps two busy waves on
// SIMD 0 and one on the others, and the whole CU waits for SIMD 0 — measured 62% of the nb = 16 rate.
// Longest-processing-time assignment over SIMDs instead, then over the waves of the SIMD.
static void assign_blocks(TiledArgs& T, unsigned NW) {
    static const int mode = 1;
    for (int w = 0; w < 8; ++w) T.blk1[w] = T.blk2[w] = 0xff;
    // (classic pairs are only balanced when every block c costs c + 1: compact operands — the piece-split products'
    // untruncated inner axis costs min(c + 1, nxc, nyc, nb - c) — always take the LPT assignment)
    const bool full = T.nxc >= T.nb && T.nyc >= T.nb;
    if (mode == 0 || (full && (2 * NW == T.nb || NW < 4))) {
        for (unsigned w = 0; w < NW; ++w) {
            unsigned c1 = w, c2 = T.nb - 1 - w;
            if (c1 < T.nb && c1 <= c2) T.blk1[w] = (unsigned char)c1;
            if (c2 < T.nb && c2 > c1) T.blk2[w] = (unsigned char)c2;
        }
        return;
    }
    auto cost = [&](unsigned c) {
        unsigned n = 0;
        for (unsigned xc = 0; xc <= c && xc < T.nxc; ++xc)
            if (c - xc < T.nyc) n++;
        return n;
    };
    unsigned simd_load[4] = {0, 0, 0, 0}, wave_load[8] = {0}, wave_cnt[8] = {0};
    auto simd_of = [&](unsigned w) { return mode == 2 ? (w / 2) % 4 : w % 4; };
    for (unsigned i = 0; i < T.nb; ++i) {
        unsigned c = T.nb - 1 - i;  // heaviest first
        int best = -1;
        for (unsigned w = 0; w < NW; ++w) {
            if (wave_cnt[w] >= 2) continue;
            if (best < 0) { best = (int)w; continue; }
            unsigned sb = simd_load[simd_of((unsigned)best)], sw = simd_load[simd_of(w)];
            if (sw < sb || (sw == sb && wave_load[w] < wave_load[best])) best = (int)w;
        }
        if (wave_cnt[best] == 0) T.blk1[best] = (unsigned char)c;
        else T.blk2[best] = (unsigned char)c;
        wave_cnt[best]++;
        wave_load[best] += cost(c);
        simd_load[simd_of((unsigned)best)] += cost(c);
    }
}

template <int NW, int VAR, int TSH>
hipError_t launch_main_t(hipStream_t st, const Plan& P, const TiledArgs& T) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_tiled<NW, VAR, TSH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    // (a failed launch is latched by the launch thread and raised by the next drain: gft_launch.hpp lq_note)
    GFT_LAUNCH((k_conv_tiled<NW, VAR, TSH>), dim3(P.n_wg), dim3(NW * 64), P.lds_bytes, st, T);
    return hipSuccess;
}
template <int NW, int VAR>
hipError_t launch_main(hipStream_t st, const Plan& P, const TiledArgs& T) {
    return T.tsh == 3 ? launch_main_t<NW, VAR, 3>(st, P, T) : launch_main_t<NW, VAR, 0>(st, P, T);
}

}  // namespace

void tiled_pad_rows_f64(hipStream_t st, const double* in, double* out, size_t rows, unsigned len, unsigned P, unsigned B) {
    size_t tot = rows * P * B;
    if (!tot) return;
    GFT_LAUNCH(k_pad_rows, dim3((unsigned)std::min<size_t>((tot + 255) / 256, 4096)), dim3(256), 0, st, in, out, rows,
                       len, P, B);
}
void tiled_fold_rows_f64(hipStream_t st, const double* zt, double* z, size_t rows, size_t row_lo, size_t row_hi, unsigned Pz,
                         unsigned B, unsigned zI, int accumulate, const unsigned* guard, unsigned epoch) {
    size_t tot = (row_hi - row_lo) * zI;
    if (!tot) return;
    GFT_LAUNCH(k_fold_rows, dim3((unsigned)std::min<size_t>((tot + 255) / 256, 4096)), dim3(256), 0, st, zt, z, rows,
                       row_lo, row_hi, Pz, B, zI, accumulate, guard, epoch);
}


void tiled_set_lane_tile(int tsh) { tiled_force_tsh() = tsh; }

bool conv_tiled_f64(hipStream_t st, const double* x, const double* y, double* z, const ConvArgs& a_in, void* ws,
                    size_t ws_bytes, size_t* ws_needed, unsigned* nf_flag, unsigned nf_epoch, bool* guarded) {
    ConvArgs a = a_in;
    if (a.variant < 0) a.variant = GFT_TILED_DEFAULT_VARIANT;
    if (a.nd >= 2 && a.nd <= 4) {  // the pipelined fast path (bits 1|2) needs x and y to span every chunk of z's inner axis
        unsigned nb = (a.zs[a.nd - 1] + 7) / 8;
        if ((a.xs[a.nd - 1] + 7) / 8 < nb || (a.ys[a.nd - 1] + 7) / 8 < nb) {
            static const bool compact_fast = true;
            if (compact_fast && (a.variant & 3) == 3) a.variant = (a.variant & ~0xff) | (a.variant & 7) | 8;  // pipelined path for compact operands
            else a.variant &= ~3;
        }
    }
    PlanKey key;
    std::memset(&key, 0, sizeof(key));
    if (a.nd < 2 || a.nd > 4) return false;
    key.v[0] = (unsigned)a.nd | ((unsigned)a.slab_axis << 8) | ((unsigned)tiled_force_tsh() << 16);
    for (int i = 0; i < a.nd; ++i) {
        key.v[1 + i] = a.xs[i];
        key.v[5 + i] = a.ys[i];
        key.v[9 + i] = a.zs[i];
    }
    key.v[13] = a.slab_lo;
    key.v[14] = a.slab_hi;
    key.v[15] = (unsigned)a.accumulate;
    key.v[16] = (unsigned)(a.j0_min | (a.j0_excl << 8) | (a.j0_desc << 16));
    key.v[17] = (unsigned)a.variant;
    auto& cache = plan_cache();
    auto it = cache.find(key);
    if (it == cache.end()) {
        Plan P;
        if (cache.size() >= 1024) {  // bounded: drop everything (plans are cheap to rebuild) and reuse the arena
            (void)hipStreamSynchronize(st);
            for (auto& kv : cache)
                if (kv.second.d_tables) (void)hipFree(kv.second.d_tables);
            cache.clear();
            arena().head = 0;
        }
        if (!build_plan(a, P, st)) return false;  // may itself reset the cache when the table arena wraps
        it = cache.emplace(key, P).first;
    }
    const Plan& P = it->second;
    const TiledArgs& B = P.base;
    // workspace: [partial slabs][packed x][packed y]
    size_t b_slots = (size_t)P.n_slots * B.nb * 512 * sizeof(double);
    size_t x_rows = (size_t)B.xU * B.x0 * B.x1, y_rows = (size_t)B.yU * B.y0 * B.y1;
    // Operands are read in place where their own layout IS the packed one (round 4): rows of whole chunks, y 16-byte
    // aligned (window loads are 16-byte), and 64 readable bytes after x's last element (the pipelined path requests one
    // chunk beyond the one it uses; the library's own buffers have that slack, a caller's raw pointer may not), both spanning
    // every chunk of the result's rows.  There is then no zero padding anywhere, so inf / NaN operands cannot create NaNs the reference does not produce: no verdict,
    // no guarded fallback launch — ONE launch per product.  Everything else is packed by k_prep_operands (rows padded to
    // whole chunks plus one chunk of slack after the last row), which also takes the non-finite verdict.
    static const bool inplace_env = true;
    // (full inner extents only: the compact-operand path multiplies a zero window where a block reaches beyond y's last chunk —
    // artificial zeros again, which need the verdict)
    const bool inplace = inplace_env && a.operands_slack && B.nx8 == B.xI && B.ny8 == B.yI && B.nxc >= B.nb && B.nyc >= B.nb &&
                         !((uintptr_t)y & 15) && !((uintptr_t)x & 7);
    if (guarded) *guarded = !inplace;
    bool pack_y = !inplace && (B.ny8 != B.yI || ((uintptr_t)y & 15));  // window loads are 16-byte
    // (sized for the packed layout whether or not this call packs: a query and the launches that follow it — the high-rank
    // loop's operand blocks, say — must agree on the workspace whatever their pointers' alignment)
    size_t b_xp = (x_rows * B.nx8 + 8) * sizeof(double);
    size_t b_yp = y_rows * B.ny8 * sizeof(double);
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t need = al(b_slots) + al(b_xp) + al(b_yp) + 256;
    if (ws_needed) *ws_needed = need;
    if (!ws) return true;  // query
    if (ws_bytes < need) return false;

    char* wb = (char*)ws;
    TiledArgs T = B;
    T.ws = (double*)wb;
    double* xp = (double*)(wb + al(b_slots));
    double* yp = (double*)(wb + al(b_slots) + al(b_xp));
    if (inplace) {
        T.xp = x;
        T.yp = y;
        T.guard = nullptr;
        T.guard_epoch = 0;
    } else {
        size_t tot = x_rows * B.nx8 + (pack_y ? y_rows * B.ny8 : y_rows * B.yI);
        GFT_LAUNCH(k_prep_operands, dim3((unsigned)std::min<size_t>((tot + 255) / 256, 2048)), dim3(256), 0, st, x,
                           xp, x_rows, B.xI, B.nx8, y, pack_y ? yp : nullptr, y_rows, B.yI, B.ny8, nf_flag, nf_epoch);
        T.xp = xp;
        T.yp = pack_y ? yp : y;
        T.guard = nf_flag;
        T.guard_epoch = nf_epoch;
    }
    T.z = z;
    T.xcd_remap = ((a.variant & 4) && (P.n_wg % 8 == 0)) ? 1u : 0u;
    hipError_t e = hipSuccess;
    constexpr int DEF = GFT_TILED_DEFAULT_VARIANT;
    int variant = a.variant;
    if (variant & 8) {  // compact operands, pipelined
        switch (P.NW) {
            case 1: e = launch_main<1, 11>(st, P, T); break;
            case 2: e = launch_main<2, 11>(st, P, T); break;
            case 4: e = launch_main<4, 11>(st, P, T); break;
            default: e = launch_main<8, 11>(st, P, T); break;
        }
    } else if (P.NW == 8) {
        switch (variant) {
            case 0: case 4: e = launch_main<8, 0>(st, P, T); break;
            case 1: e = launch_main<8, 1>(st, P, T); break;
            case 3: e = launch_main<8, 3>(st, P, T); break;
            case 7: e = launch_main<8, 3>(st, P, T); break;  // 3 + XCD-contiguous ranges (runtime flag)
#ifdef GFT_TILED_DIAG
            case 17: e = launch_main<8, 17>(st, P, T); break;
            case 33: e = launch_main<8, 33>(st, P, T); break;
            case 49: e = launch_main<8, 49>(st, P, T); break;
            case 65: e = launch_main<8, 65>(st, P, T); break;
            case 113: e = launch_main<8, 113>(st, P, T); break;
#endif
            default: return false;
        }
    } else {
        if (variant != 0 && variant != 4 && variant != DEF && variant != (DEF & 3)) return false;
        if (variant & 1) {
            switch (P.NW) {
                case 1: e = launch_main<1, (DEF & 3)>(st, P, T); break;
                case 2: e = launch_main<2, (DEF & 3)>(st, P, T); break;
                default: e = launch_main<4, (DEF & 3)>(st, P, T); break;
            }
        } else {
            switch (P.NW) {
                case 1: e = launch_main<1, 0>(st, P, T); break;
                case 2: e = launch_main<2, 0>(st, P, T); break;
                default: e = launch_main<4, 0>(st, P, T); break;
            }
        }
    }
    if (e != hipSuccess) return false;
    if (P.n_red) {
        GFT_LAUNCH(k_conv_reduce, dim3(P.n_red, T.nb), dim3(256), 0, st, T, P.n_red);
    }
    return true;
}

}  // namespace gft
