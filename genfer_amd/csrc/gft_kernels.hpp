// Launcher declarations for the HIP kernels (definitions in gft_kernels.hip / gft_conv_tiled.hip).
// Everything operates on contiguous row-major device tensors ("views") of f64 or interval
// (two-plane) elements.  No launcher synchronises the stream.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <vector>

#include "gft_elem.hpp"
#include "gft_launch.hpp"  // GFT_LAUNCH: every kernel launch of the library (counted, issued by the launch thread)

namespace gft {

constexpr int MAXD = 12;  // max tensor rank after unit-axis collapsing (reference programs: <= 8 vars)

struct Shape {
    int nd;
    unsigned d[MAXD];
};

// A contiguous row-major tensor (or sub-block starting at p) in device memory.  For interval
// tensors the hi plane lives `plane` doubles after the lo plane.
struct DView {
    double* p;
    size_t plane;
    Shape sh;
};

enum GatherOp { OP_COPY = 0, OP_MUL_S = 1, OP_DIV_S = 2, OP_NEG = 3, OP_MUL_TAB = 4, OP_LMUL_S = 5, OP_MUL_POW = 6,
                OP_MUL_TAB_LMUL_S = 7 /* s * (x * tab[k]): derivative scaling, then a constant factor on the left */,
                OP_MUL_HTAB = 8 /* x * tab[k] with tab a HOST array of at most HTAB_CAP / W entries per plane: it travels to the
                                   device kernel as a kernel argument (no table launch, no per-thread running product) */ };
constexpr unsigned HTAB_CAP = 384;  // doubles in the by-value table (intervals: two planes of HTAB_CAP / 2)
struct HostTab {
    double v[HTAB_CAP];
};
enum MapOp { MAP_NEG = 0, MAP_DIV_U32 = 1, MAP_MUL_U32 = 2, MAP_MUL_S = 3, MAP_DIV_S = 4, MAP_LMUL_S = 5 };
enum FirstOp { FIRST_ADD = 0, FIRST_SUB = 1, FIRST_SUB_NEG_ALL = 2 };
enum BlockOp { BLK_ADD = 0, BLK_ADD_U32_TIMES = 1, BLK_ASSIGN = 2 };
enum ImmOp { IMM_LMUL = 0 /* b*a */, IMM_MUL = 1 /* a*b */, IMM_DIV = 2, IMM_NEG = 3, IMM_ADD = 4, IMM_SUB = 5, IMM_SUB_NEG = 6 /* -(a-b) */ };
enum ScalarOp { SC_DIV = 2 };
enum TableOp { TAB_DERIV = 0, TAB_COEFF = 1, TAB_POW = 2, TAB_INDEX = 3 };

struct GatherArgs {
    Shape out;                  // output (contiguous) shape
    int shift[MAXD];            // src index along axis a = k_a + shift[a]
    unsigned src_len[MAXD];     // valid src index range [0, src_len[a]); outside -> zero
    size_t src_stride[MAXD];    // src element strides
    int op;                     // GatherOp
    Scalar2 s;
    int tab_axis;               // axis whose OUTPUT index selects tab[] / keep[]
    const double* tab;          // OP_MUL_TAB: x * tab[k_axis]  (interval: two planes, tab_plane apart);
                                // OP_MUL_POW: x * m^k_axis with m = tab[0], the power formed as ((1*m)*m)*.. in the thread
                                // (the reference's running product, mt:557-565: no table launch for short axes)
    size_t tab_plane;
    const unsigned char* keep;  // optional: keep[k_axis] == 0 -> write zero
};

// ---- deferred elementwise chains (round 3) ---------------------------------------------------------------------------
// Genfer's evaluator emits runs of cheap elementwise operations on one tensor — `subst_var` by a pure scaling m*x_v
// (x * m^k along v, mt:557-565), `+ const` (element 0, mt:862-869), `* const` (mt:1041-1047), truncation (mt:183-204) —
// that end in an Add of two such results (the two arms of an `if`, generating_function.rs:557-566).  One launch per
// operation is 4 us of nothing: the host keeps such a result as a CHAIN — a leading (prefix) box of a contiguous base
// tensor plus up to CHAIN_MAX elementwise stages — and the consuming kernel applies the stages to each element it
// loads, in the recorded order, with the same functors: the same roundings in the same order as one launch per stage.
constexpr int CHAIN_MAX = 6;
enum ChainKind {
    CH_LMUL_S = 0,             // s * x
    CH_MUL_S = 1,              // x * s
    CH_DIV_S = 2,              // x / s
    CH_NEG = 3,                // -x
    CH_FIRST_ADD = 4,          // element 0: x + s
    CH_FIRST_SUB = 5,          // element 0: x - s
    CH_FIRST_SUB_NEG_ALL = 6,  // element 0: -(x - s), every other element: -x
    CH_MUL_TAB = 7,            // x * tab[k_axis]  (device table, tab_plane apart for intervals)
};
struct ChainStage {
    int kind, axis;
    Scalar2 s;
    const double* tab;
    size_t tab_plane;
};
struct ChainSrc {
    const double* p;           // base tensor (contiguous)
    size_t plane;
    unsigned box[MAXD];        // extent of the operand's data per output axis (base coordinates): outside it the operand is zero
    int pad[MAXD];             // output index k reads base index k - pad (>= 0: zeros in front — mul_var's shift, mt:589-608)
    size_t stride[MAXD];       // base strides per output axis
    int nstages;
    ChainStage st[CHAIN_MAX];
};

// An operand of a NESTED chain add (round 5): either a chain `a`, or a RECORDED sum (0 + a) (+|-) b of two chains over its own
// box — what k_chain<E, true> would have written to memory — with the stages `post` a view of that sum carried.  An `if`
// whose arms both end in mul_linear (State ~ Bernoulli(p) in hmm: c * t + m * shift(t), itself a two-chain add) and the Add of
// the arms are then ONE launch of four sources instead of three launches; per element the operations of the three.
struct NestSrc {
    int nested, sub_inner;
    ChainSrc a, b;
    unsigned box[MAXD];
    int npost;
    ChainStage post[CHAIN_MAX];
};

// Fused observation step (generating_function.rs:678-700): out = c * ((D * 1)>>v + x * D) with
// D = derivative(a, v, 1) truncated, i.e. the reference sequence derivative -> truncate -> * (x + eps_v)
// (mul_linear: mul_var + constant scale + add) -> * c, element for element in the same operation order.
struct ObserveArgs {
    Shape out;                 // result shape (collapsed)
    unsigned d_len[MAXD];      // shape of D' (the truncated derivative) per axis
    size_t a_stride[MAXD];     // strides of the input tensor a
    int axis;                  // collapsed index of v
    Scalar2 x, c;
    int x_is_zero, x_is_one, c_is_one;
    const double* tab;         // derivative factors ff_j (mt:472-478), tab_plane apart for intervals
    size_t tab_plane;
};

// A whole chain of observation steps (the loop of generating_function.rs:684-689: `order` times derivative -> truncate ->
// * (x + eps_v) -> * c_k) in ONE launch: the steps couple positions along v only, so every line along v runs all of them
// on its own (one workgroup per line, intermediates in LDS), and only the lines inside the FINAL truncated box are
// computed at all.  Per step and element exactly ObserveArgs' operations.
constexpr int OC_MAX = 16;  // steps per launch (longer chains are cut by the host; round 6: 48 -> 16 — the argument block of a chain is copied
                            // four times on its way into a batch, and programs observe counts of 0 .. 12)
struct ObserveChainArgs {
    int nd;                    // collapsed rank of the final shape
    unsigned fs[MAXD];         // final shape
    size_t a_stride[MAXD];     // strides of the input tensor
    size_t o_stride[MAXD];     // strides of the output tensor
    int axis;                  // collapsed index of v
    unsigned nsteps, len0;     // steps; input length along v
    unsigned dl[OC_MAX];       // per step: length of the truncated derivative along v
    unsigned lo[OC_MAX];       // per step: length of the step's result along v
    Scalar2 x;
    Scalar2 c[OC_MAX];
    unsigned long long c_one;  // bit t: c[t] is exactly one
    int x_is_zero, x_is_one;
    const double* tab;         // derivative factors ff_j (mt:472-478), tab_plane apart for intervals
    size_t tab_plane;
    unsigned lw_pad;           // LDS line pitch (>= the longest line)
};

// Epilogue of an observation chain that was recorded lazily and is consumed by an Add / Sub of two tensors (round 5): the
// kernel applies the recorded elementwise stages `post` to each finished value X (the chain that sat on top of the
// observation's result), evaluates the OTHER operand's chain Y at the same output index and stores the sum — per element
// exactly k_chain<E, true>'s operations: mode 1 = (0 + Y) (+|-) X (the observation is the right operand), mode 2 =
// (0 + X) (+|-) Y.  The output has the observation's own shape, so every element receives its X.
struct ObsEpi {
    int mode;                  // 0: none (plain store)
    int subtract;
    int npost;
    ChainStage post[CHAIN_MAX];  // axis: index into the kernel's collapsed axes
    ChainSrc y;                // over the kernel's collapsed axes
};

// One Horner step of subst_var with a LINEAR substitution s = c + m*eps_w (mt:566-579, 589-623, 873-880):
//   out = res * s + coeff_i,   coeff_i = a[.., i, ..] along the substituted axis,
// computed per element in exactly the order of the reference's op sequence
//   A = mul_var(res, m, w);  B = res * c (skipped if c == 1);  P = (0 + A) + B (or P = A if c == 0);
//   out = (0 + P) + coeff_i   (or P with element 0 += coeff_i when the slab is a single coefficient)
// in ONE launch instead of gather + gather + gather + addsub + addsub, and without the per-step
// extract_linear(res) round trip.
struct HornerArgs {
    Shape out;                 // result shape (collapsed)
    unsigned rs[MAXD];         // shape of res
    unsigned sh[MAXD];         // shape of P = res * s  (rs with axis w grown by one, capped by degrees_p1)
    unsigned oc[MAXD];         // box of coeff_i (1 along the substituted axis)
    size_t rstr[MAXD];         // strides of res
    size_t astr[MAXD];         // strides of a
    size_t a_base;             // i * stride of the substituted axis in a
    int w;                     // collapsed index of the substitution's variable
    unsigned upper;            // mul_var: source indices 0 .. upper-1 along w
    Scalar2 c, m;
    int c_zero, c_one, coeff_scalar;
    const unsigned* guard;     // optional (device kernel only): nothing is done if *guard != 0 — see HornerLoopArgs::guard
};

// ALL remaining Horner steps of a subst_var with a linear substitution in ONE launch: the recursion couples
// positions along the substitution axis w only, so each line along w is one workgroup that runs every step on
// its own (per-step shapes re-derived on the device, intermediates in LDS).  Same per-element operations as
// HornerArgs.
struct HornerLoopArgs {
    int nd;                    // collapsed rank (axes where the final shape is > 1)
    unsigned deg[MAXD];        // degrees_p1, clamped to 2^31
    unsigned rs0[MAXD];        // shape of the incoming accumulator
    unsigned fs[MAXD];         // final shape
    unsigned lw_pad;           // LDS line pitch (>= fs[w])
    unsigned oc[MAXD];         // box of a coefficient slab
    size_t rstr0[MAXD];        // strides of the incoming accumulator (compact)
    size_t fstr[MAXD];         // strides of the final shape (all in-kernel intermediates use these)
    size_t astr[MAXD];         // strides of a (0 on the substituted axis)
    size_t a_vstride;          // stride of the substituted axis in a
    int w;
    unsigned first_i, nsteps;  // coefficient index of the first in-kernel step; steps i = first_i, first_i-1, ...
    Scalar2 c, m;
    int c_zero, c_one, coeff_scalar;
    int diag;                  // timing diagnostics (GFT_HORNER_DIAG, wrong results): 2 = no barrier
    unsigned long long* stat;  // GFT_HORNER_DIAG & 64: {lean wave-steps, wave-steps} of the POINT pipeline are added here
    const unsigned* guard;     // optional: the launch does nothing if *guard != 0 — the verdict word of the linear scan queued just
                               // before it said "the accumulator IS linear", i.e. the speculation this launch embodies failed
};

// A whole linear Horner loop as an item of a batch (K<E>::horner_batch): everything its launch needs.
struct alignas(16) HornerRider {
    const double* res0;
    size_t rp0;
    const double* a;
    size_t ap;
    double* out;
    size_t plane;
    HornerLoopArgs g;
    unsigned lines;
};

// ---- batched launches (round 6) ------------------------------------------------------------------------------------------
// A Genfer program evaluates the same statement at many input points that do not depend on each other (the d + 1 points of
// depth d of the memoised recursion, generating_function.rs:186-222): the host records such operations and issues all the
// independent ones of one kind as ONE launch (gft_api.hip, "deferred launch graph").  A batch is an array of ITEMS in device
// memory — each the argument block the unbatched kernel takes — and a two-dimensional grid: blockIdx.y = item, blockIdx.x =
// what blockIdx.x is in the unbatched launch.  A workgroup copies its item to LDS first (as the unbatched kernels copy their
// kernel-argument segment) and then runs the unbatched kernel's body on it: the same instructions on the same data, so the
// same bits as one launch per item.
struct alignas(16) ObsItem {               // one observation chain (K<E>::observe_chain_multi without a rider); epi.mode == 0: plain store
    const double* a;
    size_t ap;
    double* out;
    size_t op;
    ObserveChainArgs g;
    ObsEpi epi;
    unsigned lines, longest;
};
struct alignas(16) ChainItem {             // out = chain(a) (two == 0) or (0 + a) (+|-) b (K<E>::chain_copy / chain_addsub)
    double* out;
    size_t out_plane;
    Shape sh;
    ChainSrc a, b;
    int subtract, two;
    size_t total;
};
struct alignas(16) NestItem {              // K<E>::chain_nest
    double* out;
    size_t out_plane;
    Shape sh;
    NestSrc a, b;
    int subtract;
    unsigned total;
};
// (a whole linear Horner loop on the POINT pipeline: HornerRider is the item)

// Host mailbox in mapped, coherent pinned memory: a kernel writes up to 7 doubles of payload and then the
// sequence number (system-scope release); the host polls the sequence word instead of paying a D2H copy launch
// plus hipStreamSynchronize (22 -> ~7 us per host round trip, see tools/bench_sync.py).
struct Mailbox {
    double* payload;              // device-visible address of the pinned slot (8 doubles)
    unsigned long long* seq;      // payload + 8
    unsigned long long value;     // sequence number this kernel must publish
    unsigned* dev_word;           // optional: the linear scans also store their verdict mask here, in DEVICE memory, for kernels
                                  // queued behind them that must not run if the speculation they embody failed (HornerLoopArgs::guard)
};
__device__ inline void mailbox_publish(const Mailbox& mb) {
    __threadfence_system();
    __hip_atomic_store(mb.seq, mb.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// copies n <= 7 doubles (src[i * stride]) into the mailbox
void peek_to_mailbox(hipStream_t st, const double* src, size_t stride, unsigned n, const Mailbox& mb);
// payload[0] = number of zero words among flags[0 .. n): the verdict of a speculative Horner loop (K<E>::witness)
void witness_verdict(hipStream_t st, const unsigned* flags, unsigned n, const Mailbox& mb);
// dst[0 .. n) = src[0 .. n) for a host-tier tensor that meets a device operand: the values travel as kernel
// arguments (480 doubles per launch), so the host block may be reused the moment this returns
void upload_small(hipStream_t st, double* dst, const double* host_src, size_t n);

struct ConvArgs {
    int nd;
    unsigned xs[MAXD], ys[MAXD], zs[MAXD];
    size_t xstr[MAXD], ystr[MAXD], zstr[MAXD];
    unsigned slab_lo, slab_hi;  // output range on axis 0
    int slab_axis;              // tiled kernel only: the axis the range applies to instead of 0 (piece-split problems)
    int accumulate;             // 0: start from zero, 1: start from the stored value
    int j0_min;                 // exp/log recurrences start at j = 1
    int j0_excl;                // 1: exclude j0 == k0 (div/log: res[k] is not known yet)
    int j0_desc;                // iterate j0 downwards (log's summation order)
    int inner_from_zero;        // last axis' partial sum is formed from zero, then added (mul_1d, mt:971-982)
    int variant;                // tiled-kernel variant (gft_set_conv_variant; -1 = library default)
    int operands_slack;         // 64 bytes after x's last element are readable (the library's own buffers): the tiled kernel's in-place operands, the row-pair kernel's last partial x chunk
    const unsigned* guard;      // optional device word: the reference-order kernels run only if *guard == guard_epoch
    unsigned guard_epoch;       // (fallback for non-finite operands of the tiled kernel, decided on the device)
};

// What a shallow product's kernel does with each finished sum (K<E>::conv_shallow): the Add that follows the product in
// a general Horner step  res * subst + slab_i  (mt:569-579), applied where the sum stands.
struct ConvEpi {
    int mode;              // 0: out = prod; 1: out[k] = ((0 + prod[k])? + slab[k]?) on the boxes zs / abox of the output
                           // shape os (mt:873-880); 2: out = prod with element 0 = prod[0] + slab[0] (mt:862-869)
    unsigned os[MAXD];     // output shape (collapsed axes; modes 0 / 2: == zs)
    unsigned abox[MAXD];   // slab box (mode 1)
    size_t astr[MAXD];     // slab strides in its source tensor (mode 1)
    const double* ap;      // slab element 0 (modes 1, 2); interval: hi plane `aplane` doubles further
    size_t aplane;
    unsigned* wit;         // optional: *wit = 1 if the result has a non-linearity witness (K<E>::witness's predicate)
};

template <class E>
struct K {
    // out[k] = f(src[k + shift]) or zero outside the source box
    static void gather(hipStream_t st, const double* src, size_t src_plane, double* out, size_t out_plane,
                       const GatherArgs& a);
    // out[k] = chain(base[k]) inside the operand's box, zero outside: materialises a deferred chain
    static void chain_copy(hipStream_t st, double* out, size_t out_plane, const Shape& out_shape, const ChainSrc& a);
    // the same plus linear_scan's verdict on the materialised tensor, in one launch (state / mailbox as linear_scan)
    static void chain_copy_scan(hipStream_t st, double* out, size_t out_plane, const Shape& out_shape, const ChainSrc& a, unsigned axes_mask,
                                unsigned* state, const Mailbox& mb);
    // out[k] = ((0 + A[k]?) +/- B[k]?) like addsub_padded, A / B deferred chains evaluated on the fly
    static void chain_nest(hipStream_t st, double* out, size_t out_plane, const Shape& out_shape, const NestSrc& a, const NestSrc& b, int subtract);
    static void chain_addsub(hipStream_t st, double* out, size_t out_plane, const Shape& out_shape, const ChainSrc& a,
                             const ChainSrc& b, int subtract);
    // out[k] = ((0 + a[k]?) +/- b[k]?) with a, b leading blocks of out's shape  (mt:873-880, 927-934)
    static void addsub_padded(hipStream_t st, const DView& out, const DView& a, const DView& b, int subtract);
    // out[k] = (0 + a[k]?) + (c * b[k])?   — Add of a and the constant multiple c * b in one pass (mt:873-880 after mt:1041-1047)
    static void add_scaled_padded(hipStream_t st, const DView& out, const DView& a, const DView& b, Scalar2 c);
    // dst = copy of src (n contiguous elements) with element 0 replaced by src[0] (+|-) s; FIRST_SUB_NEG_ALL:
    // dst[0] = -(src[0] - s), dst[i>0] = -src[i]  (mt:862-868, 919-925 in one launch)
    static void copy_first(hipStream_t st, const double* src, size_t src_plane, double* dst, size_t dst_plane, size_t n,
                           int op, const double* s, size_t s_plane, Scalar2 s_value);  // s == nullptr: use s_value
    // p[0] = v0, p[1] = v1 (if n == 2): constants and `var` constructors without a host->device copy
    static void set_small(hipStream_t st, double* p, size_t plane, unsigned n, Scalar2 v0, Scalar2 v1);
    // extract_linear (mt:275-294) for all axes in ONE launch.  Bit a of the mask survives iff the tensor is
    // "linear in axis a" (every non-zero entry sits at index 0 or at e_a); blocks AND their verdicts into
    // state[0] and the last block to arrive (ticket in state[1]) writes out[0] = mask (as a double),
    // out[1..2] = coeffs[0], out[3..4] = coeffs[e_v] for the first surviving axis v — one 40-byte read-back.
    // `state` must be {0xffffffff, 0} on entry; the last block restores it, so back-to-back calls on one
    // stream need no memset.
    static void linear_scan(hipStream_t st, const DView& t, unsigned axes_mask, unsigned* state, const Mailbox& mb);
    static void horner_linear(hipStream_t st, const double* res, size_t res_plane, const double* a, size_t a_plane, double* out,
                              size_t out_plane, const HornerArgs& args);
    static constexpr unsigned HORNER_LINE_MAX = 2048;  // longest line along the substitution axis of horner_linear_loop
    // `wit` (optional): wit[t] = 1 if the accumulator after in-kernel step t < nsteps-1 has a non-linearity witness
    static void horner_linear_loop(hipStream_t st, const double* res0, size_t res0_plane, const double* a, size_t a_plane,
                                   double* out, size_t plane, const HornerLoopArgs& args, unsigned lines, unsigned* wit);
    // the loop runs on the POINT pipeline without a guard word: it can be an item of a batch (horner_batch)
    static bool horner_can_ride(const HornerLoopArgs& args);
    // *flag = 1 if `t` holds a non-zero coefficient at an index with two non-zero coordinates or a coordinate >= 2:
    // such a tensor is not of the form c + m*x_v (extract_linear, mt:275-294, would say None).  Sticky, no read-back.
    static void witness(hipStream_t st, const DView& t, unsigned* flag);
    static void observe_step(hipStream_t st, const double* a, size_t a_plane, double* out, size_t out_plane,
                             const ObserveArgs& args);
    static constexpr unsigned OBSERVE_LINE_MAX = 2048;  // longest line along v of observe_chain
    // `epi` (optional): the consumer's Add as the chain's epilogue
    static void observe_chain_epi(hipStream_t st, const double* a, size_t a_plane, double* out, size_t out_plane,
                                  const ObserveChainArgs& args, unsigned lines, unsigned longest, const ObsEpi* epi);
    static void observe_chain(hipStream_t st, const double* a, size_t a_plane, double* out, size_t out_plane,
                              const ObserveChainArgs& args, unsigned lines, unsigned longest);
    // ---- batches (see ObsItem): `items` is DEVICE memory holding n items that share the launch geometry the *_geometry
    // functions report (the host groups by it); n >= 1
    struct Geometry {
        unsigned gx = 0, threads = 0;   // grid.x, workgroup size
        size_t lds = 0;                 // dynamic LDS bytes
        int variant = 0;                // which instantiation (kernel-specific)
        bool ok = true;                 // false: this item cannot run in a batch (launch it on its own)
        bool operator==(const Geometry& o) const { return gx == o.gx && threads == o.threads && lds == o.lds && variant == o.variant; }
    };
    static Geometry observe_chain_geometry(unsigned lines, unsigned longest, unsigned lw_pad, unsigned nsteps, bool epi);
    static void observe_chain_batch(hipStream_t st, const ObsItem* items, unsigned n, const Geometry& g);
    static Geometry chain_geometry(const ChainItem& it);
    static void chain_batch(hipStream_t st, const ChainItem* items, unsigned n, const Geometry& g);
    static Geometry chain_nest_geometry(size_t total);
    static void chain_nest_batch(hipStream_t st, const NestItem* items, unsigned n, const Geometry& g);
    static Geometry horner_geometry(const HornerLoopArgs& g, unsigned lines);
    static void horner_batch(hipStream_t st, const HornerRider* items, unsigned n, const Geometry& g);
    // in-place elementwise map over n contiguous elements
    static void map_inplace(hipStream_t st, double* p, size_t plane, size_t n, int op, unsigned u, Scalar2 s);
    // like map_inplace with MAP_*_S but the scalar is read from device memory (s_ptr[0], s_ptr[s_plane])
    static void map_inplace_dev(hipStream_t st, double* p, size_t plane, size_t n, int op, const double* s_ptr,
                                size_t s_plane);
    // dst (contiguous, shape D) leading block of src's shape: BLK_ADD r += x ; BLK_ADD_U32_TIMES r += u*x ; BLK_ASSIGN r = x
    static void block_op(hipStream_t st, const DView& dst, const DView& src, int op, unsigned u);
    // 0-dim op: out = a / b (SC_DIV)
    static void scalar_op(hipStream_t st, int op, const double* a, size_t a_plane, const double* b, size_t b_plane,
                          double* out, size_t out_plane);
    // sequential 1-D recurrences (mt:1270-1283, 1319-1333)
    // `seed` = exp(xs[0]) / ln(xs[0]), formed by the caller on the host (one libm value)
    static void exp_1d(hipStream_t st, const double* xs, size_t x_plane, unsigned nx, double* res, size_t r_plane,
                       unsigned n, Scalar2 seed);
    static void log_1d(hipStream_t st, const double* xs, size_t x_plane, unsigned nx, double* res, size_t r_plane,
                       unsigned n, Scalar2 seed);
    // last-axis level of the division recurrence (mt:1162-1192 with 0-dim base): res = xs / ys, 1-D
    // 1-d base case of the division recurrence; fused: the dividend is (-res) (+ xs inside its length) taken in place.
    // false (fused only): shape outside the lock-step kernel — the caller prepares the dividend itself
    static bool div_1d(hipStream_t st, const double* xs, size_t x_plane, unsigned nx, const double* ys,
                       size_t y_plane, unsigned ny, double* res, size_t r_plane, unsigned n, int fused = 0);
    // The last TWO axes of the division recurrence in one launch (gft_div2d.hip): res[n1, n2] = dividend / y[ny1, ny2],
    // bit-identical to the host-driven recursion.  fused == 0: dividend = x (box nx1 x nx2, row stride x_rstride);
    // fused == 1: dividend = (-res) (+ x inside its box), res holding the leading-axis partial sums on entry;
    // fused == 2: log's slab step — dividend = (-res) (+ log_k * x), stores res = q / log_k and res2 = res * log_k
    // (mt:1186-1189).  Returns false (nothing launched) if the slab does not fit one workgroup's LDS.
    static bool div_2d(hipStream_t st, const double* x, size_t x_plane, unsigned nx1, unsigned nx2, size_t x_rstride, const double* y,
                       size_t y_plane, unsigned ny1, unsigned ny2, double* res, size_t r_plane, unsigned n1, unsigned n2, int fused,
                       unsigned log_k = 0, double* res2 = nullptr, size_t r2_plane = 0);
    // The whole quotient res = xs / ys (ranks 2-4, rows of at most 64 coefficients) as a row wavefront in ONE launch
    // (gft_div2d.hip): bit-identical to the host-driven recursion.  `flags_and_counter`: (rows + 1) zeroed device words.
    // false: shape outside the kernel's domain, nothing launched.
    static bool div_wavefront(hipStream_t st, const double* xs, size_t x_plane, const unsigned* xshape, const double* ys, size_t y_plane,
                              const unsigned* yshape, double* res, size_t r_plane, const unsigned* rshape, int nd, unsigned* flags_and_counter);
    // res[1..] = log(xs)[1..]: the slabs k0 >= 1 of the log recurrence as the same row wavefront (slab 0 is the caller's)
    static bool log_wavefront(hipStream_t st, const double* xs, size_t x_plane, const unsigned* xshape, double* res, size_t r_plane,
                              const unsigned* rshape, int nd, double* qbuf, size_t q_plane, unsigned* flags_and_counter);
    // res[1..] = exp(xs)[1..] likewise (slab 0, an exp one dimension down, is the caller's and complete in stream order)
    // `arrival_order`: the source slabs of each row's sum in descending j0 (1e-10 contract instead of the reference's order)
    static bool exp_wavefront(hipStream_t st, const double* xs, size_t x_plane, const unsigned* xshape, double* res, size_t r_plane,
                              const unsigned* rshape, int nd, unsigned* flags_and_counter, int arrival_order = 0);
    // Rank-2 quotient (mode 0), or the rows >= 1 of a rank-2 log (1) / exp (2), with rows of 65 .. 4096 coefficients as a
    // coefficient-level wavefront in ONE launch (gft_div2d.hip k_rows_wavefront): tasks are 64-coefficient segments of rows,
    // bit-identical to the host-driven recursion.  `flags_and_counter`: rows * ceil(row length / 64) + 1 zeroed words;
    // `qbuf` (mode 1): a tensor like res.  false: outside the kernel's domain, nothing launched.
    static bool rows_wavefront(hipStream_t st, int mode, const double* xs, size_t x_plane, const unsigned* xshape, const double* ys,
                               size_t y_plane, const unsigned* yshape, double* res, size_t r_plane, const unsigned* rshape, double* qbuf,
                               size_t q_plane, unsigned* flags_and_counter);
    // Quotients (mode 0) and the slabs k0 >= 1 of logarithms (mode 1) of rank 3 / 4 with rows of 65 .. 4096 coefficients: the segment
    // wavefront with leading axes (gft_div2d.hip k_seg_wavefront), bit-identical to the host-driven recursion.
    // `flags_and_counter`: (rows of res) * ceil(row length / 64) + 1 zeroed words; `qbuf` (mode 1): a tensor like res.
    static bool seg_wavefront(hipStream_t st, int mode, const double* xs, size_t x_plane, const unsigned* xshape, const double* ys, size_t y_plane,
                              const unsigned* yshape, double* res, size_t r_plane, const unsigned* rshape, int nd, double* qbuf, size_t q_plane,
                              unsigned* flags_and_counter);
    // factor tables computed on device in the reference's operation order (mt:472-478, 499-506, 557-565)
    static void factor_table(hipStream_t st, int op, unsigned n, unsigned len, const double* m, size_t m_plane,
                             double* tab, size_t tab_plane);
    // out[o, i] = sum_k in[o, k, i] (sequential ascending k; unrolled8 = ndarray's 8-way fold for the lane)
    static void sum_axis(hipStream_t st, const double* in, size_t in_plane, unsigned outer, unsigned len,
                         unsigned inner, size_t axis_stride_outer, double* out, size_t out_plane, int mode);
    // *count += number of positions where a != b  (n contiguous elements)
    static void any_zero(hipStream_t st, const double* p, size_t plane, size_t n, unsigned* state, const Mailbox& mb);
    // payload[0] = exact zeros among the n < 2^32 coefficients, payload[1 + u] = those in slab 0 of axis u < 6 of `sh`; state: 8 zeroed words
    static void zero_pattern(hipStream_t st, const double* p, size_t plane, const Shape& sh, size_t n, unsigned* state, const Mailbox& mb);
    static void count_neq(hipStream_t st, const double* a, size_t a_plane, const double* b, size_t b_plane,
                          size_t n, unsigned* count);
    // reference-order truncated N-d Cauchy product, one thread per output element (mt:984-1012)
    static void conv_naive(hipStream_t st, const double* x, size_t x_plane, const double* y, size_t y_plane,
                           double* z, size_t z_plane, const ConvArgs& a);
    // the same loop nest (bit-exact) for products with few terms per output, with the following Add (and the Horner
    // loop's witness) fused in: ConvEpi.  false: outside the kernel's domain, nothing launched.
    static bool conv_shallow(hipStream_t st, const double* x, size_t x_plane, const double* y, size_t y_plane, double* out,
                             size_t out_plane, const ConvArgs& a, const ConvEpi& e);
    // a plain product one operand of which is a LINE (extent 1 on every axis but one, which is not the last): a tile of the
    // other operand in LDS, the line through uniform loads, the reference's order per output (bit-exact).  false: outside
    // the kernel's domain, nothing launched.
    static bool conv_line(hipStream_t st, const double* x, size_t x_plane, const double* y, size_t y_plane, double* z, size_t z_plane,
                          const ConvArgs& a);
};

enum SumMode { SUM_SEQ = 0, SUM_UNROLL8 = 1, SUM_WAVE = 2 };

// LDS-tiled f64 convolution (gft_conv_tiled.hip).  Returns false if the shape is not supported
// by the tiled kernel (caller falls back to conv_naive).  `ws`/`ws_bytes`: workspace for
// split-J partial tiles (may be null to query the needed size via *ws_needed).
// `nf_flag`/`nf_epoch`: the packing/scan kernels raise *nf_flag to nf_epoch if an operand holds inf/NaN, in which
// case the main and reduce kernels leave z untouched (zero padding times inf would create NaNs the reference
// does not produce) and the caller's guarded reference-order launch computes z instead — no host round trip.
// `guarded` (optional): set to false when the launch read its operands in place — no zero padding, hence no verdict and
// no guarded fallback launch needed (ConvArgs::operands_slack; round 4) — true otherwise.
bool conv_tiled_f64(hipStream_t st, const double* x, const double* y, double* z, const ConvArgs& a, void* ws,
                    size_t ws_bytes, size_t* ws_needed, unsigned* nf_flag, unsigned nf_epoch, bool* guarded = nullptr);

// Inner-axis splitting helpers for the tiled kernel (see gft_conv_tiled.hip): zero-pad rows to `plen`, and the
// overlap-add that folds the (Pz, 2B-1) pieces of every row back into a row of zI coefficients.
template <class E>
bool conv_pairs(hipStream_t st, const double* x, size_t xp, const double* y, size_t yp, double* z, size_t zp, const ConvArgs& a, double max_macs);
void staged_set_rb_min_macs(double v);  // threshold of the register-blocked interval product (negative: never)
void staged_set_rb_pairs(double v);    // row-pair form (k_pair_sums + k_pair_collect): 0 never, 1 by size, 2 always, negative: the default
void staged_set_rb_pairs_cap(double bytes);  // bytes of row sums it may hold, all streams together (< 1: the default, 2 GiB)
void shallow_set_pair_min(double v);         // smallest result of k_conv_shallow's two-outputs-per-thread form (negative: never; default 4096)
void staged_set_rb_pairs_lanes(double v);    // slab ranges on two lanes: 0 never, 1 always, negative: when half the cap leaves the ranges as they are
size_t staged_scratch_bytes();  // bytes the grow-only kernel workspaces of gft_conv_staged.hip hold (gft_pool_stats)
void staged_release_scratch();  // frees the register-blocked interval product's row-flag scratch (gft_shutdown)
void dwf_release_orders();  // frees the row wavefront's cached claim-order tables (gft_shutdown)
void tiled_set_lane_tile(int tsh);  // 0 = planner's choice, 3..6 = force T1 = 1 << tsh lanes along k1 (tests, A/B)
void tiled_pad_rows_f64(hipStream_t st, const double* in, double* out, size_t rows, unsigned len, unsigned P, unsigned B);
void tiled_fold_rows_f64(hipStream_t st, const double* zt, double* z, size_t rows, size_t row_lo, size_t row_hi, unsigned Pz,
                         unsigned B, unsigned zI, int accumulate, const unsigned* guard, unsigned epoch);

// LDS-staged reference-order convolution (gft_conv_staged.hip): bit-identical to K<E>::conv_naive, operands
// staged through LDS once per workgroup step.  Returns false (nothing launched) if the shape does not suit
// it; `force` ignores the "worth it" thresholds.
template <class E>
bool conv_staged(hipStream_t st, const double* x, size_t x_plane, const double* y, size_t y_plane, double* z,
                 size_t z_plane, const ConvArgs& a, bool force);

}  // namespace gft
