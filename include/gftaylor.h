/* gftaylor — C ABI of the MI355X-native multivariate-Taylor arithmetic core.
 *
 * This is the drop-in boundary for the hot path of fzaiser/genfer: the `TaylorPoly<T>` type of
 * src/multivariate_taylor.rs, instantiated at T = F64 (prefix gft_) and T = Interval<F64>
 * (prefix gfti_, `--bounds`).  The reference has no FFI today (the boundary is a Rust generic
 * used only by src/generating_function.rs:8); each entry point below replaces one Rust item and
 * cites it as `mt:<lines>` (= src/multivariate_taylor.rs).  INTEGRATION.md shows the Rust
 * `extern "C"` block + newtype shim a maintainer would add.
 *
 * Model
 *   - A polynomial is an opaque handle (`gft_poly*`).  Its coefficient tensor lives in HBM as a
 *     contiguous row-major f64 array of the *compact* stored shape (mt:13-19); `degrees_p1`
 *     (conceptual truncation orders, SIZE_MAX = untruncated) lives on the host.
 *   - Value arithmetic runs in HIP kernels on gfx950.  There is no CPU fallback: if no device /
 *     code object is available every call fails and gft_last_error() says why.
 *   - Size-threshold dispatch: Genfer issues 10^5-10^6 operations on tensors of a
 *     few hundred elements, each far below a kernel launch.  A tensor with at most
 *     "host_max_elems" elements whose operands are all host-resident (built from host data or by
 *     such operations) stays in host memory and is computed there by the same element functions
 *     the kernels use (one source, same bits, -ffp-contract=off); scalars never leave the host.
 *     A host-resident tensor that meets a device operand is mirrored to the device once.
 *     gft_set_option("host_max_elems", 0) keeps every tensor on the device.
 *   - The ABI is NON-consuming: `out = op(a, b)` never frees or mutates its inputs (the Rust
 *     operators consume by value, mt:857,914,1017,1197; a shim maps that to "call, then drop").
 *     Handles are immutable values; gft_clone is O(1) (shared device buffer).
 *   - Kernel launches are issued by a LAUNCH THREAD of the library, in program order (a launch-bound program spends ~3 us
 *     of host time inside every hipLaunchKernel; the calling thread only records the launch).  Every value inspection,
 *     gft_synchronize, gft_event_record and the raw entry points gft_conv_raw* / gft_dist_* return with all launches in
 *     the stream; a caller that shares the stream with the HANDLE API (gft_set_stream) and records its own events or
 *     launches its own kernels there calls gft_synchronize() or gft_event_record() first.
 *   - Single calling thread per process; one HIP stream (gft_set_stream to adopt the caller's).
 *     The host only synchronises when a VALUE is inspected (to_host, coefficient, constant_term,
 *     is_zero/is_one/extract_*, equal) — and data-dependent dispatch inside mul/div/subst_var
 *     (mt:1021-1061), which the reference also performs.
 *   - Stored (compact) shapes and degrees_p1 equal the reference's in every case (integer
 *     bookkeeping is bit-exact); subst_var's Horner loop speculates on the data-dependent dispatch
 *     of mt:1052-1061 on the device, verifies the speculation with one read-back per call and
 *     redoes the loop step by step if it failed.
 *   - Errors: functions returning a handle return NULL, functions returning int return a
 *     negative value; gft_last_error() holds the message.  Reference panics (assert!/unwrap,
 *     `panic = "abort"`, Cargo.toml:29) map to such errors; IEEE inf/NaN propagate as values.
 *   - Scalars cross the boundary as `const double*` pointing at WIDTH doubles: 1 for gft_
 *     (the f64), 2 for gfti_ ({lo, hi}).  Coefficient data is plane-major: WIDTH planes of
 *     numel doubles each (SoA (lo,hi) planes for intervals).
 */
#ifndef GFTAYLOR_H
#define GFTAYLOR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gft_poly gft_poly;

/* ---- runtime ------------------------------------------------------------------------- */
/* Select the HIP device and create the stream + memory pool.  Idempotent. 0 on success. */
int gft_init(int device);
void gft_shutdown(void);
/* Adopt an existing hipStream_t (e.g. torch's current stream); NULL restores the library's own. */
int gft_set_stream(void* hip_stream);
void* gft_get_stream(void);
int gft_synchronize(void);
const char* gft_last_error(void);
/* Device-memory statistics in bytes: {in_use, cached, peak_in_use}.  in_use and peak include the kernels' grow-only
 * workspaces (the row-pair sums of the reference-order product, at most conv_rb_pairs_cap = 2 GiB over all streams; the tiled
 * product's plan workspace), which are not pool blocks. */
void gft_pool_stats(size_t out[3]);
/* Cumulative operation counters since gft_init: {extract_linear device scans (each a host round trip),
 * 1-element value read-backs, coefficient() read-backs, products on the tiled kernel, on the LDS-staged
 * reference-order kernel, on the one-thread-per-output kernel, operations computed on the host tier,
 * host-resident tensors mirrored to the device}.  Diagnostics. */
void gft_op_stats(size_t out[8]);
/* More counters (returns how many exist — 18 — and writes min(cap, that many)): {kernel launches of the library, elementwise
 * operations deferred into a chain instead of launched, chains materialised by a consumer that needed the tensor in
 * memory, add/sub launches that evaluated deferred chains on the fly, launches whose argument block did not fit a
 * slot of the launch ring and were issued in place after a full drain, shallow (stencil) products on the fused
 * reference-order kernel, of which whole general Horner steps res * subst + slab in one launch, two unused words (0),
 * one more unused word (0), recorded observation chains launched with the consumer's Add as their epilogue, linearity scans
 * answered by a proof about exact zeros (intervals), Adds that evaluated recorded sums in their own launch, executions of the
 * deferred launch graph, recordings issued through them, batched launches, items in them, microseconds of the calling thread
 * inside graph executions}.  Diagnostics; bench.py's e2e rows. */
size_t gft_op_stats_ex(size_t* out, size_t cap);
/* hipEvent timing on the library's stream: record into slot 0..63, elapsed in ms (syncs on b). */
int gft_event_record(int slot);
float gft_event_elapsed_ms(int slot_a, int slot_b);
/* Which convolution kernel `mul` may use: 0 = auto, 1 = force the simple one-thread-per-output
 * kernel, 2 = force the LDS-tiled FMA kernel (errors if the shape is unsupported), 3 = force the
 * LDS-staged reference-order kernel wherever its shape limits allow.  Modes 1 and 3 are bit-identical
 * to each other and to the CPU algorithm.  Test/bench knob. */
int gft_set_conv_mode(int mode);
/* Switches and thresholds by name (returns -1 for an unknown name).  README lists them with their environment variables.
 * SWITCHES (1 = on, the default; 0 = the simpler form, for A/B runs, bisecting and the verification matrix):
 *   "batch_dag"     recordings form a deferred launch graph that is issued level by level as batched launches
 *   "lazy_observe"  observation chains are recorded on their result instead of launched (and take the consumer's Add as epilogue)
 *   "lazy_sum"      Adds of two deferred chains are recorded (nested Adds evaluate them in one launch)
 *   "lazy_horner"   proven linear Horner loops are recorded (with "batch_dag")
 *   "nz_proofs"     interval Horner loops skip the linearity scan where the operand's exact zeros are proven to be none, or whole
 *                   leading slabs
 *   "defer"         elementwise operations are deferred into chains instead of launched one by one
 *   "async_launch"  kernels are issued by the library's launch thread instead of the calling thread
 *   "div_wavefront" division / log / exp recurrences as one-launch wavefronts (0: the slab-by-slab blocked form)
 *   "exp_right"     large f64 exponentials add their terms in arrival order (1e-10 contract; 0: the reference's order everywhere)
 *   "recur_overlap" the blocked recurrences overlap their bulk updates on a second stream (test knob: same bits either way)
 * THRESHOLDS: "host_max_elems" / "host_max_macs" (size-threshold dispatch: largest result, in elements, and largest general
 * product, in multiply-adds, computed on the host tier; 0 = everything on the device), "tiled_min_macs" (auto-mode crossover to
 * the tiled product), "horner_loop_max" (largest final tensor, in elements, whose linear Horner steps all run in one launch; 0 =
 * one launch per step), "shallow_max_terms" (plain products whose outputs receive at most this many terms run on the fused
 * reference-order kernel; 0: never; negative: the default, 256), "shallow_pair_min" (smallest result for which a flat stencil
 * computes two outputs per thread; -1: the default 4096, below: never), "conv_rb_min_macs" (smallest interval product on the
 * register-blocked rows kernel; negative: never), "conv_rb_pairs" (the reference-order product as row-pair sums: 0 never, 1 by
 * size, 2 whenever it applies, negative: the default), "conv_rb_pairs_cap" (bytes of row sums that form may hold at a time;
 * default 2 GiB), "conv_rb_pairs_lanes" (its slab ranges on two lanes: 0 never, 1 always, negative: the default rule),
 * "tiled_tile" (3..6 force the tiled product's lane tile 8x8 .. 1x64; 0: the planner's choice), "dist_min_macs" (smallest
 * general product gft_mul shards over the GPUs of gft_dist_init), "dist_event_slot".
 * TEST KNOBS: "debug_fail_next_launch" (1: the next kernel launch requests 1 MB of LDS and fails — on the launch thread; the
 * failure is reported by the next gft_synchronize / value inspection), "trace_lq_report". */
int gft_set_option(const char* name, double value);
/* Tiled-kernel variant for A/B measurements (-1 = library default).  Test/bench knob. */
int gft_set_conv_variant(int variant);

/* ---- raw device-pointer entry points (no handles) ---------------------------------------- */
/* res[k] (+)= sum_j x[j]*y[k-j] for k0 in [slab_lo, slab_hi): the truncated N-d Cauchy product
 * `mul` of mt:984-1012 on caller-owned contiguous row-major device buffers (e.g. torch tensors),
 * restricted to a range of leading-axis output slabs — the unit of multi-GPU sharding and of the
 * div/exp/log recurrences.  accumulate=0 overwrites the slabs, 1 adds to them.  Runs on the
 * current stream; does not synchronise. */
int gft_conv_raw(const double* x, const size_t* xshape, const double* y, const size_t* yshape, double* res,
                 const size_t* rshape, size_t ndim, size_t slab_lo, size_t slab_hi, int accumulate);
/* Exact multiply-accumulate count of those slabs (sum_k prod_v #{valid j_v}). */
double gft_conv_macs(const size_t* xshape, const size_t* yshape, const size_t* rshape, size_t ndim,
                     size_t slab_lo, size_t slab_hi);
/* Leading-axis slab assignment for `world` ranks (work of slab k is proportional to the number
 * of valid j0, i.e. triangular): writes for `rank` up to 2 half-open ranges
 * {lo0,hi0,lo1,hi1} (folded: low group + mirrored high group).  Pure integer logic; needs no GPU.
 * Returns 1 if every rank's two groups have equal sizes (all-gather friendly), 0 otherwise. */
int gft_plan_slabs(size_t n0, int world, int rank, size_t out[4]);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI, collectives internal to the library --------
 * The reference is single-process; a host that wants one large product spread over the GPUs of a node starts one
 * process per GPU (each with its own gft_init(device)), lets rank 0 call gft_dist_unique_id, hands the 128 bytes to
 * the other ranks by whatever means it has (MPI, torch.distributed, a file), and has every rank call gft_dist_init.
 * From then on every rank runs the SAME program on replicated data; gft_mul shards a general f64 product of at least
 * "dist_min_macs" multiply-adds (gft_set_option, default 1e10) over the leading output axis — slab k depends on
 * x[0..=k], y[0..=k] and on no other output (mt:1001-1011), so there is no reduction: folded slab assignment
 * (gft_plan_slabs), local kernels, then in-place ncclAllGather of the low groups and grouped ncclSend/ncclRecv of the
 * mirrored high groups (zero-filled ncclAllReduce when the axis does not divide evenly) on the library's stream.
 * RCCL is dlopen'ed by gft_dist_init; single-GPU users never load it. */
int gft_dist_unique_id(void* out128);                                   /* ncclGetUniqueId (rank 0) */
int gft_dist_init(int rank, int world, const void* unique_id128);       /* ncclCommInitRank on the gft_init device */
int gft_dist_world(void);
int gft_dist_rank(void);
int gft_dist_comm_count(void);                                          /* ncclCommCount of the communicator (0: none) */
int gft_dist_shutdown(void);
/* Every rank (collective): small sharded products through both sharded entries — even split (all-gather + point-to-point)
 * and uneven split (zero-filled all-reduce) — against the rank's own full product, bit for bit.  0 = the exchange
 * delivers every slab; run automatically by gft_dist_init when GFT_DIST_SELFTEST=1 is in the environment. */
int gft_dist_selftest(void);
/* ncclBroadcast of `count` doubles at `buf` (device memory) from `root`: replicating operands that originate on one rank */
int gft_dist_broadcast(double* buf, size_t count, int root);
/* res = x (*) y like gft_conv_raw over ALL leading slabs, sharded over the communicator; x, y replicated on every rank,
 * every rank receives the full result.  With world == 1 (or before gft_dist_init) it is the plain product. */
int gft_conv_raw_sharded(const double* x, const size_t* xshape, const double* y, const size_t* yshape, double* res,
                         const size_t* rshape, size_t ndim);

/* ---- constructors ------------------------------------------------------------------------- */
gft_poly* gft_from_host(const double* coeffs, const size_t* shape, const size_t* degrees_p1,
                        size_t ndim);                                   /* TaylorPoly::new        mt:33-41   */
gft_poly* gft_scalar(const double* x);                                  /* From<T>                mt:626-630 */
gft_poly* gft_from_u32(uint32_t c);                                     /* from_u32               mt:219-225 */
gft_poly* gft_zero_with(const size_t* degrees_p1, size_t ndim);         /* zero_with              mt:208-216 */
gft_poly* gft_var(size_t v, const double* x, size_t len);               /* var                    mt:239-248 */
gft_poly* gft_var_at_zero(size_t v, size_t len);                        /* var_at_zero            mt:228-237 */
gft_poly* gft_var_with_degrees_p1(size_t v, const double* x, const size_t* degrees_p1,
                                  size_t ndim);                         /* var_with_degrees_p1    mt:250-259 */
gft_poly* gft_clone(const gft_poly* p);                                 /* Clone                  mt:10      */
void gft_free(gft_poly* p);                                             /* Drop                              */

/* ---- queries ------------------------------------------------------------------------------ */
int gft_width(void);                                                    /* 1 (gfti_width() = 2)              */
size_t gft_num_vars(const gft_poly* p);                                 /* num_vars               mt:48-51   */
size_t gft_numel(const gft_poly* p);                                    /* coeffs.len()                      */
void gft_shape(const gft_poly* p, size_t* out);                         /* coeffs.shape() (compact)          */
void gft_degrees_p1(const gft_poly* p, size_t* out);                    /* shape()                mt:53-56   */
int gft_to_host(const gft_poly* p, double* out);                        /* array()/into_array     mt:58-66   */
size_t gft_len_of(const gft_poly* p, size_t v);                         /* len_of                 mt:72-79   */
int gft_is_constant(const gft_poly* p);                                 /* is_constant            mt:68-70   */
int gft_is_zero(const gft_poly* p);                                     /* Zero::is_zero          mt:643-645 */
int gft_is_one(const gft_poly* p);                                      /* One::is_one            mt:653-655 */
int gft_equal(const gft_poly* a, const gft_poly* b);                    /* PartialEq              mt:10      */
/* Display (debug == 0: fmt_polynomial, mt:694-730, e.g. "1.0 + 2.0b + 3.0a^2") or Debug (debug != 0: "TaylorPoly([degrees_p1],
 * <polynomial>)", mt:632-636) as NUL-terminated UTF-8 into out[0..cap); returns the full length (call with cap 0 to size
 * the buffer), -1 on error.  Floats print like the reference's F64 (ryu shortest round-trip, f64.rs:41-45). */
long gft_format(const gft_poly* p, int debug, char* out, size_t cap);   /* Display / Debug        mt:632-636,694-730 */
int gft_constant_term(const gft_poly* p, double* out);                  /* constant_term          mt:296-299 */
int gft_extract_constant(const gft_poly* p, double* out);               /* extract_constant       mt:262-269 */
int gft_extract_linear(const gft_poly* p, double* c, double* m, size_t* v); /* extract_linear     mt:275-294 */
int gft_coefficient(const gft_poly* p, const size_t* index, size_t n, double* out); /* coefficient mt:314-339 */

/* ---- algebra ------------------------------------------------------------------------------ */
gft_poly* gft_add(const gft_poly* a, const gft_poly* b);                /* Add                    mt:854-882 */
gft_poly* gft_sub(const gft_poly* a, const gft_poly* b);                /* Sub                    mt:911-937 */
/* a + b * from(c) in one pass (the accumulation `sum += term * TaylorPoly::from(lah)` of the negative-binomial
 * observation, generating_function.rs:743-746): per element (0 + a) + (c * b), same operations and order as the two calls. */
gft_poly* gft_add_scaled(const gft_poly* a, const gft_poly* b, const double* c);
gft_poly* gft_neg(const gft_poly* a);                                   /* Neg                    mt:902-909 */
gft_poly* gft_mul(const gft_poly* a, const gft_poly* b);                /* Mul + mul/mul_1d       mt:971-1072 */
gft_poly* gft_div(const gft_poly* a, const gft_poly* b);                /* Div + div              mt:1162-1231 */
gft_poly* gft_exp(const gft_poly* a);                                   /* exp                    mt:406-417,1270-1317 */
gft_poly* gft_log(const gft_poly* a);                                   /* log                    mt:419-430,1319-1386 */
gft_poly* gft_pow(const gft_poly* a, uint32_t e);                       /* pow                    mt:433-451 */

/* ---- structure ---------------------------------------------------------------------------- */
gft_poly* gft_derivative(const gft_poly* a, size_t v, size_t n);        /* derivative             mt:457-481 */
gft_poly* gft_taylor_expansion_of_coeff(const gft_poly* a, size_t v, size_t n); /*                mt:484-509 */
gft_poly* gft_shift_down(const gft_poly* a, size_t v, size_t n);        /* shift_down (axis sum)  mt:514-536 */
/* Fused form of three reference calls: (derivative(a, v, 1).truncate_to_degree_p1(d) * var(v, x, d))
 * * from(c) — one step of the compound-Poisson observation loop, generating_function.rs:684-689 — same
 * per-element operation order, one kernel launch, no dispatch read-backs. */
gft_poly* gft_observe_step(const gft_poly* a, size_t v, const double* x, const double* c, size_t degree_p1);
/* n such steps in one call, innermost first: a <- gft_observe_step(a, v, x, cs + i*WIDTH, degree_p1 + (n - 1 - i)) for
 * i = 0..n-1 — the whole loop of generating_function.rs:684-689 as the evaluator unfolds it (each level one degree
 * lower than the one inside it).  One launch for the chain: every line along v runs all steps on its own. */
gft_poly* gft_observe_chain(const gft_poly* a, size_t v, const double* x, const double* cs, size_t n, size_t degree_p1);
/* The same for observations from a Poisson with a CONTINUOUS rate (generating_function.rs:703-706):
 * derivative(a, v, 1).truncate_to_degree_p1(d) * from(c), i.e. c * (x * ff) per element, in one launch. */
gft_poly* gft_derive_scale(const gft_poly* a, size_t v, const double* c, size_t degree_p1);
/* Fused form of derivative(a, v, n).truncate_to_degree_p1(degree_p1) — the evaluator's Derivative arm
 * (generating_function.rs:628-633: operand evaluated to degree_p1 + n, differentiated, cut back): one launch,
 * same values (truncation is slicing). */
gft_poly* gft_derivative_truncated(const gft_poly* a, size_t v, size_t n, size_t degree_p1);
gft_poly* gft_subst_var(const gft_poly* a, size_t v, const gft_poly* subst); /* subst_var (Taylor shift / marginalize / Horner) mt:540-580 */
gft_poly* gft_coefficients_of_term(const gft_poly* a, size_t v, size_t order); /*                 mt:341-358 */
gft_poly* gft_taylor_polynomial_terms(const gft_poly* a, size_t v, const size_t* orders,
                                      size_t n);                        /*                        mt:380-404 */
gft_poly* gft_truncate_to_degree_p1(const gft_poly* a, size_t degree_p1); /*                      mt:183-193 */
gft_poly* gft_remove_last_variable(const gft_poly* a);                  /*                        mt:172-181 */
gft_poly* gft_extend_to_dim(const gft_poly* a, size_t ndim, size_t degree_p1); /* extend          mt:81-89   */
gft_poly* gft_extend(const gft_poly* a, const size_t* new_size, size_t n); /* (test-only) extend  mt:91-112  */
gft_poly* gft_mul_var(const gft_poly* a, const double* m, size_t v, const size_t* shape,
                      const size_t* degrees_p1, size_t n);              /* mul_var                mt:589-608 */
gft_poly* gft_mul_linear(const gft_poly* a, const double* c, const double* m, size_t v,
                         const size_t* shape, const size_t* degrees_p1, size_t n); /* mul_linear  mt:611-623 */

/* ---- Interval<F64> twins (src/interval.rs; `--bounds`, main.rs:115-127) -------------------- */
/* Same functions with prefix gfti_; scalars are {lo,hi}, data is two planes (lo then hi).      */
#define GFT_DECLARE_INTERVAL_TWINS 1
const char* gfti_last_error(void);
int gfti_width(void);
gft_poly* gfti_from_host(const double* planes, const size_t* shape, const size_t* degrees_p1, size_t ndim);
gft_poly* gfti_scalar(const double* x);
gft_poly* gfti_from_u32(uint32_t c);
gft_poly* gfti_zero_with(const size_t* degrees_p1, size_t ndim);
gft_poly* gfti_var(size_t v, const double* x, size_t len);
gft_poly* gfti_var_at_zero(size_t v, size_t len);
gft_poly* gfti_var_with_degrees_p1(size_t v, const double* x, const size_t* degrees_p1, size_t ndim);
gft_poly* gfti_clone(const gft_poly* p);
void gfti_free(gft_poly* p);
size_t gfti_num_vars(const gft_poly* p);
size_t gfti_numel(const gft_poly* p);
void gfti_shape(const gft_poly* p, size_t* out);
void gfti_degrees_p1(const gft_poly* p, size_t* out);
int gfti_to_host(const gft_poly* p, double* out);
size_t gfti_len_of(const gft_poly* p, size_t v);
int gfti_is_constant(const gft_poly* p);
int gfti_is_zero(const gft_poly* p);
int gfti_is_one(const gft_poly* p);
int gfti_equal(const gft_poly* a, const gft_poly* b);
long gfti_format(const gft_poly* p, int debug, char* out, size_t cap);
int gfti_constant_term(const gft_poly* p, double* out);
int gfti_extract_constant(const gft_poly* p, double* out);
int gfti_extract_linear(const gft_poly* p, double* c, double* m, size_t* v);
int gfti_coefficient(const gft_poly* p, const size_t* index, size_t n, double* out);
gft_poly* gfti_add(const gft_poly* a, const gft_poly* b);
gft_poly* gfti_sub(const gft_poly* a, const gft_poly* b);
gft_poly* gfti_add_scaled(const gft_poly* a, const gft_poly* b, const double* c);
gft_poly* gfti_neg(const gft_poly* a);
gft_poly* gfti_mul(const gft_poly* a, const gft_poly* b);
gft_poly* gfti_div(const gft_poly* a, const gft_poly* b);
gft_poly* gfti_exp(const gft_poly* a);
gft_poly* gfti_log(const gft_poly* a);
gft_poly* gfti_pow(const gft_poly* a, uint32_t e);
gft_poly* gfti_derivative(const gft_poly* a, size_t v, size_t n);
gft_poly* gfti_taylor_expansion_of_coeff(const gft_poly* a, size_t v, size_t n);
gft_poly* gfti_shift_down(const gft_poly* a, size_t v, size_t n);
gft_poly* gfti_observe_step(const gft_poly* a, size_t v, const double* x, const double* c, size_t degree_p1);
gft_poly* gfti_observe_chain(const gft_poly* a, size_t v, const double* x, const double* cs, size_t n, size_t degree_p1);
gft_poly* gfti_derive_scale(const gft_poly* a, size_t v, const double* c, size_t degree_p1);
gft_poly* gfti_derivative_truncated(const gft_poly* a, size_t v, size_t n, size_t degree_p1);
gft_poly* gfti_subst_var(const gft_poly* a, size_t v, const gft_poly* subst);
gft_poly* gfti_coefficients_of_term(const gft_poly* a, size_t v, size_t order);
gft_poly* gfti_taylor_polynomial_terms(const gft_poly* a, size_t v, const size_t* orders, size_t n);
gft_poly* gfti_truncate_to_degree_p1(const gft_poly* a, size_t degree_p1);
gft_poly* gfti_remove_last_variable(const gft_poly* a);
gft_poly* gfti_extend_to_dim(const gft_poly* a, size_t ndim, size_t degree_p1);
gft_poly* gfti_extend(const gft_poly* a, const size_t* new_size, size_t n);
gft_poly* gfti_mul_var(const gft_poly* a, const double* m, size_t v, const size_t* shape,
                       const size_t* degrees_p1, size_t n);
gft_poly* gfti_mul_linear(const gft_poly* a, const double* c, const double* m, size_t v, const size_t* shape,
                          const size_t* degrees_p1, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* GFTAYLOR_H */
