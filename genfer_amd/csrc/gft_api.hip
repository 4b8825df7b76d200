// Host side of libgftaylor: runtime (device, stream, memory pool), the TaylorPoly bookkeeping
// of src/multivariate_taylor.rs (compact shapes, degrees_p1, broadcast, shortcut dispatch) and
// the C ABI of include/gftaylor.h.  Value arithmetic is done by the kernels in gft_kernels.hip /
// gft_conv_*.hip — and, below the size threshold of SURVEY §8f-2, by the host tier of gft_host.hpp
// (same element functors, same bits; the reference computes everything on the host).  There is no CPU
// fallback for a missing device: without a usable gfx950 every entry point fails with an error message.
//
// Citations `mt:<lines>` refer to /root/reference/src/multivariate_taylor.rs.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen'ed by gft_dist_init (single-GPU users never load it)

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <tuple>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gftaylor.h"
#include "gft_kernels.hpp"
#include "gft_small_alloc.hpp"
#include "gft_host.hpp"
#include "gft_fmt.hpp"

using namespace gft;
static const size_t UMAX = SIZE_MAX;

// Shape / degree lists.  A std::vector here cost ~20 heap round trips per operation — visible when a whole
// operation is a 4 us kernel launch (Genfer programs are 10^5-10^6 tiny operations) — so the list lives inline:
// the reference's programs have <= 8 variables, the ABI admits 32.
template <class T>
struct SmallVecT {
    // (round 6: 8 entries inline, up to 32 on the heap.  With 32 inline a gft_poly was 616 bytes and every recording, handle
    // copy and argument of the host logic moved two of them: memcpy was 13 % of mixture's calling thread.)
    static constexpr size_t INL = 8, CAP = 32;
    size_t n = 0;
    T inl[INL];
    T* heap = nullptr;  // CAP entries once n has exceeded INL
    T* data() { return heap ? heap : inl; }
    const T* data() const { return heap ? heap : inl; }
    SmallVecT() {}
    SmallVecT(size_t count, T val) {
        reserve(count);
        T* v = data();
        for (size_t i = 0; i < count; ++i) v[i] = val;
        n = count;
    }
    SmallVecT(std::initializer_list<T> l) {
        reserve(l.size());
        T* v = data();
        for (T x : l) v[n++] = x;
    }
    template <class It>
    SmallVecT(It a, It b) {
        for (; a != b; ++a) push_back((T)*a);
    }
    SmallVecT(const SmallVecT& o) {
        if (__builtin_expect(o.heap == nullptr, 1)) {  // (the common case is a fixed-size copy: no loop, no branch on n)
            n = o.n;
            std::memcpy(inl, o.inl, sizeof(inl));
        } else
            assign(o);
    }
    SmallVecT(SmallVecT&& o) noexcept {
        n = o.n;
        if (__builtin_expect(o.heap != nullptr, 0)) {
            heap = o.heap;
            o.heap = nullptr;
            o.n = 0;
        } else
            std::memcpy(inl, o.inl, sizeof(inl));
    }
    SmallVecT& operator=(const SmallVecT& o) {
        if (__builtin_expect(o.heap == nullptr && heap == nullptr, 1)) {
            n = o.n;
            std::memcpy(inl, o.inl, sizeof(inl));
        } else if (this != &o)
            assign(o);
        return *this;
    }
    SmallVecT& operator=(SmallVecT&& o) noexcept {
        if (this == &o) return *this;
        if (o.heap) {
            delete[] heap;
            heap = o.heap;
            o.heap = nullptr;
            n = o.n;
            o.n = 0;
        } else
            assign(o);
        return *this;
    }
    ~SmallVecT() {
        if (__builtin_expect(heap != nullptr, 0)) delete[] heap;
    }
    void assign(const SmallVecT& o) {
        reserve(o.n);
        T* v = data();
        const T* w = o.data();
        for (size_t i = 0; i < o.n; ++i) v[i] = w[i];
        n = o.n;
    }
    void reserve(size_t want) {
        if (want > CAP) throw std::runtime_error("more than 32 variables are not supported");
        if (want > INL && !heap) {
            heap = new T[CAP];
            for (size_t i = 0; i < n; ++i) heap[i] = inl[i];
        }
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return data()[i]; }
    const T& operator[](size_t i) const { return data()[i]; }
    T* begin() { return data(); }
    T* end() { return data() + n; }
    const T* begin() const { return data(); }
    const T* end() const { return data() + n; }
    T& back() { return data()[n - 1]; }
    const T& back() const { return data()[n - 1]; }
    void push_back(T x) {
        reserve(n + 1);
        data()[n++] = x;
    }
    void pop_back() { --n; }
    void clear() { n = 0; }
    void resize(size_t m, T val = 0) {
        reserve(m);
        T* v = data();
        for (size_t i = n; i < m; ++i) v[i] = val;
        n = m;
    }
    template <class It>
    void insert(const T* pos, It a, It b) {  // only appending is used
        (void)pos;
        for (; a != b; ++a) push_back((T)*a);
    }
    bool operator==(const SmallVecT& o) const {
        if (n != o.n) return false;
        const T* v = data();
        const T* w = o.data();
        for (size_t i = 0; i < n; ++i)
            if (v[i] != w[i]) return false;
        return true;
    }
    bool operator!=(const SmallVecT& o) const { return !(*this == o); }
};
typedef SmallVecT<size_t> Dims;
typedef SmallVecT<long long> Shifts;  // per-axis source offsets of a gather

// ------------------------------------------------------------------------------------------
// runtime
// ------------------------------------------------------------------------------------------
namespace {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

thread_local std::string g_err;

#define HIP_OK(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            throw Error(std::string("HIP error: ") + hipGetErrorString(e_) + " in " #call);           \
    } while (0)

struct Runtime {
    bool ready = false;
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    size_t stats_side[4] = {0, 0, 0, 0};    // {-, -, recordings that rode along with another launch of their kind, lazy observations fused}
    size_t stats_nz = 0;                    // linearity scans answered by a "no exact zero" proof
    size_t stats_sum = 0;                   // Adds that evaluated a recorded Add of two chains in their own launch (K<E>::chain_nest)
    bool lazy_sum = true;                   // "lazy_sum" / GFT_LAZY_SUM: Adds of two chains are recorded, not launched (Ops::fuse_lazy_sums)
    bool nz_proofs = true;                  // "nz_proofs" / GFT_NZ_PROOFS: interval tensors proven free of exact zeros skip the Horner loops' linearity scans
    bool lazy_horner = true;                // "lazy_horner" / GFT_LAZY_HORNER: proven Horner loops on old operands are recorded (Ops::horner_linear_rest)
    bool batch_dag = true;                  // "batch_dag" / GFT_BATCH: recordings form a launch graph issued level by level as batches (gft_batch.hpp)
    bool lazy_observe = true;               // "lazy_observe" / GFT_LAZY_OBSERVE: observation chains are recorded, not launched (Ops::observe_chain)
    // Side stream of the blocked recurrences (div / log): the bulk of a right-looking update runs here while the main
    // stream already divides the next slab.  Joined before the recurrence returns, so the pool's "one stream" rule holds
    // for every buffer that outlives it.
    hipStream_t side = nullptr;
    hipEvent_t ev_main = nullptr, ev_bulk = nullptr;
    bool side_pending = false;     // a bulk update is (possibly) still running on `side`
    bool recur_overlap = true;     // GFT_RECUR_OVERLAP=0 / "recur_overlap": everything on the main stream (A/B, bisecting)
    // size-class pool: freed blocks are reused immediately — legal because every kernel, memset and
    // copy of this library is ordered on the one stream.
    std::map<size_t, std::vector<void*>> free_blocks;  // size class -> free device blocks (vectors: no node churn)
    size_t in_use = 0, cached = 0, peak = 0;
    size_t peak_total = 0;       // largest (pool blocks in use + the kernels' grow-only workspaces) seen at an allocation or by gft_pool_stats
    unsigned* d_flag = nullptr;  // small device scratch for predicates / counters
    double* d_scratch = nullptr; // small device scratch for packed read-backs
    unsigned* d_wit = nullptr;   // sticky non-linearity witnesses of a speculative Horner loop (Ops::WIT_SLOTS words)
    double* h_pinned = nullptr;  // pinned staging for small D2H reads
    double* h_mail = nullptr;    // mailbox (mapped coherent pinned memory): 8 doubles payload + sequence word
    double* d_mail = nullptr;    // the same slot as the device sees it
    unsigned long long mail_seq = 0;
    hipEvent_t events[64] = {};
    int conv_mode = 0;
    static constexpr int pairs_first = 1;  // small plain f64 products ask the row-pair form before the tiled kernel ("pairs_first", GFT_PAIRS_FIRST)
    double pairs_first_max = 1.0e7, pairs_first_max_rank2 = 2.0e8;  // ... up to this many multiply-adds (rank >= 3 / rank 2)
    double tiled_min_macs = 2.0e5;  // auto mode: products below this stay on the reference-order kernels
    double tiled_min_override = -1;  // >= 0 while a div / log recurrence issues its accumulation products (recur_tiled_min_macs)
    // div / log: accumulation steps of at least this many multiply-adds may take the tiled kernel.  Off by default: the
    // quotient of a division cancels, and the tiled kernel's summation order showed up as 4e-10 relative on single
    // coefficients of a 64^3 quotient (profiles/r02/recurrences.txt) — inside the normwise bound of SURVEY 8d, outside
    // the 1e-10-per-coefficient contract.  gft_set_option("recur_tiled_min_macs", 5e7) trades that for ~20 % at 64^3.
    static constexpr double recur_tiled_min_macs = 1.0e300;  // (round 6: no longer an option — the recurrences keep the reference's order)
    size_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // see gft_op_stats
    size_t stats_ex[4] = {0, 0, 0, 0};  // {operations deferred into a chain, chains materialised, fused chain add/sub launches, -}
    bool defer = true;             // GFT_DEFER=0 / "defer": one launch per elementwise operation (A/B, bisecting)
    size_t horner_loop_max = (size_t)1 << 40;  // elements of the final tensor up to which the whole Horner loop is one launch
    static constexpr bool fuse_horner = true;  // (the fused / speculative Horner loop; the generic loop is horner_exact)
    static constexpr bool div2d = true;  // (the last two axes of the division recurrence in one launch; the host-driven recursion is the fallback for shapes outside the kernels' domains)
    bool div_wavefront = true;     // GFT_DIV_WAVEFRONT=0 / "div_wavefront": the blocked recurrence instead of the one-launch row wavefront
    static constexpr bool rows_wavefront = true;  // (rank-2 recurrences with rows > 64 as a coefficient-level wavefront)
    bool exp_right = true;         // GFT_EXP_RIGHT=0 / "exp_right": left-looking exp steps everywhere (A/B and bisecting)
    // Shallow products (round 4): a plain product whose outputs receive at most this many terms each (prod_i min(xs_i, ys_i):
    // one operand is a stencil — the substitutions of `+~ Binomial(other, p)` statements are 3-6 coefficients) runs on the
    // reference-order one-thread-per-output kernel with the Horner step's Add fused in (K<E>::conv_shallow) instead of the
    // tiled / staged kernels.  0 = off ("shallow_max_terms" / GFT_SHALLOW_MAX_TERMS; A/B and bisecting).
    size_t shallow_max_terms = 256;  // (64 -> 256: hmm's [2,72,1] and mixture's [209,1] factors 67 -> 35 us, 99 -> 48 us; profiles/r04/shallow_max_terms.txt)
    size_t stats_shallow[2] = {0, 0};  // {shallow products, of which fused Horner steps}
    unsigned nf_epoch = 0;          // non-finite verdict stamp of the current tiled product (d_flag[2])
    int conv_variant = -1;
    void* conv_ws = nullptr;
    size_t conv_ws_bytes = 0;
    // size-threshold dispatch: an operation whose operands are all host-resident runs on the host
    // tier (gft_host.hpp) if its result has at most host_max_elems elements (and, for a general product, at most
    // host_max_macs multiply-adds); 0 = everything on the device.  Crossovers measured with tools/xover_host.py.
    // Measured on MI355X + its host (profiles/r02/xover_host.txt): a streaming operation costs the device ~4 us whatever
    // its size and the host ~2.4 us per 1024 elements; a general product costs the device >= 15-20 us (launches + the
    // dispatcher's two extract_linear read-backs) and the host 0.4 / 0.55 / 1.1 ns per multiply-add at rank 1 / 2 / >= 3.
    static constexpr size_t HOST_MAX_ELEMS_DEFAULT = 2048;
    static constexpr double HOST_MAX_MACS_DEFAULT = 65536;  // in rank-1 equivalents (est_macs weighs the rank)
    size_t host_max_elems = HOST_MAX_ELEMS_DEFAULT;
    double host_max_macs = HOST_MAX_MACS_DEFAULT;
    std::map<size_t, std::vector<void*>> host_blocks;  // free host-tier blocks by size class
};
Runtime& R = *new Runtime;  // (never destroyed: handles and cached tables released during static destruction still find their pool)

static size_t size_class(size_t bytes) {
    if (bytes < 256) return 256;
    if (bytes <= (1u << 20)) {  // next power of two up to 1 MiB
        size_t c = 256;
        while (c < bytes) c <<= 1;
        return c;
    }
    const size_t g = 2u << 20;  // 2 MiB granules above
    return (bytes + g - 1) / g * g;
}

static void release_kernel_scratch() { staged_release_scratch(); }  // (the device is idle when this is called)


static void* pool_alloc(size_t bytes, size_t* cls_out) {
    size_t cls = size_class(bytes);
    *cls_out = cls;
    std::vector<void*>& fl = R.free_blocks[cls];
    void* p = nullptr;
    if (!fl.empty()) {
        p = fl.back();
        fl.pop_back();
        R.cached -= cls;
    }
    if (!p) {
        hipError_t e = hipMalloc(&p, cls);
        if (e != hipSuccess) {  // release every cache (the stream idle first) and retry once
            launch_drain();
            (void)(hipDeviceSynchronize)();
            for (auto& kv : R.free_blocks)
                for (void* q : kv.second) (void)hipFree(q);
            R.free_blocks.clear();
            R.cached = 0;
            release_kernel_scratch();  // the row-pair workspace and the register-blocked kernel's scratch (gft_conv_staged.hip)
            HIP_OK(hipMalloc(&p, cls));
        }
    }
    R.in_use += cls;
    if (R.in_use > R.peak) {  // (a new pool high: what do the workspaces hold at this moment?)
        R.peak = R.in_use;
        R.peak_total = std::max(R.peak_total, R.in_use + staged_scratch_bytes() + R.conv_ws_bytes);
    }
    return p;
}

// Everything is ordered on one stream: a freed block is reusable at once.
static void pool_free(void* p, size_t cls) {
    R.in_use -= cls;
    R.cached += cls;
    R.free_blocks[cls].push_back(p);
}

static void* host_alloc(size_t bytes, size_t* cls_out) {
    size_t cls = size_class(bytes);
    *cls_out = cls;
    std::vector<void*>& fl = R.host_blocks[cls];
    if (!fl.empty()) {
        void* p = fl.back();
        fl.pop_back();
        return p;
    }
    void* p = std::malloc(cls);
    if (!p) throw std::runtime_error("out of host memory");
    return p;
}
static void host_free(void* p, size_t cls) { R.host_blocks[cls].push_back(p); }

struct LazyOp;
struct Buf : std::enable_shared_from_this<Buf> {
    double* p = nullptr;
    size_t cls = 0;
    bool borrowed = false;
    bool host = false;           // p is host memory (host tier); `dev` is its device mirror once a kernel needed it
    // the contents have not been launched yet (a recorded observation chain): use_buf() launches, or the consumer fuses
    std::shared_ptr<LazyOp> lazy;
    // interval tensors: 2 = PROVEN to hold no coefficient that is exactly [0,0] (Ops::nz_of), 1 = holds one / descends from a
    // tensor that does (nobody asks again), 0 = unknown
    unsigned char nz = 0;
    // (round 6) 3 = the exact zeros are PROVEN to be exactly the leading slabs: coefficient k is [0,0] iff k_u < zpre[u] for some
    // axis u < ZAX (what `observe k ~ Poisson(l * X)` leaves behind when X is evaluated at 0: slab 0 along X) — see Support
    static constexpr int ZAX = 8;
    unsigned short zpre[ZAX] = {0, 0, 0, 0, 0, 0, 0, 0};
    // a recording's buffer gets its pool block when it is launched (ensure_alloc): `want` doubles; p == nullptr until then
    size_t want = 0;
    const char* origin = nullptr;  // the entry point whose result first owned this buffer (diagnostics: GFT_TRACE_SCANS)
    unsigned dag_mark = 0;       // run_dag: visited in this execution
    int dag_level = 0;           // ... and its level (longest path from tensors in memory)
    std::shared_ptr<Buf> dev;
    // device tensors whose coefficients are read one by one (probs_taylor / moments_taylor read `limit` of them,
    // generating_function.rs:963,992): the second read mirrors the whole (immutable) buffer to the host once
    std::shared_ptr<Buf> host_copy;
    unsigned coef_reads = 0;
    // memoised extract_linear() verdict: buffers are immutable once their polynomial is returned, and the
    // metadata-only reshapes that share a buffer (extend_to_dim, dropping a trailing unit axis) keep the
    // indices of all non-unit axes, so the verdict is a property of the buffer
    int lin_state = 0;  // 0 unknown, 1 not linear, 2 linear
    double lin_c[2] = {0, 0}, lin_m[2] = {0, 0};
    size_t lin_var = 0;
    ~Buf() {
        if (!p || borrowed) return;
        if (host) host_free(p, cls);
        else if (R.ready) pool_free(p, cls);
    }
};
// What a lazy buffer needs to become real: `run(b)` launches the producer into b->p on the current stream; `fuse` (optional)
// launches it with a consumer's Add folded into its epilogue, writing somewhere else (Ops::observe_chain).
struct DagRec;
struct LazyOp {
    std::function<void(Buf*)> run;
    std::shared_ptr<DagRec> rec;   // (round 6) the recording as a node of the deferred launch graph (gft_batch.hpp); null: launched by run() only
    std::shared_ptr<void> obs;     // Ops<E>::LazyObs for the fused form (typed by the element class that recorded it)
    std::shared_ptr<void> horner;  // Ops<E>::LazyHorner: a recorded linear Horner loop (rides along with another loop's launch)
    std::shared_ptr<void> sum;     // Ops<E>::LazySum: a recorded Add / Sub of two chains (an Add that consumes it launches both: K<E>::chain_nest)
};

// (+ 8 doubles of slack: the tiled product reads operands in place and its pipelined x loads request one 64-byte chunk
// beyond the last one they use — gft_conv_tiled.hip, ConvArgs::operands_slack)
static std::shared_ptr<Buf> alloc_doubles(size_t n) {
    auto b = std::allocate_shared<Buf>(gft_small::Alloc<Buf>());
    b->p = (double*)pool_alloc((std::max<size_t>(n, 1) + 8) * sizeof(double), &b->cls);
    return b;
}
// a recording's result: no memory yet (ensure_alloc, when the recording is launched)
static std::shared_ptr<Buf> alloc_recorded(size_t n) {
    auto b = std::allocate_shared<Buf>(gft_small::Alloc<Buf>());
    b->want = std::max<size_t>(n, 1);
    return b;
}
static void ensure_alloc(Buf* b) {
    if (b->p || b->host) return;
    b->p = (double*)pool_alloc((std::max<size_t>(b->want, 1) + 8) * sizeof(double), &b->cls);
}
static std::shared_ptr<Buf> alloc_host_doubles(size_t n) {
    auto b = std::allocate_shared<Buf>(gft_small::Alloc<Buf>());
    b->host = true;
    b->p = (double*)host_alloc(std::max<size_t>(n, 1) * sizeof(double), &b->cls);
    return b;
}
static std::shared_ptr<Buf> alloc_tier(bool host, size_t n) { return host ? alloc_host_doubles(n) : alloc_doubles(n); }

static void force_buf(Buf* b);
static void ensure_alloc(Buf* b);
#include "gft_batch.hpp"
// Every access to a device buffer's contents on behalf of work about to be issued on the stream.
static inline void use_buf(Buf* b) {
    if (!b->host && b->lazy) force_buf(b);
}
static void force_buf(Buf* b) {
    std::shared_ptr<LazyOp> op = b->lazy;
    if (op->rec && R.batch_dag) {  // a node of the deferred launch graph: everything it depends on, level by level
        run_dag(b);
        return;
    }
    ensure_alloc(b);
    b->lazy = nullptr;  // (first: run() reaches dp() of OTHER buffers only)
    try {
        op->run(b);
    } catch (...) {  // nothing was launched into b->p (pool exhaustion, a refused launch): the buffer is still a recording
        b->lazy = op;
        throw;
    }
}
static void require_ready() {
    if (!R.ready) {
        if (gft_init(-1) != 0) throw Error("gftaylor: no usable HIP device (" + g_err + "); there is no CPU fallback");
    }
}

static void read_back(void* dst, const void* dev_src, size_t bytes) {
    HIP_OK(hipMemcpyAsync(R.h_pinned, dev_src, bytes, hipMemcpyDeviceToHost, R.stream));
    HIP_OK(hipStreamSynchronize(R.stream));
    std::memcpy(dst, R.h_pinned, bytes);
}

// Host round trips through the mailbox (gft_kernels.hpp): next_mail() hands the kernel its slot + sequence number,
// wait_mail() polls the sequence word.  The poll is bounded: every ~20 us it asks the stream for errors, and a
// stream that went idle without publishing is an error (a kernel died).
// GFT_TRACE_SCANS=1: histogram of extract_linear device scans by call site and tensor size, printed at exit
struct ScanTrace {
    bool on = getenv("GFT_TRACE_SCANS") != nullptr;
    const char* ctx = "api";
    std::map<std::string, size_t> counts;
    void hit(size_t numel, size_t nd) {
        if (!on) return;
        char key[128];
        snprintf(key, sizeof key, "%s numel<=%zu nd=%zu", ctx, (size_t)1 << (numel <= 1 ? 0 : (64 - __builtin_clzll(numel - 1))), nd);
        counts[key]++;
    }
    ~ScanTrace() {
        if (!on) return;
        for (auto& kv : counts) fprintf(stderr, "[gft scans] %-40s %zu\n", kv.first.c_str(), kv.second);
    }
};
static ScanTrace g_scan_trace;
struct ScanCtx {
    const char* prev;
    explicit ScanCtx(const char* c) : prev(g_scan_trace.ctx) { g_scan_trace.ctx = c; }
    ~ScanCtx() { g_scan_trace.ctx = prev; }
};

// (All mails share ONE payload slot and the poll matches the sequence number exactly: a mail issued while a scan's mail is
// still outstanding — between extract_linear_begin and _end, where the caller queues guarded launches — would overwrite the
// payload and make the scan's wait miss its number.  Nothing does that today; this makes sure nothing starts to.)
static int g_scan_mail_open = 0;
static Mailbox next_mail() {
    if (g_scan_mail_open) throw Error("internal: a mailbox round trip was started while a linearity scan's mail is outstanding");
    Mailbox mb;
    mb.payload = R.d_mail;
    mb.seq = (unsigned long long*)(R.d_mail + 8);
    mb.value = ++R.mail_seq;
    mb.dev_word = R.d_flag + 16;  // the scans also leave their verdict in device memory (guards of speculative launches)
    return mb;
}
static void wait_mail(const Mailbox& mb, double* out, unsigned n) {
    volatile unsigned long long* seq = (volatile unsigned long long*)(R.h_mail + 8);
    for (unsigned long long spins = 0;; ++spins) {
        if (*seq == mb.value) break;
        if ((spins & 0x3fff) == 0x3fff) {
            hipError_t q = hipStreamQuery(R.stream);
            if (q == hipSuccess) {
                if (*seq == mb.value) break;
                throw Error("device read-back kernel finished without publishing its result");
            }
            if (q != hipErrorNotReady) HIP_OK(q);
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (unsigned i = 0; i < n; ++i) out[i] = ((volatile double*)R.h_mail)[i];
}
// n <= 7 doubles at src[i * stride] -> host
static void peek(double* out, const double* dev_src, size_t stride, unsigned n) {
    Mailbox mb = next_mail();
    peek_to_mailbox(R.stream, dev_src, stride, n, mb);
    wait_mail(mb, out, n);
}

// ---- deferred elementwise chains (gft_kernels.hpp ChainSrc) -------------------------------------------------------------
// A device table of factors shared by chain stages: the powers m^k of a scaling substitution (the reference's running
// product, mt:557-565), computed once per m on the host tier's functor and reused by every later substitution by the
// same m (Genfer programs substitute the same few constants thousands of times).
struct TabEntry {
    std::shared_ptr<Buf> dev;   // W planes of `len` doubles
    std::vector<double> host;   // the same values (element 0 of a chain is host-computable)
    size_t len = 0;
    int nz = -1;                // (intervals) 1: no entry is exactly [0,0]; -1: not looked at yet
};
struct PendStage {
    int kind = 0, axis = 0;
    double s[2] = {0, 0};
    std::shared_ptr<TabEntry> tab;
};
struct Pend {
    Dims base_shape;            // shape of the base tensor in `buf` (its axes align with the handle's leading axes)
    size_t base_numel = 1;      // = plane stride of the base
    size_t base_off = 0;        // element offset of the handle's element 0 inside the base (a sub-box view)
    // zeros in front (mul_var, mt:589-608: the operand shifted up along one axis): the handle's index k reads base index
    // k - pad[ax], valid inside src_box[ax]; the recorded stages apply to the data only, the padding stays +0 — which is
    // why no stage may be added AFTER a pad (c * 0 would have to flip the zero's sign): such a chain is materialised first
    bool padded = false;
    Dims pad, src_box;
    int n = 0;
    PendStage st[gft::CHAIN_MAX];
    std::shared_ptr<Buf> mat;   // the materialised tensor once some consumer needed it (shared by all copies of the handle)
    Dims mat_shape;
};

std::map<std::tuple<unsigned long long, unsigned long long>, std::shared_ptr<TabEntry>> g_pow_tabs[2];  // Ops<E>::pow_table

}  // namespace

// ------------------------------------------------------------------------------------------
// polynomial handle
// ------------------------------------------------------------------------------------------
struct gft_poly {
    // handles come from the small-block lists (gft_small_alloc.hpp): 10^5-10^6 of them per program
    static void* operator new(size_t n) { return gft_small::get(n); }
    static void operator delete(void* p, size_t n) noexcept { gft_small::put(p, n); }
    int width = 1;              // 1: F64, 2: Interval (lo plane, hi plane)
    Dims shape;                 // stored (compact) coefficient shape
    Dims deg;                   // degrees_p1
    mutable std::shared_ptr<Buf> buf;   // width * numel doubles, plane stride == numel (null: lazy host-cached scalar)
    size_t numel = 1;
    // host cache of the value when numel == 1 (filled on construction from host scalars or lazily)
    mutable bool cached = false;
    mutable double cv[2] = {0, 0};
    // lazy `x + m*eps_v` built from host scalars (numel == 2, buf == null until a kernel has to read it):
    // element 0 = cv, element 1 = cv1, v = lazy_var.  Products with it take the mul_linear path from these
    // host values, so most such tensors never reach the device.
    bool lazy_lin = false;
    double cv1[2] = {0, 0};
    size_t lazy_var = 0;
    // element 0 of a device tensor (numel > 1) when the host happens to know it: after a constant_term() read-back,
    // and after `p - constant_term(p)` (x - x = +0 exactly, F64) — the interpreter does exactly this pair on every
    // Subst node, and subst_var then needs no device scan to learn that a 2-element substitution has no constant
    mutable bool c0_known = false;
    mutable double c0[2] = {0, 0};
    // deferred elementwise chain: the value is chain(buf restricted to the leading box `shape`); buf holds the BASE
    // tensor (device).  Consumers that understand chains read it directly, everyone else goes through dp(), which
    // materialises it once (settle).
    mutable std::shared_ptr<Pend> pend;
};

namespace {

// Device pointer of a polynomial's coefficients.  1-element polynomials built from host scalars are lazy:
// their value travels as a kernel argument wherever possible (constant scaling, scalar add, division by a
// constant) and a device buffer is only created when some kernel really needs to read it from memory.
static bool same_dims_mod_trailing_ones(const Dims& a, const Dims& b) {
    const size_t n = std::max(a.size(), b.size());
    for (size_t i = 0; i < n; ++i)
        if ((i < a.size() ? a[i] : 1) != (i < b.size() ? b[i] : 1)) return false;
    return true;
}
// The chain of `p` (or the plain tensor, as a chain without stages) as a kernel operand over the output axes `keep`.
// (`touch` = false: geometry only — nothing is launched, c.p stays null: what a recording may look at before its inputs exist)
template <class E>
static gft::ChainSrc chain_src(const gft_poly& p, const Dims& keep, bool touch = true) {
    gft::ChainSrc c;
    std::memset(&c, 0, sizeof(c));
    const Dims& bs = p.pend ? p.pend->base_shape : p.shape;
    Dims st(bs.size(), 1);
    for (size_t i = bs.size(); i-- > 1;) st[i - 1] = st[i] * bs[i];
    if (touch) {
        use_buf(p.buf.get());
        c.p = p.buf->p + (p.pend ? p.pend->base_off : 0);
    }
    c.plane = p.pend ? p.pend->base_numel : p.numel;
    for (size_t j = 0; j < keep.size(); ++j) {
        const size_t ax = keep[j];
        c.box[j] = (unsigned)(ax < p.shape.size() ? p.shape[ax] : 1);
        c.pad[j] = 0;
        if (p.pend && p.pend->padded && ax < p.pend->pad.size()) {
            c.pad[j] = (int)p.pend->pad[ax];
            c.box[j] = (unsigned)p.pend->src_box[ax];
        }
        c.stride[j] = ax < bs.size() ? st[ax] : 0;
    }
    if (p.pend) {
        c.nstages = p.pend->n;
        for (int i = 0; i < p.pend->n; ++i) {
            const PendStage& g = p.pend->st[i];
            gft::ChainStage& o = c.st[i];
            o.kind = g.kind;
            o.s = gft::Scalar2{g.s[0], g.s[1]};
            o.axis = 0;
            if (g.kind == gft::CH_MUL_TAB) {
                bool found = false;
                for (size_t j = 0; j < keep.size(); ++j)
                    if (keep[j] == (size_t)g.axis) {
                        o.axis = (int)j;
                        found = true;
                    }
                if (!found) throw Error("internal: table axis of a deferred chain was collapsed");
                if (touch) use_buf(g.tab->dev.get());
                o.tab = g.tab->dev->p;
                o.tab_plane = g.tab->len;
            }
        }
    }
    return c;
}
// output axes of a chain kernel: the non-unit axes of `shape` plus every table axis of the operands
static Dims chain_keep(const Dims& shape, std::initializer_list<const gft_poly*> ops) {
    Dims keep;
    for (size_t a = 0; a < shape.size(); ++a) {
        bool k = shape[a] != 1;
        for (const gft_poly* p : ops)
            if (p->pend) {
                for (int i = 0; i < p->pend->n; ++i)
                    if (p->pend->st[i].kind == gft::CH_MUL_TAB && (size_t)p->pend->st[i].axis == a) k = true;
                if (p->pend->padded && a < p->pend->pad.size() && p->pend->pad[a] > 0) k = true;  // the pad is applied per kept axis
            }
        if (k) keep.push_back(a);
    }
    return keep;
}
// ---- "no exact zero anywhere" (round 5, Interval<F64> only) -----------------------------------------------------------------
// Interval arithmetic does not cancel: a product of two intervals is [0,0] only if a factor is (iv:164-190: the zero and one
// short-circuits return an operand, everything else is widened outwards, lo < hi), and a sum only if both terms are
// (iv:126-155).  So "no coefficient of this tensor is exactly [0,0]" is INHERITED by the operations Genfer's observation
// loops are made of — elementwise stages with non-zero constants, sub-box views, observation steps, linear Horner loops with
// non-zero c and m, sums of equal shape — and such a tensor with >= 3 coefficients is never of the form c + m*x_v: the
// per-subst_var linearity scan of the accumulator (a launch and a host round trip, 18 000 per mixture --bounds run) has a
// known answer.  2 = proven, 1 = holds a zero (or descends from such a tensor: nobody asks again), 0 = unknown.
// (round 6) ... and where the zeros ARE is inherited just as well when they form whole leading slabs.  hmm's tensors have them:
// an observation `observe k ~ Poisson(l * X)` evaluated at X = 0 (the variable whose probabilities are asked for) multiplies by
// eps_X alone, so slab 0 along X is exactly zero and everything else is not — "holds a zero", hence no proof, hence 4 464
// linearity scans with a host round trip each in hmm --bounds.  A Support says which coefficients of a tensor are exactly [0,0]:
//   kind 2: none;   kind 3: exactly those with k_u < z[u] for some axis u (z != 0);   kind 4: exactly those outside the box
//   z[u] <= k_u < h[u] (a front-padded view: zeros in front, and its data may end before the handle's box does — the
//   operands of mul_linear's Add, c * t and m * shift(t), each cover what the other leaves);   kind 5: all of them;   kind 1: some, in no such pattern — MEASURED
//   on this very buffer (nz_query);   kind 0: unknown (a descendant of a kind-1 tensor is unknown, not 1: hmm's first statements
//   have irregular zeros — `State := 1` —, the steady state has slabs, and nz_query asks again after its back-off).
struct Support {
    int kind = 0;
    unsigned z[Buf::ZAX] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned h[Buf::ZAX] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};  // kind 4 only
    bool exact() const { return kind == 2 || kind == 3; }
    void normalise() {
        if (kind != 3) return;
        bool any = false;
        for (unsigned v : z) any = any || v != 0;
        if (!any) kind = 2;
    }
    // the leading slabs as the buffer stores them (anything beyond 65535 slabs, or an axis beyond ZAX: not representable)
    void store(Buf* b) const {
        b->nz = (unsigned char)(kind == 4 ? 0 : kind);
        for (int u = 0; u < Buf::ZAX; ++u) b->zpre[u] = (unsigned short)(kind == 3 ? std::min<unsigned>(z[u], 0xffffu) : 0);
        if (kind == 3)
            for (unsigned v : z)
                if (v > 0xffffu) b->nz = 0;
    }
};
template <class E>
static Support support_of_poly(const gft_poly& p) {
    Support s;
    if (E::W != 2 || !p.buf || p.buf->host) return s;
    s.kind = p.buf->nz;
    if (s.kind == 3) {
        if (p.shape.size() > (size_t)Buf::ZAX && !p.pend) {  // (axes the buffer's record does not cover: unit ones are fine)
            for (size_t u = Buf::ZAX; u < p.shape.size(); ++u)
                if (p.shape[u] != 1) return Support{0};
        }
        for (int u = 0; u < Buf::ZAX; ++u) s.z[u] = p.buf->zpre[u];
    }
    if (s.kind == 5) {  // all zero: so is every view of it under stages that keep zeros zero
        if (p.pend)
            for (int i = 0; i < p.pend->n; ++i) {
                const int k = p.pend->st[i].kind;
                const double* sv = p.pend->st[i].s;
                if ((k == gft::CH_FIRST_ADD || k == gft::CH_FIRST_SUB || k == gft::CH_FIRST_SUB_NEG_ALL) && !(sv[0] == 0.0 && sv[1] == 0.0)) return Support{0};
                if (k == gft::CH_DIV_S) return Support{0};  // (0 / 0)
            }
        return s;
    }
    if (s.kind == 1) return Support{0};  // (what was measured on the BUFFER says nothing usable about a view of it; nz_of_poly maps it)
    if (!s.exact()) return s;
    auto clip = [&](const Dims& shape) {  // a leading box keeps the pattern; a slab pattern that swallows the box is "all zero"
        for (size_t u = 0; u < (size_t)Buf::ZAX; ++u) {
            const size_t ext = u < shape.size() ? shape[u] : 1;
            if (s.z[u] >= ext && s.z[u] > 0) {
                s.kind = 5;
                return;
            }
        }
    };
    if (!p.pend) {
        clip(p.shape);
        s.normalise();
        return s;
    }
    const Pend& q = *p.pend;
    if (q.base_shape.size() > (size_t)Buf::ZAX) {
        for (size_t u = Buf::ZAX; u < q.base_shape.size(); ++u)
            if (q.base_shape[u] != 1 && s.kind == 3) return Support{0};
    }
    if (q.base_off != 0 && s.kind == 3) {  // a sub-box that does not start at the origin: the pattern seen from its own origin
        size_t off = q.base_off;
        for (size_t u = q.base_shape.size(); u-- > 0;) {
            const size_t o = off % q.base_shape[u];
            off /= q.base_shape[u];
            if (u < (size_t)Buf::ZAX) s.z[u] = s.z[u] > o ? (unsigned)(s.z[u] - o) : 0u;
        }
    }
    for (int i = 0; i < q.n; ++i) {
        const PendStage& g = q.st[i];
        switch (g.kind) {
            case gft::CH_LMUL_S:
            case gft::CH_MUL_S:
            case gft::CH_DIV_S:
                if (g.s[0] == 0.0 && g.s[1] == 0.0) return Support{0};  // x * [0,0]
                break;
            case gft::CH_MUL_TAB: {
                TabEntry& t = *g.tab;
                if (t.nz < 0) {
                    t.nz = 1;
                    for (size_t k = 0; k < t.len; ++k)
                        if (t.host[k] == 0.0 && t.host[t.len + k] == 0.0) t.nz = 0;
                }
                if (!t.nz) return Support{0};
                break;
            }
            case gft::CH_FIRST_ADD:
            case gft::CH_FIRST_SUB:
            case gft::CH_FIRST_SUB_NEG_ALL:
                // element 0 (+|-) s: a sum is [0,0] only if both terms are — an element 0 that lies in a zero slab becomes s
                if (s.kind == 3 && !(g.s[0] == 0.0 && g.s[1] == 0.0)) return Support{0};
                break;
            default: break;  // neg
        }
    }
    if (q.padded) {  // zeros in front (mul_var): the data, shifted — and possibly zeros behind it
        for (size_t u = 0; u < q.pad.size() && u < (size_t)Buf::ZAX; ++u) {
            s.z[u] += (unsigned)q.pad[u];
            s.h[u] = (unsigned)std::min<size_t>(q.pad[u] + q.src_box[u], u < p.shape.size() ? p.shape[u] : 1);
            if (s.z[u] >= s.h[u]) return Support{5};
        }
        for (size_t u = Buf::ZAX; u < q.pad.size(); ++u)
            if (q.pad[u] != 0) return Support{0};
        s.kind = 4;
        return s;
    }
    clip(p.shape);
    s.normalise();
    return s;
}
// what the rounds-5 callers ask: 2 = no exact zero anywhere, 1 = holds one, 0 = unknown
template <class E>
static int nz_of_poly(const gft_poly& p) {
    const int k = support_of_poly<E>(p).kind;
    if (k == 0 && p.buf && !p.buf->host && p.buf->nz == 1) return 1;
    return k == 2 ? 2 : ((k == 3 || k == 4 || k == 5) ? 1 : 0);
}
// A tensor with this support is PROVEN not to be of the form c + m * x_v (mt:275-294) iff a non-zero coefficient sits at an
// index with two non-zero coordinates or a coordinate >= 2 — the largest index of the box is such a one whenever any exists
// (the support {k >= z} contains it unless the pattern swallows the box, which clip() has excluded).
static bool support_proves_nonlinear(const Support& s, const Dims& shape) {
    if (!s.exact()) return false;  // (kind 5, all zero, is the constant 0: linear)
    int nonunit = 0;
    for (size_t u = 0; u < shape.size(); ++u) {
        if (shape[u] >= 3) return true;
        if (shape[u] == 2) ++nonunit;
    }
    return nonunit >= 2;
}
static void trace_settle();  // GFT_TRACE_API: which entry point materialised a chain (below)
static void trace_mirror(size_t numel);  // ... and which one mirrored a host-tier tensor to the device
// Materialise a deferred chain (one launch; every copy of the handle shares the result).
template <class E>
static void settle(const gft_poly& p) {
    if (!p.pend) return;
    Pend& q = *p.pend;
    if (!(q.mat && same_dims_mod_trailing_ones(q.mat_shape, p.shape))) {
        std::shared_ptr<Buf> out = alloc_doubles(p.numel * E::W);
        Dims keep = chain_keep(p.shape, {&p});
        if (keep.size() > (size_t)MAXD) throw Error("tensor rank exceeds GFT MAXD after collapsing");
        Shape sh;
        sh.nd = (int)keep.size();
        for (size_t j = 0; j < keep.size(); ++j) sh.d[j] = (unsigned)p.shape[keep[j]];
        K<E>::chain_copy(R.stream, out->p, p.numel, sh, chain_src<E>(p, keep));
        trace_settle();
        support_of_poly<E>(p).store(out.get());
        q.mat = out;
        q.mat_shape = p.shape;
        R.stats_ex[1]++;
    }
    p.buf = q.mat;
    p.pend = nullptr;
}

template <class E>
static double* dp(const gft_poly& p) {
    if (p.pend) settle<E>(p);
    if (!p.buf) {
        p.buf = alloc_doubles(p.numel * E::W);
        Scalar2 v{p.cv[0], p.cv[1]};
        if (p.lazy_lin) {
            K<E>::set_small(R.stream, p.buf.get()->p, p.numel, 2, v, Scalar2{p.cv1[0], p.cv1[1]});
            Buf* b = p.buf.get();
            b->lin_state = 2;
            b->lin_c[0] = p.cv[0]; b->lin_c[1] = p.cv[1];
            b->lin_m[0] = p.cv1[0]; b->lin_m[1] = p.cv1[1];
            b->lin_var = p.lazy_var;
        } else {
            K<E>::set_small(R.stream, p.buf.get()->p, p.numel, 1, v, v);
        }
    }
    Buf* b = p.buf.get();
    if (b->host) {  // a host-tier tensor meets a device operand: mirror it once (values travel as kernel arguments)
        if (!b->dev) {
            b->dev = alloc_doubles(p.numel * E::W);
            upload_small(R.stream, b->dev->p, b->p, p.numel * E::W);
            R.stats[7]++;
            trace_mirror(p.numel);
        }
        use_buf(b->dev.get());
        return b->dev->p;
    }
    use_buf(b);
    return b->p;
}
// true iff every value of the polynomial is host-resident (host-tier buffer, or a lazy handle without a buffer)
static inline bool on_host(const gft_poly& p) { return !p.buf || p.buf->host; }
// Host pointer of a host-resident polynomial (lazy handles are materialised in host memory).
template <class E>
static double* hp(const gft_poly& p) {
    if (!p.buf) {
        p.buf = alloc_host_doubles(p.numel * E::W);
        Buf* b = p.buf.get();
        b->p[0] = p.cv[0];
        if (E::W == 2) b->p[p.numel] = p.cv[1];
        if (p.lazy_lin) {
            b->p[1] = p.cv1[0];
            if (E::W == 2) b->p[p.numel + 1] = p.cv1[1];
            b->lin_state = 2;
            b->lin_c[0] = p.cv[0]; b->lin_c[1] = p.cv[1];
            b->lin_m[0] = p.cv1[0]; b->lin_m[1] = p.cv1[1];
            b->lin_var = p.lazy_var;
        }
    }
    if (!p.buf->host) throw Error("internal: host pointer of a device tensor requested");
    return p.buf->p;
}
template <class E>
static double* tp(const gft_poly& p, bool host) { return host ? hp<E>(p) : dp<E>(p); }

static size_t prod(const Dims& s) {
    size_t n = 1;
    for (size_t x : s) n *= x;
    return n;
}

static void check_invariants(const Dims& shape, const Dims& deg) {  // mt:23-31
    if (shape.size() != deg.size()) throw Error("invariant violated: coeffs.ndim() != degrees_p1.len()");
    for (size_t v = 0; v < shape.size(); ++v)
        if (!(0 < shape[v] && shape[v] <= deg[v])) throw Error("invariant violated: 0 < shape[v] <= degrees_p1[v]");
    if (shape.size() > 32) throw Error("more than 32 variables are not supported");
}

static Shape to_shape(const Dims& s) {
    // drop nothing: callers collapse when they need to; MAXD bounds the kernel-side rank
    if (s.size() > (size_t)MAXD) throw Error("tensor rank exceeds GFT MAXD after collapsing");
    Shape r;
    r.nd = (int)s.size();
    for (size_t i = 0; i < s.size(); ++i) {
        if (s[i] > 0xffffffffull) throw Error("axis length exceeds 2^32");
        r.d[i] = (unsigned)s[i];
    }
    return r;
}

// Remove unit axes so that ranks up to 32 with few non-trivial axes fit MAXD.  `keep0` keeps axis 0.
static Dims collapse_mask(const std::vector<const Dims*>& shapes, bool keep0) {
    Dims keep;
    size_t nd = shapes[0]->size();
    for (size_t a = 0; a < nd; ++a) {
        bool all1 = true;
        for (auto s : shapes)
            if ((*s)[a] != 1) all1 = false;
        if (!all1 || (keep0 && a == 0)) keep.push_back(a);
    }
    return keep;
}
static Dims pick(const Dims& s, const Dims& keep) {
    Dims r;
    for (size_t a : keep) r.push_back(s[a]);
    return r;
}
static Dims c_strides(const Dims& s) {
    Dims st(s.size(), 1);
    for (size_t i = s.size(); i-- > 1;) st[i - 1] = st[i] * s[i];
    return st;
}

template <class E>
struct Ops {
    typedef gft_poly P;
    typedef Scalar2 V2;
    static constexpr int W = E::W;

    // A contiguous device view used inside the recurrences.
    struct HV {
        double* p;
        size_t plane;
        Dims shape;
        bool host = false;  // p is host memory (host tier)
        bool slack = false; // device memory from the library's pool: 64 bytes after the tensor's last element are readable (alloc_doubles)
        size_t numel() const { return prod(shape); }
        HV index0(size_t k) const {
            HV r;
            Dims sub(shape.begin() + 1, shape.end());
            r.p = p + k * prod(sub);
            r.plane = plane;
            r.shape = sub;
            r.host = host;
            r.slack = slack;
            return r;
        }
    };
    static HV view(const P& p, bool host = false) {
        HV v{tp<E>(p, host), p.numel, p.shape, host};
        v.slack = !host && p.buf && !p.buf->borrowed;  // (tp() has given lazy handles / host-tier tensors their pool buffer)
        return v;
    }

    // ---- size-threshold dispatch ------------------------------------------------------------
    // An operation runs on the host tier iff every operand is host-resident and its result is small.
    static bool tier_host(size_t out_numel, const P& a) {
        return R.host_max_elems && out_numel <= R.host_max_elems && on_host(a);
    }
    static bool tier_host(size_t out_numel, const P& a, const P& b) { return tier_host(out_numel, a) && on_host(b); }
    // 1-element results of the host tier become lazy scalars (value in the handle, no buffer at all)
    static P seal(P r) {
        if (r.buf && r.buf->host) {
            R.stats[6]++;
            if (r.numel == 1) {
                r.cv[0] = r.buf->p[0];
                r.cv[1] = W == 2 ? r.buf->p[1] : 0.0;
                r.cached = true;
                r.lazy_lin = false;
                r.buf = nullptr;
            }
        }
        return r;
    }
    static V2 hv(const double v[2]) { return V2{v[0], W == 2 ? v[1] : 0.0}; }
    // device-to-device copies and memsets keep their place among the queued launches (gft_launch.hpp): they are queued
    // too instead of making the API thread wait for the launch thread
    static void copy_elems(bool host, double* dst, const double* src, size_t n) {
        if (host) std::memcpy(dst, src, sizeof(double) * n);
        else {
            hipStream_t st = R.stream;
            enqueue_task([=] { lq_note((hipMemcpyAsync)(dst, src, sizeof(double) * n, hipMemcpyDeviceToDevice, st), nullptr, "hipMemcpyAsync (device to device)"); });
        }
    }
    static void zero_elems(bool host, double* dst, size_t n) {
        if (host) std::memset(dst, 0, sizeof(double) * n);
        else {
            hipStream_t st = R.stream;
            enqueue_task([=] { lq_note((hipMemsetAsync)(dst, 0, sizeof(double) * n, st), nullptr, "hipMemsetAsync"); });
        }
    }
    static DView dview(const HV& v, const Dims* keep = nullptr) {
        DView d;
        d.p = v.p;
        d.plane = v.plane;
        d.sh = to_shape(keep ? pick(v.shape, *keep) : v.shape);
        return d;
    }

    // ---- allocation ------------------------------------------------------------------------
    static P make(const Dims& shape, const Dims& deg, bool host = false) {
        check_invariants(shape, deg);
        P r;
        r.width = W;
        r.shape = shape;
        r.deg = deg;
        r.numel = prod(shape);
        r.buf = alloc_tier(host, r.numel * W);
        return r;
    }
    // the result of a RECORDED operation: shape, degrees and every host-visible fact now, memory when it is launched
    static P make_recorded(const Dims& shape, const Dims& deg) {
        check_invariants(shape, deg);
        P r;
        r.width = W;
        r.shape = shape;
        r.deg = deg;
        r.numel = prod(shape);
        r.buf = alloc_recorded(r.numel * W);
        return r;
    }
    static P with_meta(const P& src, const Dims& shape, const Dims& deg) {  // metadata-only reshape
        check_invariants(shape, deg);
        P r = src;
        r.shape = shape;
        r.deg = deg;
        return r;
    }
    static P from_host_value(typename E::V v, const Dims& shape, const Dims& deg) {
        double x[2] = {0.0, 0.0};
        E::st(x, 1, 0, v);
        R.stats[6]++;
        return from_host_scalar(x, shape, deg);
    }
    static P from_host_scalar(const double* x, const Dims& shape, const Dims& deg) {
        check_invariants(shape, deg);
        P r;
        r.width = W;
        r.shape = shape;
        r.deg = deg;
        r.numel = 1;
        r.cached = true;   // lazy: no device buffer until a kernel has to read it (see dp())
        r.cv[0] = x[0];
        r.cv[1] = W == 2 ? x[1] : 0.0;
        return r;
    }
    static P copy_of(const P& a) {
        P r = make(a.shape, a.deg);
        HIP_OK(hipMemcpyAsync(dp<E>(r), dp<E>(a), sizeof(double) * a.numel * W, hipMemcpyDeviceToDevice, R.stream));
        return r;
    }

    // ---- deferred elementwise chains (gft_kernels.hpp ChainSrc; struct Pend above) ---------------------------------------
    // An elementwise operation on a device tensor is not launched: the result handle shares the operand's buffer and
    // records the operation; the kernel that eventually consumes it applies the recorded stages, in order, to every
    // element it loads (same functors => the bits of one launch per operation).  Tensors of fewer than 4 elements and
    // everything host-resident keep the direct paths.
    static ChainSrc chain_src_dev(const P& p, const Dims& keep) {
        ChainSrc c = chain_src<E>(p, keep);
        if (!p.pend) c.p = dp<E>(p);  // a host-tier tensor is read through its device mirror
        return c;
    }
    static bool can_defer(const P& src, size_t out_numel) {
        return R.defer && src.buf && !src.buf->host && out_numel >= 4;
    }
    // element 0 after one more stage, from element 0 before it (host side of a chain: same functors, same bits)
    static void stage_apply_first(const PendStage& g, double v[2]) {
        typename E::V x = E::from(hv(v));
        const typename E::V sv = E::from(hv(g.s));
        switch (g.kind) {
            case CH_LMUL_S: x = E::mul(sv, x); break;
            case CH_MUL_S: x = E::mul(x, sv); break;
            case CH_DIV_S: x = E::div(x, sv); break;
            case CH_NEG: x = E::neg(x); break;
            case CH_FIRST_ADD: x = E::add(x, sv); break;
            case CH_FIRST_SUB: x = E::sub(x, sv); break;
            case CH_FIRST_SUB_NEG_ALL: x = E::neg(E::sub(x, sv)); break;
            case CH_MUL_TAB: x = E::mul(x, E::ld(g.tab->host.data(), g.tab->len, 0)); break;
            default: break;
        }
        double o[2] = {0.0, 0.0};
        E::st(o, 1, 0, x);
        v[0] = o[0];
        v[1] = W == 2 ? o[1] : 0.0;
    }
    // `src` restricted to its leading box `out_shape`, with room for `extra` more stages: a handle that shares src's
    // base buffer and copies its chain.  (A chain that is full, or that somebody has already materialised, restarts
    // from the materialised tensor.)
    static P deferred(const P& src, const Dims& out_shape, const Dims& out_deg, int extra) {
        check_invariants(out_shape, out_deg);
        if (src.pend && (src.pend->n + extra > CHAIN_MAX || src.pend->mat || src.pend->padded)) settle<E>(src);
        P r;
        r.width = W;
        r.shape = out_shape;
        r.deg = out_deg;
        r.numel = prod(out_shape);
        r.buf = src.buf;
        r.c0_known = src.c0_known;
        r.c0[0] = src.c0[0];
        r.c0[1] = src.c0[1];
        r.pend = std::allocate_shared<Pend>(gft_small::Alloc<Pend>());
        if (src.pend) {
            r.pend->base_shape = src.pend->base_shape;
            r.pend->base_numel = src.pend->base_numel;
            r.pend->base_off = src.pend->base_off;
            r.pend->n = src.pend->n;
            for (int i = 0; i < src.pend->n; ++i) r.pend->st[i] = src.pend->st[i];
        } else {
            r.pend->base_shape = src.shape;
            r.pend->base_numel = src.numel;
        }
        return r;
    }
    static void push_stage(P& r, int kind, const double* sv, int axis = 0, std::shared_ptr<TabEntry> tab = nullptr) {
        Pend& q = *r.pend;
        if (q.n >= CHAIN_MAX) throw Error("internal: deferred chain overflow");
        if (q.padded) throw Error("internal: stage after a pad");
        PendStage& g = q.st[q.n++];
        g.kind = kind;
        g.axis = axis;
        g.s[0] = sv ? sv[0] : 0.0;
        g.s[1] = (sv && W == 2) ? sv[1] : 0.0;
        g.tab = tab;
        if (r.c0_known) stage_apply_first(g, r.c0);
        R.stats_ex[0]++;
    }
    // powers m^k, k < len, of a scaling substitution: the reference's running product ((1*m)*m)*.. (mt:557-565) on the
    // host tier's functor, kept on the device per value of m (a longer request re-forms the table: the running product
    // makes every table a prefix of the longer one)
    static std::shared_ptr<TabEntry> pow_table(const double m[2], size_t len) {
        // (file-scope registry, one per element width: gft_shutdown releases it with the pool it allocates from)
        std::map<std::tuple<unsigned long long, unsigned long long>, std::shared_ptr<TabEntry>>& cache = g_pow_tabs[W - 1];
        unsigned long long k0, k1 = 0;
        std::memcpy(&k0, &m[0], 8);
        if (W == 2) std::memcpy(&k1, &m[1], 8);
        auto key = std::make_tuple(k0, k1);
        auto it = cache.find(key);
        if (it != cache.end() && it->second->len >= len) return it->second;
        if (cache.size() > 256) cache.clear();
        size_t want = std::max<size_t>(len, 64);
        if (it != cache.end()) want = std::max(want, 2 * it->second->len);
        auto t = std::make_shared<TabEntry>();
        t->len = want;
        t->host.resize(want * W);
        typename E::V f = E::one();
        const typename E::V mv = E::from(Scalar2{m[0], W == 2 ? m[1] : 0.0});
        for (size_t k = 0; k < want; ++k) {
            E::st(t->host.data(), want, k, f);
            f = E::mul(f, mv);
        }
        t->dev = alloc_doubles(want * W);
        upload_small(R.stream, t->dev->p, t->host.data(), want * W);
        cache[key] = t;
        return t;
    }

    // ---- value inspection (the only host syncs) ---------------------------------------------
    static void first_value(const P& p, double out[2]) {
        if ((p.numel == 1 && p.cached) || (!p.buf && p.lazy_lin)) {  // host-known element 0
            out[0] = p.cv[0];
            out[1] = p.cv[1];
            return;
        }
        if (p.c0_known) {
            out[0] = p.c0[0];
            out[1] = p.c0[1];
            return;
        }
        if (p.pend) {  // element 0 of the base, then the chain's stages on the host (same functors)
            double v[2] = {0, 0};
            R.stats[1]++;
            use_buf(p.buf.get());
            peek(v, p.buf->p + p.pend->base_off, p.pend->base_numel, W);
            for (int i = 0; i < p.pend->n; ++i) stage_apply_first(p.pend->st[i], v);
            out[0] = v[0];
            out[1] = v[1];
            p.c0_known = true;
            p.c0[0] = v[0];
            p.c0[1] = v[1];
            return;
        }
        if (p.buf && p.buf->host) {
            out[0] = p.buf->p[0];
            out[1] = W == 2 ? p.buf->p[p.numel] : 0.0;
            return;
        }
        double tmp[2] = {0, 0};
        R.stats[1]++;
        peek(tmp, dp<E>(p), p.numel, W);
        out[0] = tmp[0];
        out[1] = tmp[1];
        if (p.numel == 1) {
            p.cached = true;
            p.cv[0] = tmp[0];
            p.cv[1] = tmp[1];
        } else {
            p.c0_known = true;
            p.c0[0] = tmp[0];
            p.c0[1] = tmp[1];
        }
    }
    static bool val_is_zero(const double v[2]) { return W == 1 ? v[0] == 0.0 : (v[0] == 0.0 && v[1] == 0.0); }
    static bool val_is_one(const double v[2]) { return W == 1 ? v[0] == 1.0 : (v[0] == 1.0 && v[1] == 1.0); }
    static bool is_zero(const P& p) {  // mt:643-645
        if (p.numel != 1) return false;
        double v[2];
        first_value(p, v);
        return val_is_zero(v);
    }
    static bool is_one(const P& p) {  // mt:653-655
        if (p.numel != 1) return false;
        double v[2];
        first_value(p, v);
        return val_is_one(v);
    }

    // ---- shape helpers (mt:114-204, 832-852) ---------------------------------------------------
    static Dims min_degrees(const P& a, const P& b) {
        Dims d(std::max(a.deg.size(), b.deg.size()), UMAX);
        for (size_t v = 0; v < d.size(); ++v) {
            if (v < a.deg.size()) d[v] = std::min(d[v], a.deg[v]);
            if (v < b.deg.size()) d[v] = std::min(d[v], b.deg[v]);
        }
        return d;
    }
    static Dims max_shape(const P& a, const P& b) {
        Dims s(std::max(a.shape.size(), b.shape.size()), 1);
        for (size_t v = 0; v < s.size(); ++v) {
            if (v < a.shape.size()) s[v] = std::max(s[v], a.shape[v]);
            if (v < b.shape.size()) s[v] = std::max(s[v], b.shape[v]);
            if (v < a.deg.size()) s[v] = std::min(s[v], a.deg[v]);
            if (v < b.deg.size()) s[v] = std::min(s[v], b.deg[v]);
        }
        return s;
    }
    static Dims sum_shape(const P& a, const P& b) {
        Dims s(std::max(a.shape.size(), b.shape.size()), 0);
        for (size_t v = 0; v < s.size(); ++v) {
            if (v < a.shape.size()) s[v] += a.shape[v] - 1;
            if (v < b.shape.size()) s[v] += b.shape[v] - 1;
            s[v] += 1;
            if (v < a.deg.size()) s[v] = std::min(s[v], a.deg[v]);
            if (v < b.deg.size()) s[v] = std::min(s[v], b.deg[v]);
        }
        return s;
    }
    static void broadcast(P& x, P& y) {
        if (x.deg.size() < y.deg.size()) x.deg.insert(x.deg.end(), y.deg.begin() + x.deg.size(), y.deg.end());
        else if (y.deg.size() < x.deg.size()) y.deg.insert(y.deg.end(), x.deg.begin() + y.deg.size(), x.deg.end());
        while (x.shape.size() < y.shape.size()) x.shape.push_back(1);
        while (y.shape.size() < x.shape.size()) y.shape.push_back(1);
    }

    // ---- structured copies -----------------------------------------------------------------------
    // General gather of `src` into a fresh tensor of shape `out_shape`; per-axis shift and valid length.
    // `tier`: -1 = decide here; callers that pass `tab` / `keep` pointers decide first (gather_tier) and pass the
    // pointers of that side.
    static bool gather_tier(const P& src, const Dims& out_shape) { return tier_host(prod(out_shape), src); }
    static P gather(const P& src, const Dims& out_shape, const Dims& out_deg, const Shifts& shift,
                    const Dims& src_len, int op = OP_COPY, const double* s = nullptr, int tab_axis = -1,
                    const double* tab = nullptr, size_t tab_plane = 0, const unsigned char* keep = nullptr, int tier = -1) {
        const bool host = tier < 0 ? gather_tier(src, out_shape) : tier != 0;
        if (!host && !tab && !keep && (op == OP_COPY || op == OP_MUL_S || op == OP_DIV_S || op == OP_NEG || op == OP_LMUL_S) &&
            can_defer(src, prod(out_shape))) {
            // a box of src that lies wholly inside it (no zero padding), optionally mapped elementwise: deferred (no
            // launch).  A box that does not start at element 0 moves the view's origin, which is only meaningful for stages
            // that do not look at positions: a chain holding FIRST / table stages is materialised first.
            bool inside = out_shape.size() <= src.shape.size(), shifted = false;
            for (size_t ax = 0; ax < out_shape.size() && inside; ++ax) {
                if (shift[ax] < 0 || (size_t)shift[ax] + out_shape[ax] > std::min(src_len[ax], src.shape[ax])) inside = false;
                if (shift[ax] != 0) shifted = true;
            }
            if (inside && shifted && src.pend)
                for (int i = 0; i < src.pend->n; ++i)
                    if (src.pend->st[i].kind >= CH_FIRST_ADD) inside = false;  // FIRST_* and MUL_TAB are positional
            // zeros in front (mul_var: shift -1 along one axis, the source cut at src_len): the same view with a pad
            bool front = !inside && out_shape.size() <= src.shape.size();
            if (front) {
                bool any_neg = false;
                for (size_t ax = 0; ax < out_shape.size() && front; ++ax) {
                    if (shift[ax] > 0) front = false;
                    if (shift[ax] < 0) any_neg = true;
                }
                if (!any_neg) front = false;
                // (positional stages of the source chain — FIRST_*, MUL_TAB — are evaluated in the view's own coordinates
                // by k_chain, so they ride along under the pad; a pad on a pad is materialised by deferred())
                if (front && src.pend && src.pend->padded) front = false;
            }
            if (front) {
                P r = deferred(src, out_shape, out_deg, op == OP_COPY ? 0 : 1);
                if (op != OP_COPY) push_stage(r, op == OP_MUL_S ? CH_MUL_S : (op == OP_DIV_S ? CH_DIV_S : (op == OP_NEG ? CH_NEG : CH_LMUL_S)), s);
                Pend& q = *r.pend;
                q.padded = true;
                q.pad = Dims(out_shape.size(), 0);
                q.src_box = Dims(out_shape.size(), 0);
                for (size_t ax = 0; ax < out_shape.size(); ++ax) {
                    q.pad[ax] = (size_t)(-shift[ax]);
                    q.src_box[ax] = std::min(src_len[ax], src.shape[ax]);
                }
                r.c0_known = true;  // element 0 lies in the padding
                r.c0[0] = r.c0[1] = 0.0;
                return r;
            }
            if (inside) {
                P r = deferred(src, out_shape, out_deg, op == OP_COPY ? 0 : 1);
                if (shifted) {
                    const Dims& bs = r.pend->base_shape;
                    size_t stride = 1, off = 0;
                    for (size_t ax = bs.size(); ax-- > 0;) {
                        if (ax < out_shape.size()) off += (size_t)shift[ax] * stride;
                        stride *= bs[ax];
                    }
                    r.pend->base_off += off;
                    r.c0_known = false;  // element 0 is another element now
                }
                if (op != OP_COPY) push_stage(r, op == OP_MUL_S ? CH_MUL_S : (op == OP_DIV_S ? CH_DIV_S : (op == OP_NEG ? CH_NEG : CH_LMUL_S)), s);
                return r;
            }
        }
        P out = make(out_shape, out_deg, host);
        if (out.numel == 0) return out;
        if (W == 2 && !host && !tab && !keep && (op == OP_COPY || op == OP_MUL_S || op == OP_DIV_S || op == OP_NEG || op == OP_LMUL_S) &&
            !((op == OP_MUL_S || op == OP_LMUL_S) && s && val_is_zero(s))) {
            // (round 6, Support) a box of the source that lies inside it keeps the source's zero pattern, seen from its own origin
            Support sp = support_of_poly<E>(src);
            bool inside = sp.exact() && out_shape.size() <= src.shape.size() && out_shape.size() <= (size_t)Buf::ZAX;
            for (size_t ax = 0; ax < out_shape.size() && inside; ++ax) {
                if (shift[ax] < 0 || (size_t)shift[ax] + out_shape[ax] > std::min(src_len[ax], src.shape[ax])) inside = false;
                else {
                    sp.z[ax] = sp.z[ax] > (size_t)shift[ax] ? (unsigned)(sp.z[ax] - (size_t)shift[ax]) : 0u;
                    if (sp.z[ax] > 0 && sp.z[ax] >= out_shape[ax]) sp.kind = 5;
                }
            }
            for (size_t ax = out_shape.size(); ax < src.shape.size() && inside; ++ax)
                if (src.shape[ax] != 1) inside = false;
            if (inside) {
                if (sp.kind != 5) {
                    sp.kind = 3;
                    sp.normalise();
                }
                sp.store(out.buf.get());
            }
        }
        Dims sst = c_strides(src.shape);
        // collapse axes that are trivial in the output and read index 0 (+shift) of the source
        GatherArgs a;
        std::memset(&a, 0, sizeof(a));
        int nd = 0;
        size_t base = 0;
        a.tab_axis = -1;
        for (size_t ax = 0; ax < out_shape.size(); ++ax) {
            if (out_shape[ax] == 1 && (int)ax != tab_axis) {
                long long si = shift[ax];
                if (si < 0 || (size_t)si >= src_len[ax]) {  // whole output is outside the source box
                    zero_elems(host, tp<E>(out, host), out.numel * W);
                    return seal(out);
                }
                base += (size_t)si * sst[ax];
                continue;
            }
            if (nd >= MAXD) throw Error("tensor rank exceeds GFT MAXD after collapsing");
            a.out.d[nd] = (unsigned)out_shape[ax];
            a.shift[nd] = (int)shift[ax];
            a.src_len[nd] = (unsigned)std::min<size_t>(src_len[ax], 0xffffffffu);
            a.src_stride[nd] = sst[ax];
            if ((int)ax == tab_axis) a.tab_axis = nd;
            nd++;
        }
        // merge adjacent axes that are unmasked and contiguous in the source (elementwise maps become 1-D,
        // slab ops 3-D): less index arithmetic per element, longer unit-stride runs for 16-byte accesses
        for (int i = nd - 1; i >= 1;) {
            int o = i - 1;
            bool ok = a.shift[o] == 0 && a.shift[i] == 0 && a.out.d[o] <= a.src_len[o] && a.out.d[i] <= a.src_len[i] &&
                      a.src_stride[o] == (size_t)a.out.d[i] * a.src_stride[i] && o != a.tab_axis && i != a.tab_axis &&
                      (unsigned long long)a.out.d[o] * a.out.d[i] <= 0xffffffffull;
            if (ok) {
                a.out.d[o] = a.out.d[o] * a.out.d[i];
                a.src_len[o] = a.out.d[o];
                a.src_stride[o] = a.src_stride[i];
                for (int j = i; j + 1 < nd; ++j) {
                    a.out.d[j] = a.out.d[j + 1];
                    a.shift[j] = a.shift[j + 1];
                    a.src_len[j] = a.src_len[j + 1];
                    a.src_stride[j] = a.src_stride[j + 1];
                }
                if (a.tab_axis > i) a.tab_axis--;
                nd--;
            }
            i--;
        }
        a.out.nd = nd;
        a.op = op;
        if (s) {
            a.s.a = s[0];
            a.s.b = W == 2 ? s[1] : 0.0;
        }
        a.tab = tab;
        a.tab_plane = tab_plane;
        a.keep = keep;
        if (host) {
            HK<E>::gather(hp<E>(src) + base, src.numel, hp<E>(out), out.numel, a);
            return seal(out);
        }
        K<E>::gather(R.stream, dp<E>(src) + base, src.numel, dp<E>(out), out.numel, a);
        return out;
    }
    static P lead_block(const P& p, const Dims& lens, const Dims& deg) {  // slice 0..lens per axis
        if (lens == p.shape) return with_meta(p, p.shape, deg);
        Shifts shift(lens.size(), 0);
        return gather(p, lens, deg, shift, p.shape);
    }
    static P slab_range(const P& p, size_t v, size_t lo, size_t hi, const Dims& deg, int op = OP_COPY,
                        int tab_axis = -1, const double* tab = nullptr, size_t tab_plane = 0, int tier = -1) {
        Dims out = p.shape;
        out[v] = hi - lo;
        Shifts shift(out.size(), 0);
        shift[v] = (long long)lo;
        return gather(p, out, deg, shift, p.shape, op, nullptr, tab_axis, tab, tab_plane, nullptr, tier);
    }
    static P truncate_degrees(const P& p, const Dims& degs) {  // mt:195-204
        Dims nd = p.deg, lens = p.shape;
        for (size_t v = 0; v < p.deg.size(); ++v) {
            nd[v] = std::min(nd[v], degs[v]);
            if (v < lens.size() && lens[v] > degs[v]) lens[v] = degs[v];
        }
        return lead_block(p, lens, nd);
    }
    static P map_copy(const P& p, int op, const double* s) {  // fresh tensor = f(p) elementwise
        if (p.numel == 1 && p.cached && (op == OP_LMUL_S || op == OP_MUL_S || op == OP_DIV_S || op == OP_NEG)) {
            // both operands are host-known scalars: one IEEE operation on the host (what the reference does,
            // mt:1033-1047), no launch, no buffer — the result is again a lazy scalar
            int kind = op == OP_LMUL_S ? IMM_LMUL : (op == OP_MUL_S ? IMM_MUL : (op == OP_DIV_S ? IMM_DIV : IMM_NEG));
            Scalar2 b{s ? s[0] : 0.0, (s && W == 2) ? s[1] : 0.0};
            return from_host_value(HK<E>::scalar_imm(kind, E::from(Scalar2{p.cv[0], p.cv[1]}), E::from(b)), p.shape, p.deg);
        }
        Shifts shift(p.shape.size(), 0);
        return gather(p, p.shape, p.deg, shift, p.shape, op, s);
    }

    // ---- constructors (mt:208-259) ------------------------------------------------------------------
    static P zero_with(const Dims& deg) {
        double z[2] = {0, 0};
        return from_host_scalar(z, Dims(deg.size(), 1), deg);
    }
    static P scalar(const double* x) { return from_host_scalar(x, {}, {}); }
    static P var_like(size_t v, const double* x, bool have_x, size_t len_v_shape, bool second_is_one, const Dims& deg) {
        Dims shape(deg.size(), 1);
        shape[v] = len_v_shape;
        check_invariants(shape, deg);
        P r;  // lazy (see gft_poly / dp()): nothing is launched here
        r.width = W;
        r.shape = shape;
        r.deg = deg;
        r.numel = prod(shape);  // 1 or 2
        Scalar2 v0{0.0, 0.0}, v1{0.0, 0.0};
        if (have_x) v0 = Scalar2{x[0], W == 2 ? x[1] : 0.0};
        if (r.numel == 2 && second_is_one) v1 = Scalar2{1.0, 1.0};
        r.cv[0] = v0.a;
        r.cv[1] = W == 2 ? v0.b : 0.0;
        if (r.numel == 2) {  // x + 1*eps_v (or 0*eps_v): its extract_linear verdict is known from the host values
            r.lazy_lin = true;
            r.cv1[0] = v1.a;
            r.cv1[1] = W == 2 ? v1.b : 0.0;
            r.lazy_var = v;
        } else {
            r.cached = true;
        }
        return r;
    }

    // the two-element tensor c = [e0, e1] (planes 2 doubles apart) along axis v as a lazy handle (gft_from_host)
    static P affine_like(size_t v, const double* c, const Dims& shape, const Dims& deg) {
        check_invariants(shape, deg);
        P r;
        r.width = W;
        r.shape = shape;
        r.deg = deg;
        r.numel = 2;
        r.lazy_lin = true;
        r.lazy_var = v;
        r.cv[0] = c[0];
        r.cv1[0] = c[1];
        r.cv[1] = W == 2 ? c[2] : 0.0;
        r.cv1[1] = W == 2 ? c[3] : 0.0;
        return r;
    }
    // lazy (0 + m*eps_v) (+|-) cached scalar d -> lazy (d or -d) + m*eps_v; false if the constant is not an exact zero
    static bool lazy_zero_plus(const P& lazy, const P& scalar, bool subtract, P* out) {
        const double d0 = scalar.cv[0], d1 = scalar.cv[1];
        if (W == 1) {
            if (lazy.cv[0] != 0.0) return false;
            double r;
            if (d0 != 0.0 || d0 != d0) r = subtract ? -d0 : d0;  // 0 +/- d (NaN stays NaN)
            else {
                // signed zeros: (+0)+(+0)=+0, (+0)+(-0)=+0, (-0)+(-0)=-0, (-0)+(+0)=+0; x - y = x + (-y)
                bool ns = std::signbit(lazy.cv[0]), nd = std::signbit(d0) != subtract;
                r = (ns && nd) ? -0.0 : 0.0;
            }
            P res = lazy;
            res.cv[0] = r;
            *out = res;
            return true;
        }
        if (!(lazy.cv[0] == 0.0 && lazy.cv[1] == 0.0)) return false;
        P res = lazy;
        if (subtract) {  // add(a, neg(b)) with a == 0 -> neg(b) = (-hi, -lo)
            res.cv[0] = -d1;
            res.cv[1] = -d0;
        } else {
            res.cv[0] = d0;
            res.cv[1] = d1;
        }
        *out = res;
        return true;
    }

    // A recorded Add / Sub of two chains (addsub's chain path): the operands as they stood after broadcast and truncation.
    struct LazySum {
        P a, b;
        bool subtract = false;
        Dims shape;  // the sum's own shape
    };
    static void launch_sum(const LazySum& ls, double* outp) {
        Dims ckeep = chain_keep(ls.shape, {&ls.a, &ls.b});
        if (!ls.a.pend) (void)dp<E>(ls.a);
        if (!ls.b.pend) (void)dp<E>(ls.b);
        Shape sh;
        sh.nd = (int)ckeep.size();
        for (size_t j = 0; j < ckeep.size(); ++j) sh.d[j] = (unsigned)ls.shape[ckeep[j]];
        K<E>::chain_addsub(R.stream, outp, prod(ls.shape), sh, chain_src_dev(ls.a, ckeep), chain_src_dev(ls.b, ckeep), ls.subtract ? 1 : 0);
        R.stats_ex[2]++;
    }
    // addsub(self, other) where an operand is (a stage-carrying whole view of) a recorded sum: ONE launch evaluates the recorded
    // sum(s) and this one, element for element the operations of the separate launches (K<E>::chain_nest).  false = not this
    // case (nothing launched).
    static bool fuse_lazy_sums(const P& self, const P& other, bool subtract, const Dims& shape, const Dims& rd, P* result) {
        if (!R.lazy_sum) return false;
        auto sum_of = [&](const P& p) -> LazySum* {
            if (!p.buf || p.buf->host || !p.buf->lazy || !p.buf->lazy->sum) return nullptr;
            LazySum* ls = static_cast<LazySum*>(p.buf->lazy->sum.get());
            if (!same_dims_mod_trailing_ones(p.shape, ls->shape)) return nullptr;  // (a sub-box view: launch the sum)
            if (p.pend) {
                const Pend& q = *p.pend;
                if (q.padded || q.base_off != 0 || q.mat || !same_dims_mod_trailing_ones(q.base_shape, ls->shape)) return nullptr;
                for (int i = 0; i < q.n; ++i)
                    if (q.st[i].kind == CH_MUL_TAB) return nullptr;  // (table stages index the view's axes: keep it simple)
            }
            return ls;
        };
        LazySum* la = sum_of(self);
        LazySum* lb = sum_of(other);
        if (!la && !lb) return false;
        if (la && lb && self.buf.get() == other.buf.get()) return false;
        // the kernel's axes: the output's non-unit axes and every table / pad axis of the four leaves
        Dims keep;
        {
            std::vector<const P*> leaves;
            auto add_leaves = [&](const P& p, LazySum* l) {
                if (l) {
                    leaves.push_back(&l->a);
                    leaves.push_back(&l->b);
                } else
                    leaves.push_back(&p);
            };
            add_leaves(self, la);
            add_leaves(other, lb);
            for (size_t ax = 0; ax < shape.size(); ++ax) {
                bool k = shape[ax] != 1;
                for (const P* p : leaves)
                    if (p->pend) {
                        for (int i = 0; i < p->pend->n; ++i)
                            if (p->pend->st[i].kind == CH_MUL_TAB && (size_t)p->pend->st[i].axis == ax) k = true;
                        if (p->pend->padded && ax < p->pend->pad.size() && p->pend->pad[ax] > 0) k = true;
                    }
                if (k) keep.push_back(ax);
            }
            for (const P* p : leaves) {  // no leaf may index an axis beyond the output's rank
                if (p->pend)
                    for (int i = 0; i < p->pend->n; ++i)
                        if (p->pend->st[i].kind == CH_MUL_TAB && (size_t)p->pend->st[i].axis >= shape.size()) return false;
            }
        }
        if (keep.size() > (size_t)MAXD || prod(shape) >= 0x7fffffffull) return false;
        // hold the recordings: bringing a leaf into memory may launch other recordings, never these (they are not in any rider list)
        std::shared_ptr<LazyOp> keep_a = la ? self.buf->lazy : nullptr, keep_b = lb ? other.buf->lazy : nullptr;
        if (R.batch_dag) {  // a node of the launch graph (the leaves' geometry is known now, their memory when the level is issued)
            auto fits_meta = [&](const P& p) {
                const ChainSrc c = chain_src<E>(p, keep, false);
                unsigned long long span = 1;
                for (size_t j = 0; j < keep.size(); ++j) span += (unsigned long long)(c.box[j] ? c.box[j] - 1 : 0) * c.stride[j];
                return span < 0x7fffffffull;
            };
            auto all_fit = [&](const P& p, LazySum* l) { return l ? (fits_meta(l->a) && fits_meta(l->b)) : fits_meta(p); };
            if (!all_fit(self, la) || !all_fit(other, lb)) return false;
            auto rec = std::allocate_shared<NestRec>(gft_small::Alloc<NestRec>());
            rec->self = self;
            rec->other = other;
            rec->keep_a = keep_a;
            rec->keep_b = keep_b;
            rec->shape = shape;
            rec->keep = keep;
            rec->subtract = subtract;
            P out = make_recorded(shape, rd);
            sum_nz_store(self, other, shape, out.buf.get());
            out.buf->lazy = op_of(rec);
            *result = out;
            return true;
        }
        auto leaf = [&](const P& p) {
            if (!p.pend) (void)dp<E>(p);
            return chain_src_dev(p, keep);
        };
        auto fits = [&](const ChainSrc& c) {
            unsigned long long span = 1;
            for (size_t j = 0; j < keep.size(); ++j) span += (unsigned long long)(c.box[j] ? c.box[j] - 1 : 0) * c.stride[j];
            return span < 0x7fffffffull;
        };
        auto nest = [&](const P& p, LazySum* l, NestSrc& n) -> bool {
            std::memset(&n, 0, sizeof(n));
            if (!l) {
                n.nested = 0;
                n.a = leaf(p);
                return fits(n.a);
            }
            n.nested = 1;
            n.sub_inner = l->subtract ? 1 : 0;
            n.a = leaf(l->a);
            n.b = leaf(l->b);
            for (size_t j = 0; j < keep.size(); ++j) n.box[j] = (unsigned)(keep[j] < l->shape.size() ? l->shape[keep[j]] : 1);
            if (p.pend) {
                n.npost = p.pend->n;
                for (int i = 0; i < p.pend->n; ++i) {
                    n.post[i].kind = p.pend->st[i].kind;
                    n.post[i].s = Scalar2{p.pend->st[i].s[0], p.pend->st[i].s[1]};
                }
            }
            return fits(n.a) && fits(n.b);
        };
        NestSrc na, nb;
        if (!nest(self, la, na) || !nest(other, lb, nb)) return false;  // (leaves brought into memory stay there: no harm)
        // (a leaf's dp() may have launched one of the recordings after all — as a leaf of itself it cannot, but be safe)
        if ((la && !self.buf->lazy) || (lb && !other.buf->lazy)) return false;
        P out = make(shape, rd);
        sum_nz_store(self, other, shape, out.buf.get());
        Shape sh;
        sh.nd = (int)keep.size();
        for (size_t j = 0; j < keep.size(); ++j) sh.d[j] = (unsigned)shape[keep[j]];
        K<E>::chain_nest(R.stream, dp<E>(out), out.numel, sh, na, nb, subtract ? 1 : 0);
        R.stats_ex[2]++;
        R.stats_sum++;
        *result = out;
        return true;
    }
    // (intervals) a sum of two tensors of the result's own shape: [0,0] only where both are
    // (round 6: with Supports — the union of two "leading zero slabs" patterns over the whole box is the smaller of the two when
    // one contains the other; a front-padded operand (kind 4: zero AT LEAST there) can only be the contained one)
    static Support sum_support(const P& a, const P& b, const Dims& shape) {
        Support r;
        if (W != 2) return r;
        const Support x = support_of_poly<E>(a), y = support_of_poly<E>(b);
        // each operand's non-zero coefficients as a box lo <= k < hi of the sum's index space (interval sums do not cancel: the
        // sum is zero exactly where both operands are); the union of two boxes is a box when they agree on all axes but one and
        // touch or overlap on that one, or when one contains the other
        struct Box {
            bool known = false, empty = false;
            unsigned lo[Buf::ZAX], hi[Buf::ZAX];
        };
        auto box_of = [&](const Support& sp, const P& p) {
            Box bx;
            if (sp.kind == 5) {
                bx.known = bx.empty = true;
                return bx;
            }
            if (!(sp.kind == 2 || sp.kind == 3 || sp.kind == 4)) return bx;
            for (size_t u = Buf::ZAX; u < p.shape.size(); ++u)
                if (p.shape[u] != 1) return bx;
            bx.known = true;
            for (int u = 0; u < Buf::ZAX; ++u) {
                const unsigned ext = (size_t)u < p.shape.size() ? (unsigned)std::min<size_t>(p.shape[u], 0xfffffffeu) : 1u;
                bx.lo[u] = sp.z[u];
                bx.hi[u] = sp.kind == 4 ? std::min(sp.h[u], ext) : ext;
                if (bx.lo[u] >= bx.hi[u]) bx.empty = true;
            }
            return bx;
        };
        const Box A = box_of(x, a), B = box_of(y, b);
        auto full = [&](const Box& bx) {  // no coefficient of the sum's box is zero in this operand: none is in the sum
            if (!bx.known || bx.empty) return false;
            for (int u = 0; u < Buf::ZAX; ++u) {
                const unsigned ext = (size_t)u < shape.size() ? (unsigned)std::min<size_t>(shape[u], 0xfffffffeu) : 1u;
                if (bx.lo[u] != 0 || bx.hi[u] != ext) return false;
            }
            for (size_t u = Buf::ZAX; u < shape.size(); ++u)
                if (shape[u] != 1) return false;
            return true;
        };
        if (full(A) || full(B)) return Support{2};
        if (!A.known || !B.known) {
            if (g_scan_trace.on) {
                char key[160];
                snprintf(key, sizeof key, "sum_support -> unknown: x.kind=%d (nz %d pend %d) y.kind=%d (nz %d pend %d)", x.kind, a.buf ? (int)a.buf->nz : -1, (int)(a.pend != nullptr), y.kind,
                         b.buf ? (int)b.buf->nz : -1, (int)(b.pend != nullptr));
                g_scan_trace.counts[key]++;
            }
            return r;
        }
        Box U;
        U.known = true;
        auto contains = [](const Box& p, const Box& q) {
            for (int u = 0; u < Buf::ZAX; ++u)
                if (q.lo[u] < p.lo[u] || q.hi[u] > p.hi[u]) return false;
            return true;
        };
        if (A.empty && B.empty) return Support{5};
        if (A.empty || (!B.empty && contains(B, A))) U = B;
        else if (B.empty || contains(A, B)) U = A;
        else {
            int diff = -1;
            for (int u = 0; u < Buf::ZAX; ++u)
                if (A.lo[u] != B.lo[u] || A.hi[u] != B.hi[u]) {
                    if (diff >= 0) return r;  // (they differ on two axes: an L-shaped union)
                    diff = u;
                }
            if (std::max(A.lo[diff], B.lo[diff]) > std::min(A.hi[diff], B.hi[diff])) return r;  // (a gap between them)
            U = A;
            U.lo[diff] = std::min(A.lo[diff], B.lo[diff]);
            U.hi[diff] = std::max(A.hi[diff], B.hi[diff]);
        }
        // storable when the box reaches the end of the sum's own box on every axis: "leading zero slabs"
        for (int u = 0; u < Buf::ZAX; ++u) {
            const unsigned ext = (size_t)u < shape.size() ? (unsigned)std::min<size_t>(shape[u], 0xfffffffeu) : 1u;
            if (U.hi[u] != ext) return r;
            r.z[u] = U.lo[u];
        }
        for (size_t u = Buf::ZAX; u < shape.size(); ++u)
            if (shape[u] != 1) return Support{0};
        r.kind = 3;
        r.normalise();
        return r;
    }
    static void sum_nz_store(const P& a, const P& b, const Dims& shape, Buf* out) { sum_support(a, b, shape).store(out); }
    // ---- Add / Sub / Neg (mt:854-937) -----------------------------------------------------------------
    static P addsub(P self, P other, bool subtract) {
        Dims rd = min_degrees(self, other);
        broadcast(self, other);
        self = truncate_degrees(self, rd);
        other = truncate_degrees(other, rd);
        // a host-cached scalar operand travels as a kernel argument (no device read, no materialisation)
        auto sptr = [](const P& s) -> const double* { return (s.cached && !s.buf) ? nullptr : dp<E>(s); };
        if (other.numel == 1) {
            // (c + m*eps_v) - c for a still-lazy variable and the host-cached scalar c it was built from (the
            // reference's Subst evaluation: subst - constant_term(subst)): x - x = +0 exactly for finite x, so the
            // result stays lazy as 0 + m*eps_v.  F64 only (an interval difference widens).
            if (W == 1 && subtract && !self.buf && self.lazy_lin && other.cached && !other.buf && std::isfinite(self.cv[0]) &&
                std::memcmp(&self.cv[0], &other.cv[0], sizeof(double)) == 0 && self.deg == rd) {
                P r = self;
                r.cv[0] = 0.0;
                return r;
            }
            // (0 + m*eps_v) +/- d for a still-lazy variable with an exactly zero constant and a host-cached scalar d:
            // 0 + d = d, 0 - d = -d (IEEE identities; interval.rs:126-155 returns the other operand for an exact
            // zero), so `Var(x) + Const(d)` substitutions never touch the device before they are consumed.
            if (!self.buf && self.lazy_lin && other.cached && !other.buf && self.deg == rd && lazy_zero_plus(self, other, subtract, &self))
                return self;
            if (self.numel == 1 && self.cached && other.cached)  // two host-known scalars (mt:862-869): host arithmetic
                return from_host_value(HK<E>::scalar_imm(subtract ? IMM_SUB : IMM_ADD, E::from(Scalar2{self.cv[0], self.cv[1]}),
                                                         E::from(Scalar2{other.cv[0], other.cv[1]})), self.shape, rd);
            if (other.cached && tier_host(self.numel, self)) {
                P out = make(self.shape, rd, true);
                HK<E>::copy_first(hp<E>(self), self.numel, hp<E>(out), out.numel, self.numel, subtract ? FIRST_SUB : FIRST_ADD,
                                  Scalar2{other.cv[0], other.cv[1]});
                return seal(out);
            }
            if (other.cached && can_defer(self, self.numel)) {  // element 0 (+|-) a host-known scalar: one more stage
                P r = deferred(self, self.shape, rd, 1);
                push_stage(r, subtract ? CH_FIRST_SUB : CH_FIRST_ADD, other.cv);
                return r;
            }
            P out = make(self.shape, rd);
            K<E>::copy_first(R.stream, dp<E>(self), self.numel, dp<E>(out), out.numel, self.numel,
                             subtract ? FIRST_SUB : FIRST_ADD, sptr(other), other.numel, Scalar2{other.cv[0], other.cv[1]});
            if (W == 1 && subtract && self.c0_known && other.cached && std::isfinite(self.c0[0]) &&
                std::memcmp(&self.c0[0], &other.cv[0], sizeof(double)) == 0) {
                out.c0_known = true;  // x - x = +0 exactly: the device computes it, the host merely knows the outcome
                out.c0[0] = out.c0[1] = 0.0;
            }
            return out;
        }
        if (self.numel == 1) {
            if (!subtract && !other.buf && other.lazy_lin && self.cached && !self.buf && other.deg == rd &&
                lazy_zero_plus(other, self, false, &other))
                return other;
            if (self.cached && tier_host(other.numel, other)) {
                P out = make(other.shape, rd, true);
                HK<E>::copy_first(hp<E>(other), other.numel, hp<E>(out), out.numel, other.numel,
                                  subtract ? FIRST_SUB_NEG_ALL : FIRST_ADD, Scalar2{self.cv[0], self.cv[1]});
                return seal(out);
            }
            if (self.cached && can_defer(other, other.numel)) {
                P r = deferred(other, other.shape, rd, 1);
                push_stage(r, subtract ? CH_FIRST_SUB_NEG_ALL : CH_FIRST_ADD, self.cv);
                return r;
            }
            P out = make(other.shape, rd);
            K<E>::copy_first(R.stream, dp<E>(other), other.numel, dp<E>(out), out.numel, other.numel,
                             subtract ? FIRST_SUB_NEG_ALL : FIRST_ADD, sptr(self), self.numel, Scalar2{self.cv[0], self.cv[1]});
            return out;
        }
        Dims shape = max_shape(self, other);
        const bool host = tier_host(prod(shape), self, other);
        auto recorded = [](const P& p) { return p.buf && !p.buf->host && p.buf->lazy; };
        if (!host && (self.pend || other.pend || recorded(self) || recorded(other))) {
            // deferred operands: their chains are evaluated inside the add itself (one launch for the whole run of
            // operations that led here)
            Dims ckeep = chain_keep(shape, {&self, &other});
            if (ckeep.size() <= (size_t)MAXD) {
                P fused;
                if (fuse_lazy_observe(self, other, subtract, shape, rd, &fused)) return fused;
                if (fuse_lazy_sums(self, other, subtract, shape, rd, &fused)) return fused;
                const bool shifted = (self.pend && self.pend->padded) || (other.pend && other.pend->padded);
                const bool record = R.lazy_sum && ((prod(shape) >= 64 && shifted) || (R.batch_dag && prod(shape) >= 4));
                P out = record ? make_recorded(shape, rd) : make(shape, rd);
                sum_nz_store(self, other, shape, out.buf.get());
                if (record) {
                    // The Add inside mul_linear (c * t + m * shift(t): one operand carries a front pad) is RECORDED, not launched:
                    // if an Add consumes it (the merge of an `if` whose arms both end in `State ~ Bernoulli(p)`), both run as one
                    // launch (fuse_lazy_sums); anybody else launches it through use_buf().  Other two-chain Adds are launched
                    // where they stand: their consumers are observation chains, and a later launch is only less overlap
                    // (mixture f64 0.129 -> 0.145 s with every Add recorded)
                    auto ls = std::allocate_shared<LazySum>(gft_small::Alloc<LazySum>());
                    ls->a = self;
                    ls->b = other;
                    ls->subtract = subtract;
                    ls->shape = shape;
                    auto op = std::allocate_shared<LazyOp>(gft_small::Alloc<LazyOp>());
                    op->sum = ls;
                    op->run = [ls](Buf* b) { launch_sum(*ls, b->p); };
                    auto rec = std::allocate_shared<SumRec>(gft_small::Alloc<SumRec>());
                    rec->ls = ls;
                    op->rec = rec;
                    out.buf->lazy = op;
                    return out;
                }
                if (!self.pend) (void)dp<E>(self);   // plain operands: lazy handles / host-tier tensors get their device buffer
                if (!other.pend) (void)dp<E>(other);
                Shape sh;
                sh.nd = (int)ckeep.size();
                for (size_t j = 0; j < ckeep.size(); ++j) sh.d[j] = (unsigned)shape[ckeep[j]];
                K<E>::chain_addsub(R.stream, dp<E>(out), out.numel, sh, chain_src_dev(self, ckeep), chain_src_dev(other, ckeep), subtract ? 1 : 0);
                R.stats_ex[2]++;
                return out;
            }
        }
        P out = make(shape, rd, host);
        Dims keep = collapse_mask({&shape}, false);
        HV vo = view(out, host), va = view(self, host), vb = view(other, host);
        if (host) {
            HK<E>::addsub_padded(dview(vo, &keep), dview(va, &keep), dview(vb, &keep), subtract ? 1 : 0);
            return seal(out);
        }
        K<E>::addsub_padded(R.stream, dview(vo, &keep), dview(va, &keep), dview(vb, &keep), subtract ? 1 : 0);
        return out;
    }
    static P neg(const P& a) { return map_copy(a, OP_NEG, nullptr); }

    // a + b * from(c) — the accumulation step of the reference's negative-binomial observation (gf.rs:743-746:
    // `sum += term * TaylorPoly::from(lah)`) — in one pass, no intermediate tensor: per element (0 + a) + (c * b), the
    // operations of Mul's constant path (mt:1041-1047) and of Add (mt:873-880) in their order.  Everything that is not
    // the general case (scalar operands, c = 0 / 1 / non-finite) takes the two reference calls.
    static P add_scaled(const P& a, const P& b, const double* c) {
        auto unfused = [&]() { return addsub(a, mul(b, scalar(c)), false); };
        if (a.numel == 1 || b.numel == 1 || val_is_zero(c) || val_is_one(c)) return unfused();
        for (int i = 0; i < W; ++i)
            if (!(c[i] - c[i] == 0.0)) return unfused();
        P self = a, other = b;  // other = b * c has b's shape and degrees
        Dims rd = min_degrees(self, other);
        broadcast(self, other);
        self = truncate_degrees(self, rd);
        other = truncate_degrees(other, rd);
        if (self.numel == 1 || other.numel == 1) return unfused();
        Dims shape = max_shape(self, other);
        const bool host = tier_host(prod(shape), self, other);
        P out = make(shape, rd, host);
        Dims keep = collapse_mask({&shape}, false);
        HV vo = view(out, host), va = view(self, host), vb = view(other, host);
        const Scalar2 cs{c[0], W == 2 ? c[1] : 0.0};
        if (host) {
            HK<E>::add_scaled_padded(dview(vo, &keep), dview(va, &keep), dview(vb, &keep), cs);
            return seal(out);
        }
        K<E>::add_scaled_padded(R.stream, dview(vo, &keep), dview(va, &keep), dview(vb, &keep), cs);
        return out;
    }

    // ---- extract_linear (mt:275-294) --------------------------------------------------------------------
    // extract_linear in two halves (round 4): `begin` answers from the host where it can (done = true) and otherwise launches
    // the device scan WITHOUT waiting for it; `end` waits for the mailbox and memoises the verdict on the buffer.  Between
    // the two the caller may queue work that is guarded on the device by the scan's verdict word (Mailbox::dev_word): the
    // speculative Horner loop runs behind the scan instead of behind a host round trip (horner_speculative).
    struct ScanToken {
        bool done = false, result = false;   // answered without a device scan
        double c[2] = {0, 0}, m[2] = {0, 0};
        size_t var = 0;
        Mailbox mb;
        Dims keep;                           // collapsed axes the kernel's mask bits refer to
        std::shared_ptr<Buf> buf;            // the buffer the verdict belongs to
    };
    static ScanToken extract_linear_begin(const P& p) {
        ScanToken t;
        unsigned mask = 0;
        for (size_t v = 0; v < p.shape.size(); ++v)
            if (p.shape[v] >= 2) mask |= 1u << v;
        if (!mask) {
            t.done = true;
            return t;
        }
        if (!p.buf && p.lazy_lin) {
            t.done = t.result = true;
            t.c[0] = p.cv[0]; t.c[1] = p.cv[1];
            t.m[0] = p.cv1[0]; t.m[1] = p.cv1[1];
            t.var = p.lazy_var;
            return t;
        }
        if (p.pend && !p.pend->mat) {
            // a deferred chain whose first consumer asks this question: materialise it and scan it in ONE launch
            // (k_chain_scan); the verdict is memoised on the new buffer like any other
            Pend& q = *p.pend;
            Dims ckeep = chain_keep(p.shape, {&p});
            if (ckeep.size() <= (size_t)MAXD && ckeep.size() <= 32) {
                std::shared_ptr<Buf> outb = alloc_doubles(p.numel * E::W);
                Shape sh;
                sh.nd = (int)ckeep.size();
                unsigned cm = 0;
                for (size_t j = 0; j < ckeep.size(); ++j) {
                    sh.d[j] = (unsigned)p.shape[ckeep[j]];
                    if (p.shape[ckeep[j]] >= 2) cm |= 1u << j;
                }
                t.mb = next_mail();
                g_scan_mail_open = 1;
                K<E>::chain_copy_scan(R.stream, outb->p, p.numel, sh, chain_src<E>(p, ckeep), cm, R.d_flag + 8, t.mb);
                trace_settle();
                R.stats_ex[1]++;
                R.stats[0]++;
                g_scan_trace.hit(p.numel, ckeep.size());
                q.mat = outb;
                q.mat_shape = p.shape;
                p.buf = q.mat;
                p.pend = nullptr;
                t.keep = ckeep;
                t.buf = p.buf;
                return t;
            }
        }
        settle<E>(p);  // the verdict is memoised per buffer: a deferred chain is materialised first
        if (p.buf && p.buf->lin_state) {
            t.done = true;
            t.result = p.buf->lin_state == 2;
            if (t.result) {
                t.c[0] = p.buf->lin_c[0]; t.c[1] = p.buf->lin_c[1];
                t.m[0] = p.buf->lin_m[0]; t.m[1] = p.buf->lin_m[1];
                t.var = p.buf->lin_var;
            }
            return t;
        }
        // kernel works on the collapsed view; map collapsed axis bits back to real axes
        Dims keep = collapse_mask({&p.shape}, false);
        unsigned cmask = 0;
        for (size_t i = 0; i < keep.size(); ++i)
            if (p.shape[keep[i]] >= 2) cmask |= 1u << i;
        t.keep = keep;
        t.buf = p.buf;
        if (p.buf->host) {
            double res[5];
            HV v = view(p, true);
            res[0] = (double)HK<E>::linear_scan(dview(v, &keep), cmask, res + 1, res + 3);  // {mask, c.lo, c.hi, m.lo, m.hi}
            t.done = true;
            finish_scan(t, res);
            return t;
        }
        HV v = view(p);
        DView dv = dview(v, &keep);
        t.mb = next_mail();
        g_scan_mail_open = 1;
        K<E>::linear_scan(R.stream, dv, cmask, R.d_flag + 8, t.mb);  // one launch (state words 8, 9), result by mailbox
        R.stats[0]++;
        g_scan_trace.hit(p.numel, keep.size());
        return t;
    }
    static void finish_scan(ScanToken& t, const double res[5]) {
        const unsigned got = (unsigned)res[0];
        if (!got) {
            t.buf->lin_state = 1;
            t.result = false;
            return;
        }
        size_t ci = 0;
        while (!((got >> ci) & 1u)) ci++;
        t.c[0] = res[1];
        t.c[1] = res[2];
        t.m[0] = res[3];
        t.m[1] = res[4];
        t.var = t.keep[ci];
        t.buf->lin_state = 2;
        t.buf->lin_c[0] = t.c[0]; t.buf->lin_c[1] = t.c[1];
        t.buf->lin_m[0] = t.m[0]; t.buf->lin_m[1] = t.m[1];
        t.buf->lin_var = t.var;
        t.result = true;
    }
    static bool extract_linear_end(ScanToken& t, double c[2], double m[2], size_t* var) {
        if (!t.done) {
            double res[5];
            g_scan_mail_open = 0;
            wait_mail(t.mb, res, 5);
            finish_scan(t, res);
            t.done = true;
        }
        if (t.result) {
            c[0] = t.c[0]; c[1] = t.c[1];
            m[0] = t.m[0]; m[1] = t.m[1];
            *var = t.var;
        }
        return t.result;
    }
    static bool extract_linear(const P& p, double c[2], double m[2], size_t* var) {
        ScanToken t = extract_linear_begin(p);
        return extract_linear_end(t, c, m, var);
    }

    // Inner-axis split for the tiled kernel (gft_conv_tiled.hip, k_pad_rows / k_fold_rows): rank 2/3 whose last axis
    // exceeds the kernel's 128.  Fills the rank-4 problem (pieces, rows.., piece length) into `t` — the piece axis
    // LEADS, which makes it the kernel's wave-uniform axis — and the piece length.
    static bool plan_inner_split(const ConvArgs& a, ConvArgs& t, unsigned* B_out) {
        const int nd = a.nd;
        if (nd != 2 && nd != 3) return false;
        if (a.j0_min || a.j0_excl || a.j0_desc) return false;
        const unsigned zI = a.zs[nd - 1];
        if (zI <= 128) return false;
        // piece length: multiple of 8, <= 64 (2B - 1 <= 127).  P pieces cost P (P + 1) / 2 untruncated B x B piece
        // products for zI^2 / 2 useful ones, and the kernel's efficiency grows with the number of 8-wide output blocks.
        unsigned best = 0;
        double best_score = -1.0;
        for (unsigned B = 64; B >= 32; B -= 8) {
            const double P = (double)((zI + B - 1) / B);
            const double score = (double)zI * zI / (P * (P + 1) * B * B) * (0.45 + 0.55 * B / 64.0);
            if (score > best_score) {
                best_score = score;
                best = B;
            }
        }
        const unsigned B = best;
        t = a;
        t.nd = 4;
        t.xs[0] = (a.xs[nd - 1] + B - 1) / B;
        t.ys[0] = (a.ys[nd - 1] + B - 1) / B;
        t.zs[0] = std::min((zI + B - 1) / B, t.xs[0] + t.ys[0] - 1);  // pieces that receive piece products (the fold knows)
        if (nd == 2) {
            t.xs[1] = t.ys[1] = t.zs[1] = 1;
            t.xs[2] = a.xs[0]; t.ys[2] = a.ys[0]; t.zs[2] = a.zs[0];
            t.slab_axis = 2;
        } else {
            t.xs[1] = a.xs[0]; t.ys[1] = a.ys[0]; t.zs[1] = a.zs[0];
            t.xs[2] = a.xs[1]; t.ys[2] = a.ys[1]; t.zs[2] = a.zs[1];
            t.slab_axis = 1;
        }
        t.xs[3] = B;
        t.ys[3] = B;
        t.zs[3] = 2 * B - 1;
        t.accumulate = 0;  // the fold applies it
        *B_out = B;
        return true;
    }

    // ---- the product (mt:971-1072) --------------------------------------------------------------------------
    // conv: z[slab range] (+)= x (*) y in the reference's summation structure.  `slab_mode` marks a
    // recurrence step (axis 0 is always an "outer" axis, see gft_kernels.hip).
    static void conv(const HV& x, const HV& y, const HV& z, size_t slab_lo, size_t slab_hi, bool accumulate,
                     bool slab_mode, int j0_min, int j0_excl, int j0_desc) {
        Dims keep = collapse_mask({&z.shape}, slab_mode);
        Dims xs = pick(x.shape, keep), ys = pick(y.shape, keep), zs = pick(z.shape, keep);
        if (zs.size() > (size_t)MAXD) throw Error("tensor rank exceeds GFT MAXD after collapsing");
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.nd = (int)zs.size();
        Dims xst = c_strides(xs), yst = c_strides(ys), zst = c_strides(zs);
        for (int i = 0; i < a.nd; ++i) {
            a.xs[i] = (unsigned)xs[i];
            a.ys[i] = (unsigned)ys[i];
            a.zs[i] = (unsigned)zs[i];
            a.xstr[i] = xst[i];
            a.ystr[i] = yst[i];
            a.zstr[i] = zst[i];
        }
        if (!z.shape.empty() && slab_hi <= slab_lo) return;  // empty slab range
        const bool axis0_kept = !keep.empty() && keep[0] == 0;
        if (a.nd == 0) {
            a.slab_lo = 0;
            a.slab_hi = 1;
        } else if (!axis0_kept) {  // the (unit) leading axis was collapsed away: its only slab is everything
            a.slab_lo = 0;
            a.slab_hi = a.zs[0];
        } else {
            a.slab_lo = (unsigned)slab_lo;
            a.slab_hi = (unsigned)slab_hi;
        }
        a.accumulate = accumulate ? 1 : 0;
        a.j0_min = j0_min;
        a.j0_excl = j0_excl;
        a.j0_desc = j0_desc;
        a.variant = R.conv_variant;
        a.operands_slack = x.slack ? 1 : 0;
        // number of non-unit axes that take part in the reference's "1-d like" inner product
        int first_inner_axis = slab_mode ? 1 : 0;
        int nonunit = 0;
        for (int i = first_inner_axis; i < a.nd; ++i)
            if (a.zs[i] != 1) nonunit++;
        a.inner_from_zero = nonunit >= 1 ? 1 : 0;

        if (z.host) {  // host tier: the reference's loop nest, operands and result in host memory
            if (!x.host || !y.host) throw Error("internal: host-tier product with a device operand");
            HK<E>::conv_naive(x.p, x.plane, y.p, y.plane, z.p, z.plane, a);
            return;
        }
        if (x.host || y.host) throw Error("internal: device product with a host operand");

        // Shallow product (a stencil): few terms per output, HBM-bound streaming work — the reference-order per-output
        // kernel (bit-exact) beats the compute-bound tiled kernel and the barrier-per-step staged one by an order of
        // magnitude there (three_populations' Horner steps: 80 -> 8 us).
        if (R.conv_mode == 0 && R.shallow_max_terms && !slab_mode && !accumulate && a.nd >= 1 && a.slab_lo == 0 && a.slab_hi == a.zs[0]) {
            size_t terms = 1;
            for (int i = 0; i < a.nd; ++i) terms *= std::min(a.xs[i], a.ys[i]);
            if (terms <= R.shallow_max_terms) {
                // (one operand a LINE of 16+ coefficients along an outer axis: the tile form, gft_kernels.hip k_conv_line)
                if (K<E>::conv_line(R.stream, x.p, x.plane, y.p, y.plane, z.p, z.plane, a)) {
                    R.stats[5]++;
                    R.stats_shallow[0]++;
                    return;
                }
                ConvEpi e;
                std::memset(&e, 0, sizeof(e));
                for (int i = 0; i < a.nd; ++i) e.os[i] = a.zs[i];
                if (K<E>::conv_shallow(R.stream, x.p, x.plane, y.p, y.plane, z.p, z.plane, a, e)) {
                    R.stats[5]++;
                    R.stats_shallow[0]++;
                    return;
                }
            }
        }
        // Small plain f64 products: the row-pair form of the reference-order product (gft_conv_staged.hip) is bit-exact AND
        // faster than the tiled kernel there (two launches, no planning: 12^3 38 -> 14 us, 16^3 27 -> 20, 64^2 27 -> 18,
        // 100^2 42 -> 27 us; crossover at ~20^3 resp. ~200^2, profiles/r04/f64_pairs_vs_tiled.txt)
        if (W == 1 && R.conv_mode == 0 && R.pairs_first && !slab_mode && !accumulate && a.slab_lo == 0 && a.slab_hi == a.zs[0] && a.nd >= 2) {
            if (conv_pairs<E>(R.stream, x.p, x.plane, y.p, y.plane, z.p, z.plane, a, a.nd == 2 ? R.pairs_first_max_rank2 : (a.nd == 3 ? R.pairs_first_max : 0.1 * R.pairs_first_max))) {  // (rank 4: 8^4 27 vs 22 us already)
                R.stats[4]++;
                return;
            }
        }
        bool want_tiled = (W == 1) && (R.conv_mode == 0 || R.conv_mode == 2);
        // A recurrence step (one output slab k with j0 >= j0_min and/or j0 < k) is a plain slab product of shifted
        // operand views:  sum_{j0 >= m, j0 <= k - e} x[j0] y[k - j0]  =  slab k - m - e of  x[m:] (*) y[e:]  (e = 1
        // for "exclusive").  Only the slabs that can contribute are part of the views, so uninitialised later slabs of
        // the result (which is also an operand in div/exp/log) are never read.  Not bit-exact (tiled order), so only
        // above the tiled crossover; j0_desc is an ordering, irrelevant here.
        const double *tx = x.p, *ty = y.p;
        double* tz = z.p;
        ConvArgs ash = a;  // the problem handed to the tiled kernel
        bool shifted_empty = false;
        if (want_tiled && (a.j0_min || a.j0_excl || a.j0_desc)) {
            const unsigned m = (unsigned)a.j0_min, e = a.j0_excl ? 1u : 0u;
            // div / log steps (exclusive j0) stay on the reference-order kernel: those recurrences subtract and divide,
            // and a different summation order showed up as 3e-10 relative on near-cancelling coefficients at 64^3 —
            // outside the 1e-10 contract — while their time is dominated by the lower-dimensional divisions anyway.
            // exp steps (all-additive) agree to 1e-14 and run 15x faster at 64^3.
            if (a.nd < 2 || a.slab_hi != a.slab_lo + 1 || a.accumulate || a.j0_excl || R.conv_mode == 2) {
                want_tiled = false;
            } else if (a.slab_lo < m + e || a.xs[0] <= m || a.ys[0] <= e) {
                shifted_empty = true;  // no admissible j0: the slab is zero
            } else {
                const unsigned kp = a.slab_lo - m - e;
                ash.j0_min = ash.j0_excl = ash.j0_desc = 0;
                ash.xs[0] = std::min(a.xs[0] - m, kp + 1);
                ash.ys[0] = std::min(a.ys[0] - e, kp + 1);
                ash.zs[0] = kp + 1;
                ash.slab_lo = kp;
                ash.slab_hi = kp + 1;
                tx = x.p + (size_t)m * a.xstr[0];
                ty = y.p + (size_t)e * a.ystr[0];
                tz = z.p + (size_t)(m + e) * a.zstr[0];
                for (int i = 1; i < a.nd; ++i)  // operands never exceed the (virtual) result
                    if (ash.xs[i] > ash.zs[i] || ash.ys[i] > ash.zs[i]) want_tiled = false;
            }
        }
        if (want_tiled && shifted_empty) {
            HV cur = z.index0(a.slab_lo);
            HIP_OK(hipMemsetAsync(cur.p, 0, sizeof(double) * cur.numel(), R.stream));
            return;
        }
        if (want_tiled && ash.nd > 4 && !ash.accumulate && ash.zs[ash.nd - 1] <= 128 && conv_tiled_high_rank(x, y, z, a, ash, tx, ty, tz)) return;
        if (want_tiled) {
            // rank 2, or a last axis longer than the tiled kernel's 128: split the last axis into (P, B) pieces
            ConvArgs at = ash;
            unsigned B = 0;
            const bool split = plan_inner_split(ash, at, &B);
            if (split) at.operands_slack = 0;  // its operands are zero-padded pieces: packed and scanned (the non-finite verdict)
            size_t need = 0;
            bool ok = true;
            if (R.conv_mode == 0) {
                // auto: below this the bit-exact reference-order kernels are as fast (fixed costs dominate).  Decided
                // from the shapes alone, BEFORE the tiled planner is asked: a plan query builds tables, takes an arena
                // slice and a cache entry, which small ever-changing shapes (Genfer's supports grow statement by
                // statement) would pay for nothing.
                double macs = 1.0;
                for (int i = 0; i < ash.nd; ++i) {
                    double f = 0.5 * (double)ash.zs[i] * (double)std::min(ash.xs[i], ash.ys[i]);
                    if (i == 0 && ash.slab_hi - ash.slab_lo < ash.zs[0]) f = (double)(ash.slab_hi - ash.slab_lo) * std::min(ash.xs[0], ash.ys[0]);
                    macs *= f;
                }
                const double tmin = R.tiled_min_override >= 0 ? R.tiled_min_override : R.tiled_min_macs;
                // rank 2 runs the kernel with a 1 x 64 lane tile (half-empty below 64 rows), split products pay the pad /
                // fold passes: their crossovers sit higher (tools/xover_host.py: 32^2 staged 34 us / tiled 48, 48^2 60 / 44)
                if (macs < tmin * (split ? 10.0 : (ash.nd == 2 ? 4.0 : 1.0))) ok = false;
            }
            if (ok) ok = conv_tiled_f64(R.stream, tx, ty, tz, at, nullptr, 0, &need, nullptr, 0);
            if (ok) {
                if (need > R.conv_ws_bytes) {  // grow geometrically: supports (and workspaces) grow statement by statement
                    size_t want = std::max(need, std::min<size_t>(2 * R.conv_ws_bytes, (size_t)1 << 32));
                    want = std::max<size_t>(want, (size_t)8 << 20);
                    if (R.conv_ws) HIP_OK(hipFree(R.conv_ws));
                    R.conv_ws = nullptr;
                    R.conv_ws_bytes = 0;
                    if (hipMalloc(&R.conv_ws, want) != hipSuccess) {
                        (void)hipGetLastError();
                        want = need;
                        HIP_OK(hipMalloc(&R.conv_ws, want));
                    }
                    R.conv_ws_bytes = want;
                }
                // Zero padding times inf/NaN would create NaNs the reference does not produce.  The verdict stays
                // on the device: the packing/scan kernels stamp R.d_flag[2] with this product's epoch if an
                // operand is not finite; the tiled kernels then leave z alone and the guarded reference-order
                // launch below computes it (and is a no-op otherwise).  No host round trip.
                if (++R.nf_epoch == 0) {
                    HIP_OK(hipMemsetD32Async((hipDeviceptr_t)(R.d_flag + 2), 0, 1, R.stream));
                    R.nf_epoch = 1;
                }
                unsigned* flag = R.d_flag + 2;
                bool guarded = true;
                if (!split) {
                    if (!conv_tiled_f64(R.stream, tx, ty, tz, ash, R.conv_ws, R.conv_ws_bytes, &need, flag, R.nf_epoch, &guarded))
                        throw Error("tiled convolution launch failed");
                } else {
                    const ConvArgs& a = ash;  // the (possibly shifted) problem; the guarded fallback below uses the original
                    const double* xsrc = tx;
                    const double* ysrc = ty;
                    double* zdst = tz;
                    const int nd = a.nd;
                    size_t xrows = 1, yrows = 1, zrows = 1, zrows_per0 = 1;
                    for (int i = 0; i + 1 < nd; ++i) {
                        xrows *= a.xs[i];
                        yrows *= a.ys[i];
                        zrows *= a.zs[i];
                        if (i > 0) zrows_per0 *= a.zs[i];
                    }
                    const unsigned Px = at.xs[0], Py = at.ys[0], Pz = at.zs[0], RI = 2 * B - 1;
                    std::shared_ptr<Buf> xt = alloc_doubles(xrows * Px * B), yt = alloc_doubles(yrows * Py * B);
                    std::shared_ptr<Buf> zt = alloc_doubles(zrows * Pz * RI);
                    tiled_pad_rows_f64(R.stream, xsrc, xt->p, xrows, a.xs[nd - 1], Px, B);
                    tiled_pad_rows_f64(R.stream, ysrc, yt->p, yrows, a.ys[nd - 1], Py, B);
                    if (!conv_tiled_f64(R.stream, xt->p, yt->p, zt->p, at, R.conv_ws, R.conv_ws_bytes, &need, flag, R.nf_epoch))
                        throw Error("tiled convolution launch failed");
                    // rank 2: the slab range is a row range; rank 3: slabs of z.shape[1] rows
                    size_t per0 = nd == 2 ? 1 : zrows_per0;
                    tiled_fold_rows_f64(R.stream, zt->p, zdst, zrows, a.slab_lo * per0, a.slab_hi * per0, Pz, B, a.zs[nd - 1],
                                        a.accumulate, flag, R.nf_epoch);
                }
                R.stats[3]++;
                if (!guarded) return;  // operands read in place: no zero padding, no verdict, nothing to fall back from
                a.guard = flag;
                a.guard_epoch = R.nf_epoch;
                if (!conv_staged<E>(R.stream, x.p, x.plane, y.p, y.plane, z.p, z.plane, a, false))
                    K<E>::conv_naive(R.stream, x.p, x.plane, y.p, y.plane, z.p, z.plane, a);
                return;
            }
            if (R.conv_mode == 2) throw Error("conv_mode=2 (tiled) requested but the shape is not supported by the tiled kernel");
        }
        if (R.conv_mode == 0 || R.conv_mode == 3) {
            if (conv_staged<E>(R.stream, x.p, x.plane, y.p, y.plane, z.p, z.plane, a, R.conv_mode == 3)) {
                R.stats[4]++;
                return;
            }
        }
        R.stats[5]++;
        K<E>::conv_naive(R.stream, x.p, x.plane, y.p, y.plane, z.p, z.plane, a);
    }

    // Rank >= 5 on the tiled kernel (the reference's product is rank-generic, mt:984-1012; the kernel takes the last four
    // axes): the leading axes are walked on the host — for every leading output index u and every admissible j <= u one
    // accumulate-mode rank-4 launch  z[u] += x[j] (*) y[u - j]  on the trailing blocks, which are contiguous in the
    // operands' own layout.  Same multiply-adds as the rank-4 kernel performs, summed per u in ascending j: the tiled
    // contract (1e-10), not the reference's order.  false: not worth it / not supported — the caller takes the
    // reference-order kernels.
    static bool conv_tiled_high_rank(const HV& x, const HV& y, const HV& z, const ConvArgs& a, const ConvArgs& ash, const double* tx,
                                     const double* ty, double* tz) {
        const int extra = ash.nd - 4;
        ConvArgs sub;
        std::memset(&sub, 0, sizeof(sub));
        sub.nd = 4;
        double macs = 1.0, lead_pairs = 1.0;
        for (int i = 0; i < ash.nd; ++i) macs *= 0.5 * (double)ash.zs[i] * (double)std::min(ash.xs[i], ash.ys[i]);
        for (int i = 0; i < 4; ++i) {
            sub.xs[i] = ash.xs[extra + i];
            sub.ys[i] = ash.ys[extra + i];
            sub.zs[i] = ash.zs[extra + i];
            sub.xstr[i] = ash.xstr[extra + i];
            sub.ystr[i] = ash.ystr[extra + i];
            sub.zstr[i] = ash.zstr[extra + i];
        }
        sub.slab_lo = 0;
        sub.slab_hi = sub.zs[0];
        sub.variant = ash.variant;
        for (int i = 0; i < extra; ++i) lead_pairs *= 0.5 * (double)ash.zs[i] * (double)std::min(ash.xs[i], ash.ys[i]) + 0.5;
        // every launch must be worth a launch: the trailing rank-4 product above the tiled crossover
        if (R.conv_mode == 0 && macs / lead_pairs < R.tiled_min_macs) return false;
        size_t need = 0;
        if (!conv_tiled_f64(R.stream, tx, ty, tz, sub, nullptr, 0, &need, nullptr, 0)) return false;
        if (need > R.conv_ws_bytes) {
            size_t want = std::max<size_t>(std::max(need, std::min<size_t>(2 * R.conv_ws_bytes, (size_t)1 << 32)), (size_t)8 << 20);
            if (R.conv_ws) HIP_OK(hipFree(R.conv_ws));
            R.conv_ws = nullptr;
            R.conv_ws_bytes = 0;
            HIP_OK(hipMalloc(&R.conv_ws, want));
            R.conv_ws_bytes = want;
        }
        if (++R.nf_epoch == 0) {
            HIP_OK(hipMemsetD32Async((hipDeviceptr_t)(R.d_flag + 2), 0, 1, R.stream));
            R.nf_epoch = 1;
        }
        unsigned* flag = R.d_flag + 2;
        // odometer over the leading output index u (axis 0 restricted to the slab range) and, inside, over j
        unsigned u[MAXD] = {0}, j[MAXD] = {0};
        for (int i = 0; i < extra; ++i) u[i] = i == 0 ? ash.slab_lo : 0;
        const unsigned u0_hi = ash.slab_hi;
        if (u0_hi <= ash.slab_lo) return true;
        for (;;) {
            size_t zoff = 0;
            unsigned jlo[MAXD], jhi[MAXD];
            bool any = true;
            for (int i = 0; i < extra; ++i) {
                zoff += (size_t)u[i] * ash.zstr[i];
                jlo[i] = u[i] + 1 > ash.ys[i] ? u[i] + 1 - ash.ys[i] : 0;
                jhi[i] = std::min(u[i], ash.xs[i] - 1);
                if (jlo[i] > jhi[i]) any = false;
                j[i] = jlo[i];
            }
            bool first = true;
            if (!any) {  // no admissible j (compact operands): the block is zero
                zero_elems(false, tz + zoff, (size_t)ash.zstr[extra - 1]);
            } else {
                for (;;) {
                    size_t xoff = 0, yoff = 0;
                    for (int i = 0; i < extra; ++i) {
                        xoff += (size_t)j[i] * ash.xstr[i];
                        yoff += (size_t)(u[i] - j[i]) * ash.ystr[i];
                    }
                    sub.accumulate = first ? 0 : 1;
                    first = false;
                    if (!conv_tiled_f64(R.stream, tx + xoff, ty + yoff, tz + zoff, sub, R.conv_ws, R.conv_ws_bytes, &need, flag, R.nf_epoch))
                        throw Error("tiled convolution launch failed");
                    int ax = extra - 1;
                    for (; ax >= 0; --ax) {
                        if (++j[ax] <= jhi[ax]) break;
                        j[ax] = jlo[ax];
                    }
                    if (ax < 0) break;
                }
            }
            int ax = extra - 1;
            for (; ax >= 0; --ax) {
                const unsigned hi = ax == 0 ? u0_hi : ash.zs[ax];
                if (++u[ax] < hi) break;
                u[ax] = ax == 0 ? ash.slab_lo : 0;
            }
            if (ax < 0) break;
        }
        R.stats[3]++;
        // non-finite operands: the tiled launches left z alone from the launch that noticed on; the guarded reference-order
        // launch recomputes ALL of z then (accumulate is off here, so nothing of the earlier partial sums survives)
        ConvArgs g = a;
        g.guard = flag;
        g.guard_epoch = R.nf_epoch;
        if (!conv_staged<E>(R.stream, x.p, x.plane, y.p, y.plane, z.p, z.plane, g, false))
            K<E>::conv_naive(R.stream, x.p, x.plane, y.p, y.plane, z.p, z.plane, g);
        return true;
    }

    static P mul_var(const P& self, const double* m, size_t v, const Dims& shape, const Dims& deg) {  // mt:589-608
        if (v >= self.shape.size() || shape.size() != self.shape.size()) throw Error("mul_var: bad axis/shape");
        size_t upper = std::min(shape[v] - 1, self.shape[v]);
        Shifts shift(shape.size(), 0);
        shift[v] = -1;
        Dims src_len = self.shape;
        src_len[v] = upper;
        return gather(self, shape, deg, shift, src_len, OP_MUL_S, m);
    }
    static P mul_linear(const P& self, const double* c, const double* m, size_t v, const Dims& shape, const Dims& deg) {  // mt:611-623
        if (val_is_zero(c)) return mul_var(self, m, v, shape, deg);
        return addsub(mul_var(self, m, v, shape, deg), mul(self, scalar(c)), false);
    }

    // c * eps_v for a still-lazy plain variable (0 + 1*eps_v) and a finite host-known scalar c: the result is
    // (c*0, c*1) = (0 with c's sign, c) by IEEE identities (interval: the exact zero / one short-circuits of
    // interval.rs:164-190), so it can stay lazy — and a later subst_var(.., c*eps_v) knows from the host that the
    // substitution is a pure scaling instead of scanning a 2-element device tensor (one host round trip less).
    static bool scaled_lazy_var(const P& var, const double c[2], P* out) {
        if (var.buf || !var.lazy_lin || var.numel != 2) return false;
        auto is_pz = [](double x) { return x == 0.0 && !std::signbit(x); };
        if (!is_pz(var.cv[0]) || var.cv1[0] != 1.0) return false;
        if (W == 2 && (!is_pz(var.cv[1]) || var.cv1[1] != 1.0)) return false;
        if (!std::isfinite(c[0]) || (W == 2 && !std::isfinite(c[1]))) return false;
        P r = var;
        if (W == 1) {
            r.cv[0] = std::copysign(0.0, c[0]);
            r.cv[1] = 0.0;
        } else {
            r.cv[0] = r.cv[1] = 0.0;
        }
        r.cv1[0] = c[0];
        r.cv1[1] = W == 2 ? c[1] : 0.0;
        *out = r;
        return true;
    }

    static P mul(P self, P other) {  // mt:1014-1072
        Dims deg = min_degrees(self, other);
        if (is_zero(self) || is_zero(other)) return zero_with(deg);
        broadcast(self, other);
        Dims shape = sum_shape(self, other);
        self = truncate_degrees(self, deg);
        other = truncate_degrees(other, deg);
        if (is_one(self)) return other;
        if (is_one(other)) return self;
        double c[2], m[2];
        if (self.numel == 1) {
            first_value(self, c);
            P lazy;
            if (scaled_lazy_var(other, c, &lazy)) return lazy;
            return map_copy(other, OP_LMUL_S, c);
        }
        if (other.numel == 1) {
            first_value(other, c);
            P lazy;
            if (scaled_lazy_var(self, c, &lazy)) return lazy;
            return map_copy(self, OP_LMUL_S, c);
        }
        size_t v;
        ScanCtx sc_self("mul.self");
        if (extract_linear(self, c, m, &v)) {
            Dims sh = other.shape;
            sh[v] = std::min(deg[v], sh[v] + 1);
            return mul_linear(other, c, m, v, sh, deg);
        }
        ScanCtx sc_other("mul.other");
        if (extract_linear(other, c, m, &v)) {
            Dims sh = self.shape;
            sh[v] = std::min(deg[v], sh[v] + 1);
            return mul_linear(self, c, m, v, sh, deg);
        }
        const bool host = tier_host(prod(shape), self, other) && est_macs(self.shape, other.shape, shape) <= R.host_max_macs;
        P out = make(shape, deg, host);
        if (!host && W == 1 && dist_shard(self, other, out)) return out;  // multi-GPU: leading axis sharded (gft_dist_init)
        conv(view(self, host), view(other, host), view(out, host), 0, shape.empty() ? 1 : shape[0], false, false, 0, 0, 0);
        return seal(out);
    }
    static bool dist_shard(const P& self, const P& other, const P& out);  // defined after the RCCL plumbing
    // multiply-adds of a full product, estimated from the shapes (the dispatch criterion for general products)
    static double est_macs(const Dims& xs, const Dims& ys, const Dims& zs) {
        double macs = 1.0;
        int rank = 0;
        for (size_t i = 0; i < zs.size(); ++i) {
            macs *= 0.5 * (double)zs[i] * (double)std::min(xs[i], ys[i]) + 0.5;
            if (zs[i] > 1) rank++;
        }
        return macs * (rank <= 1 ? 1.0 : (rank == 2 ? 1.4 : 2.8));  // host cost in rank-1 multiply-adds (xover_host.txt)
    }

    // ---- division (mt:1162-1231) -----------------------------------------------------------------------------
    static int nonunit_axes(const Dims& s) {
        int n = 0;
        for (size_t x : s)
            if (x != 1) n++;
        return n;
    }
    struct TiledMin {  // tiled-kernel crossover for the products issued inside this scope
        double prev;
        explicit TiledMin(double v) : prev(R.tiled_min_override) { R.tiled_min_override = v; }
        ~TiledMin() { R.tiled_min_override = prev; }
    };
    // tier-dispatched launches used by the recurrences (views carry their side)
    static void x_map_inplace(const HV& v, int op, unsigned u) {
        if (v.host) HK<E>::map_inplace(v.p, v.plane, v.numel(), op, u, Scalar2{0, 0});
        else K<E>::map_inplace(R.stream, v.p, v.plane, v.numel(), op, u, Scalar2{0, 0});
    }
    static void x_block_op(const HV& dst, const HV& src, int op, unsigned u) {
        Dims keep = collapse_mask({&dst.shape}, false);
        if (dst.host) HK<E>::block_op(dview(dst, &keep), dview(src, &keep), op, u);
        else K<E>::block_op(R.stream, dview(dst, &keep), dview(src, &keep), op, u);
    }
    static void x_set_scalar(const HV& dst, Scalar2 v) {
        if (dst.host) E::st(dst.p, dst.plane, 0, E::from(v));
        else K<E>::set_small(R.stream, dst.p, dst.plane, 1, v, v);
    }
    static void copy_planes(bool host, double* dst, size_t dplane, const double* src, size_t splane, size_t n) {
        copy_elems(host, dst, src, n);
        if (W == 2) copy_elems(host, dst + dplane, src + splane, n);
    }
    // z[0..m) += a (*) b for the m slabs that follow a slab the recurrence has just finalised (right-looking update, see
    // div_rec).  Only z[0] is needed by the next step of the recurrence: it is updated on the main stream, the other m - 1
    // slabs on the side stream, overlapping the next slab's division (a single-workgroup latency chain that leaves the
    // GPU empty).  Order per slab is unchanged — slab s receives term j from bulk(j) for j < s - 1 (side stream, in
    // order), then from the critical update of step s - 1, which waits for the latest bulk first.
    static void right_update(const HV& a, const HV& b, const HV& z, size_t m) {
        TiledMin guard(R.recur_tiled_min_macs);
        // The overlap is only legal for launches that own nothing but their arguments: the reference-order kernels.  A
        // tiled product takes pooled temporaries (freed for reuse by the MAIN stream as soon as conv() returns), the
        // one conv workspace and the non-finite epoch word — shared state that two streams must not touch at once.  So
        // whenever a product of this recurrence could take the tiled kernel ("recur_tiled_min_macs" lowered, or
        // conv_mode 2) everything stays on the main stream.
        const bool may_tile = W == 1 && (R.conv_mode == 2 || (R.conv_mode == 0 && R.recur_tiled_min_macs < 1.0e299));
        if (!R.recur_overlap || !R.side || m < 2 || may_tile) {
            join_side();
            conv(a, b, z, 0, m, true, false, 0, 0, 0);
            return;
        }
        join_side();  // the previous bulk added its term to z[0] (and beyond)
        conv(a, b, z, 0, 1, true, false, 0, 0, 0);
        {
            hipStream_t ms = R.stream, ss = R.side;
            hipEvent_t ev = R.ev_main;
            enqueue_task([=] {  // slab final (+ critical update): the bulk may read it
                lq_note((hipEventRecord)(ev, ms), nullptr, "hipEventRecord (main stream)");
                lq_note((hipStreamWaitEvent)(ss, ev, 0), nullptr, "hipStreamWaitEvent (side stream)");
            });
        }
        std::swap(R.stream, R.side);
        try {
            conv(a, b, z, 1, m, true, false, 0, 0, 0);
        } catch (...) {
            std::swap(R.stream, R.side);
            throw;
        }
        std::swap(R.stream, R.side);
        {
            hipStream_t ss = R.side;
            hipEvent_t ev = R.ev_bulk;
            enqueue_task([=] { lq_note((hipEventRecord)(ev, ss), nullptr, "hipEventRecord (side stream)"); });
        }
        R.side_pending = true;
    }
    // Unwinding out of a recurrence with a bulk update in flight: its operands (rsbuf / tmp / the quotient itself) are
    // about to be released, so the side stream is drained first.
    struct SideDrain {
        ~SideDrain() {
            if (R.side_pending && std::uncaught_exceptions()) {
                launch_drain_nothrow();  // (a destructor during unwinding: a latched launch failure stays latched)
                (void)(hipStreamSynchronize)(R.side);
                R.side_pending = false;
            }
        }
    };
    // the main stream waits for everything issued on the side stream so far
    static void join_side() {
        if (!R.side_pending) return;
        {
            hipStream_t ms = R.stream;
            hipEvent_t ev = R.ev_bulk;
            enqueue_task([=] { lq_note((hipStreamWaitEvent)(ms, ev, 0), nullptr, "hipStreamWaitEvent (main stream)"); });
        }
        R.side_pending = false;
    }
    static void div_rec(const HV& xs, const HV& ys, const HV& res) {
        if (xs.numel() == 0) return;
        SideDrain drain_on_unwind;
        const bool host = res.host;
        if (res.shape.empty()) {
            if (host) E::st(res.p, res.plane, 0, E::div(E::ld(xs.p, xs.plane, 0), E::ld(ys.p, ys.plane, 0)));
            else K<E>::scalar_op(R.stream, SC_DIV, xs.p, xs.plane, ys.p, ys.plane, res.p, res.plane);
            return;
        }
        if (res.shape.size() == 1) {  // last level: fused sequential recurrence
            if (host)
                HK<E>::div_1d(xs.p, xs.plane, (unsigned)xs.shape[0], ys.p, ys.plane, (unsigned)ys.shape[0], res.p, res.plane,
                              (unsigned)res.shape[0]);
            else
                K<E>::div_1d(R.stream, xs.p, xs.plane, (unsigned)xs.shape[0], ys.p, ys.plane, (unsigned)ys.shape[0], res.p,
                             res.plane, (unsigned)res.shape[0]);
            return;
        }
        if (!host && R.div2d && res.shape.size() == 2 &&
            K<E>::div_2d(R.stream, xs.p, xs.plane, (unsigned)xs.shape[0], (unsigned)xs.shape[1], xs.shape[1], ys.p, ys.plane,
                         (unsigned)ys.shape[0], (unsigned)ys.shape[1], res.p, res.plane, (unsigned)res.shape[0], (unsigned)res.shape[1], 0))
            return;  // the last two axes in one launch (gft_div2d.hip), same bits
        size_t n0 = res.shape[0];
        HV y0 = ys.index0(0);
        // Device tier: RIGHT-LOOKING accumulation.  The reference forms cur = sum_{j<k} res[j] (*) ys[k-j] when it reaches
        // slab k (one product per (k, j): 64 small dependent launches a slab).  Here the quotient's own memory holds the
        // running sums: as soon as slab k is final, ONE product adds res[k] (*) ys[1..m] into the m slabs that follow.
        // Every slab still receives its terms in ascending j, each term's row products formed from zero (mt:971-982), so
        // the bits are the reference's; what changes is that a step is a product of m slabs wide — hundreds of
        // workgroups instead of four.  Steps of at least recur_tiled_min_macs multiply-adds may take the tiled kernel
        // (different summation order: 1e-10 contract), smaller ones keep the reference order.
        const bool right = !host && R.div2d;
        Dims rest(res.shape.begin() + 1, res.shape.end()), yrest(ys.shape.begin() + 1, ys.shape.end());
        if (right) {
            zero_elems(false, res.p, res.numel());
            if (W == 2) zero_elems(false, res.p + res.plane, res.numel());
        }
        for (size_t k = 0; k < n0; ++k) {
            HV cur = res.index0(k);
            if (!right) conv(res, ys, res, k, k + 1, false, true, 0, 1, 0);  // cur = sum_{j<k} res[j] (*) ys[k-j]
            struct Scatter {  // runs when slab k is final, whichever branch below finalised it
                const HV &res, &ys, &cur;
                const Dims &rest, &yrest;
                size_t k, n0;
                bool on;
                ~Scatter() noexcept(false) {
                    if (!on || k + 1 >= n0 || ys.shape[0] < 2 || std::uncaught_exceptions()) return;
                    const size_t m = std::min(n0 - 1 - k, ys.shape[0] - 1);
                    Dims xs1{1}, ysm{m}, zsm{m};
                    xs1.insert(xs1.end(), rest.begin(), rest.end());
                    ysm.insert(ysm.end(), yrest.begin(), yrest.end());
                    zsm.insert(zsm.end(), rest.begin(), rest.end());
                    HV xk{cur.p, res.plane, xs1, false}, ym{ys.p + prod(yrest), ys.plane, ysm, false},
                        zm{res.p + (k + 1) * prod(rest), res.plane, zsm, false};
                    right_update(xk, ym, zm, m);
                }
            } scatter{res, ys, cur, rest, yrest, k, n0, right};
            if (!host && R.div2d && cur.shape.size() == 2) {
                // neg, += xs[k], copy and the whole 2-d division of the slab fused into one launch
                const bool have_x = k < xs.shape[0];
                HV xk = have_x ? xs.index0(k) : HV{nullptr, 0, Dims{0, 0}, false};
                if (K<E>::div_2d(R.stream, xk.p, xk.plane, have_x ? (unsigned)xk.shape[0] : 0u, have_x ? (unsigned)xk.shape[1] : 0u,
                                 have_x ? xk.shape[1] : 0, y0.p, y0.plane, (unsigned)y0.shape[0], (unsigned)y0.shape[1], cur.p, cur.plane,
                                 (unsigned)cur.shape[0], (unsigned)cur.shape[1], 1))
                    continue;
            }
            if (!host && R.div2d && cur.shape.size() == 1) {
                // 2-d quotient whose rows do not fit the slab kernel: row by row, each row's neg, += xs[k], copy and 1-d
                // division in one launch
                const bool have_x = k < xs.shape[0];
                HV xk = have_x ? xs.index0(k) : HV{nullptr, 0, Dims{0}, false};
                if (K<E>::div_1d(R.stream, xk.p, xk.plane, have_x ? (unsigned)xk.shape[0] : 0u, y0.p, y0.plane, (unsigned)y0.shape[0], cur.p,
                                 cur.plane, (unsigned)cur.shape[0], 1))
                    continue;
            }
            x_map_inplace(cur, MAP_NEG, 0);
            if (k < xs.shape[0]) x_block_op(cur, xs.index0(k), BLK_ADD, 0);
            std::shared_ptr<Buf> tmp = alloc_tier(host, cur.numel() * W);
            HV copy{tmp->p, cur.numel(), cur.shape, host};
            copy_planes(host, copy.p, copy.plane, cur.p, cur.plane, cur.numel());
            div_rec(copy, y0, cur);
        }
        if (right) join_side();
    }
    static P div(P self, P other) {
        broadcast(self, other);
        Dims deg = min_degrees(self, other);
        self = truncate_degrees(self, deg);
        other = truncate_degrees(other, deg);
        if (is_one(other)) return self;
        if (other.numel == 1) {
            double c[2];
            first_value(other, c);
            return map_copy(self, OP_DIV_S, c);
        }
        Dims rs = deg;
        for (size_t i = 0; i < rs.size(); ++i)
            if (other.shape[i] == 1) rs[i] = self.shape[i];
        for (size_t i = 0; i < rs.size(); ++i)
            if (rs[i] == UMAX) throw Error("div: untruncated result shape (degrees_p1 == usize::MAX)");
        const bool host = tier_host(prod(rs), self, other) && est_macs(rs, other.shape, rs) <= R.host_max_macs;
        P out = make(rs, deg, host);
        if (!host && div_wavefront(self, other, out)) return out;
        div_rec(view(self, host), view(other, host), view(out, host));
        return seal(out);
    }
    // The whole quotient in one launch (gft_div2d.hip k_div_wavefront): every row a task of one wave, consumed in the
    // reference's order, dependencies through per-row flags.  Ranks 2-4 (after dropping the axes on which all three
    // tensors are trivial) with rows of at most 64 coefficients and enough rows to be worth a persistent launch.
    static bool div_wavefront(const P& self, const P& other, const P& out) {
        if (!R.div_wavefront || !R.div2d) return false;
        Dims keep = collapse_mask({&out.shape}, false);
        if (keep.size() < 2 || keep.size() > 4) return false;
        // dropped axes have extent 1 in the result, hence in both operands (shapes never exceed the result's)
        HV x{dp<E>(self), self.numel, pick(self.shape, keep), false}, y{dp<E>(other), other.numel, pick(other.shape, keep), false},
            z{dp<E>(out), out.numel, pick(out.shape, keep), false};
        return div_wavefront_hv(x, y, z);
    }
    // (contiguous views of rank 2-4, no unit axes to drop)
    static bool div_wavefront_hv(const HV& x, const HV& y, const HV& z) {
        if (!R.div_wavefront || !R.div2d) return false;
        const size_t nd = z.shape.size();
        if (nd < 2 || nd > 4) return false;
        unsigned xs[4], ys[4], zs[4];
        size_t rows = 1;
        for (size_t i = 0; i < nd; ++i) {
            xs[i] = (unsigned)x.shape[i];
            ys[i] = (unsigned)y.shape[i];
            zs[i] = (unsigned)z.shape[i];
            if (i + 1 < nd) rows *= zs[i];
        }
        if (nd == 2 && zs[1] > 64 && zs[1] <= 4096 && rows >= 8 && R.rows_wavefront) {
            // long rows, rank 2: the coefficient-level wavefront (tasks are 64-coefficient segments of rows)
            const size_t words = rows * ((zs[1] + 63) / 64) + 1;
            std::shared_ptr<Buf> fl = alloc_doubles((words + 1) / 2 + 1);
            zero_elems(false, fl->p, (words + 1) / 2 + 1);
            return K<E>::rows_wavefront(R.stream, 0, x.p, x.plane, xs, y.p, y.plane, ys, z.p, z.plane, zs, nullptr, 0, reinterpret_cast<unsigned*>(fl->p));
        }
        if (nd >= 3 && zs[nd - 1] > 64 && zs[nd - 1] <= 4096 && rows >= 8) {
            // long rows, rank 3 / 4 (round 6): the segment wavefront with leading axes — one launch instead of the slab-by-slab form
            const size_t words = rows * ((zs[nd - 1] + 63) / 64) + 1;
            std::shared_ptr<Buf> fl = alloc_doubles((words + 1) / 2 + 1);
            zero_elems(false, fl->p, (words + 1) / 2 + 1);
            return K<E>::seg_wavefront(R.stream, 0, x.p, x.plane, xs, y.p, y.plane, ys, z.p, z.plane, zs, (int)nd, nullptr, 0, reinterpret_cast<unsigned*>(fl->p));
        }
        if (zs[nd - 1] > 64 || zs[nd - 1] < 2 || rows < 64) return false;
        std::shared_ptr<Buf> fl = alloc_doubles((rows + 1 + 1) / 2 + 1);
        zero_elems(false, fl->p, (rows + 1 + 1) / 2 + 1);
        return K<E>::div_wavefront(R.stream, x.p, x.plane, xs, y.p, y.plane, ys, z.p, z.plane, zs, (int)nd, reinterpret_cast<unsigned*>(fl->p));
    }
    // ---- exp / log (mt:406-430, 1270-1386) ---------------------------------------------------------------------
    // xs scaled slab-wise by T::from(j) along axis 0 (mt:1308-1310): xs[j] * j
    static std::shared_ptr<Buf> scaled_by_index(const HV& xs, HV* out) {
        const bool host = xs.host;
        std::shared_ptr<Buf> buf = alloc_tier(host, xs.numel() * W);
        std::shared_ptr<Buf> tab = cached_table(TAB_INDEX, 0, xs.shape[0], host);
        GatherArgs a;
        std::memset(&a, 0, sizeof(a));
        size_t inner = xs.numel() / std::max<size_t>(xs.shape[0], 1);
        a.out.nd = 2;
        a.out.d[0] = (unsigned)xs.shape[0];
        a.out.d[1] = (unsigned)inner;
        a.src_len[0] = a.out.d[0];
        a.src_len[1] = a.out.d[1];
        a.src_stride[0] = inner;
        a.src_stride[1] = 1;
        a.op = OP_MUL_TAB;
        a.tab_axis = 0;
        a.tab = tab->p;
        a.tab_plane = xs.shape[0];
        *out = HV{buf->p, xs.numel(), xs.shape, host};
        if (host) HK<E>::gather(xs.p, xs.plane, buf->p, xs.numel(), a);
        else K<E>::gather(R.stream, xs.p, xs.plane, buf->p, xs.numel(), a);
        return buf;
    }
    // The scalar seeds exp(xs[0]) / ln(xs[0]) (mt:1286-1289, 1336-1339; f64.rs:54-61 = the platform libm) are formed on
    // the host from the constant term — one value, SURVEY §8a row S — so that they are the same libm result whichever
    // side runs the recurrence; every coefficient operation after the seed is device (or host-tier) arithmetic.
    // The 1-d base case of exp / log on DEVICE tensors.  Its order (the newest coefficient's term first, mt:1296-1310)
    // makes it one serial chain of n^2/2 multiply-adds: a single GPU lane needs 75 ns a step (1.5 ms for a 200-long line,
    // 30 ms for 900 — slower than the CPU from ~100 coefficients on), a host core 1 ns.  Lines of 48 coefficients or more
    // therefore make the round trip: the argument line to pinned host memory, the host tier's own loop (the same functor,
    // the same bits), the result line back.  ~25 us of synchronisation against milliseconds.
    template <class F>
    static bool line_on_host(const HV& xs, const HV& res, F&& compute) {
        const size_t nxh = xs.numel(), nrh = res.numel();
        if (nrh < 48 || nrh > 65536) return false;
        std::vector<double> hx(nxh * W), hr(nrh * W);
        for (size_t pl = 0; pl < (size_t)W; ++pl)
            HIP_OK(hipMemcpyAsync(hx.data() + pl * nxh, xs.p + pl * xs.plane, sizeof(double) * nxh, hipMemcpyDeviceToHost, R.stream));
        HIP_OK(hipStreamSynchronize(R.stream));
        compute(hx.data(), nxh, hr.data(), nrh);
        for (size_t pl = 0; pl < (size_t)W; ++pl)
            HIP_OK(hipMemcpyAsync(res.p + pl * res.plane, hr.data() + pl * nrh, sizeof(double) * nrh, hipMemcpyHostToDevice, R.stream));
        HIP_OK(hipStreamSynchronize(R.stream));  // hr is pageable and dies here
        R.stats[6]++;
        return true;
    }
    static void exp_rec(const HV& xs, const HV& res, Scalar2 seed) {
        if (xs.numel() == 0) return;
        if (res.shape.empty()) {
            x_set_scalar(res, seed);
            return;
        }
        if (nonunit_axes(res.shape) == 1) {
            if (res.host) HK<E>::exp_1d(xs.p, xs.plane, (unsigned)xs.numel(), res.p, res.plane, (unsigned)res.numel(), seed);
            else if (!line_on_host(xs, res, [&](const double* hx, size_t nxh, double* hr, size_t nrh) {
                         HK<E>::exp_1d(hx, nxh, (unsigned)nxh, hr, nrh, (unsigned)nrh, seed);
                     }))
                K<E>::exp_1d(R.stream, xs.p, xs.plane, (unsigned)xs.numel(), res.p, res.plane, (unsigned)res.numel(), seed);
            return;
        }
        exp_rec(xs.index0(0), res.index0(0), seed);
        if (res.shape[0] <= 1) return;
        // below the right-looking tiled path's crossover (and for intervals at every size): the row wavefront, one launch,
        // the reference's summation order
        double total = 1.0;
        for (size_t i = 0; i < res.shape.size(); ++i) total *= 0.5 * (double)res.shape[i] * (double)std::min(xs.shape[i], res.shape[i]) + 0.5;
        const bool right_tiled = !res.host && W == 1 && R.conv_mode == 0 && R.exp_right && total >= 64.0 * R.tiled_min_macs;
        // (rank 2 with long rows: the coefficient-level wavefront keeps the reference's order AND beats the right-looking tiled
        // form — 400^2: see profiles/r04/recurrences.txt)
        // rank 2 where the right-looking tiled form would be taken: the wavefront kernels with each row's terms in the order of
        // their ARRIVAL (that form's order, same 1e-10 contract) — 400^2 62 -> 4.5 ms, 1000 x 32 10.6 -> see recurrences.txt
        const bool wf_2d = res.shape.size() == 2 && res.shape[1] <= 4096 && R.rows_wavefront;
        if (!res.host && (!right_tiled || wf_2d) && exp_wavefront(xs, res, right_tiled)) return;
        HV xsc;
        std::shared_ptr<Buf> hold = scaled_by_index(xs, &xsc);
        // Large f64 exponentials (their slab steps would take the tiled kernel anyway, i.e. the 1e-10 contract, not the
        // reference's summation order): RIGHT-LOOKING.  As soon as res[i] is final one wide product adds
        // (j xs[j]) (*) res[i] for j = 1..m into the m slabs that follow — 64 wide launches at full tile efficiency
        // instead of 64 single-slab launches that are all launch / reduce overhead (64^3: 15.6 -> see recurrences.txt).
        const size_t n0 = res.shape[0];
        if (right_tiled) {
            Dims rest(res.shape.begin() + 1, res.shape.end()), xrest(xs.shape.begin() + 1, xs.shape.end());
            HV tail = res.index0(1);
            zero_elems(false, tail.p, res.numel() - prod(rest));
            for (size_t i = 0; i + 1 < n0; ++i) {
                HV cur = res.index0(i);
                if (i >= 1) x_map_inplace(cur, MAP_DIV_U32, (unsigned)i);  // res[i] = acc / i: final
                if (xs.shape[0] < 2) continue;
                const size_t m = std::min(n0 - 1 - i, xs.shape[0] - 1);
                Dims xsm{m}, ys1{1}, zsm{m};
                xsm.insert(xsm.end(), xrest.begin(), xrest.end());
                ys1.insert(ys1.end(), rest.begin(), rest.end());
                zsm.insert(zsm.end(), rest.begin(), rest.end());
                HV xm{xsc.p + prod(xrest), xsc.plane, xsm, false}, yi{cur.p, res.plane, ys1, false},
                    zm{res.p + (i + 1) * prod(rest), res.plane, zsm, false};
                conv(xm, yi, zm, 0, m, true, false, 0, 0, 0);
            }
            x_map_inplace(res.index0(n0 - 1), MAP_DIV_U32, (unsigned)(n0 - 1));
            return;
        }
        for (size_t k = 1; k < n0; ++k) {
            HV cur = res.index0(k);
            conv(xsc, res, res, k, k + 1, false, true, 1, 0, 0);
            x_map_inplace(cur, MAP_DIV_U32, (unsigned)k);
        }
    }
    // `arrival_order`: the caller would otherwise take the right-looking tiled form (1e-10 contract) — the long-row kernel may
    // then add each row's terms in the order the source rows become available instead of the reference's
    static bool exp_wavefront(const HV& xs, const HV& res, bool arrival_order = false) {
        if (!R.div_wavefront || !R.div2d) return false;
        const size_t nd = res.shape.size();
        if (nd < 2 || nd > 4 || xs.shape.size() != nd) return false;
        unsigned xsh[4], rsh[4];
        size_t rows = 1;
        for (size_t i = 0; i < nd; ++i) {
            if (res.shape[i] < 2 || xs.shape[i] > res.shape[i] || xs.shape[i] == 0 || res.shape[i] > 0x7fffffffu) return false;
            xsh[i] = (unsigned)xs.shape[i];
            rsh[i] = (unsigned)res.shape[i];
            if (i + 1 < nd) rows *= res.shape[i];
        }
        if (nd == 2 && rsh[1] > 64 && rsh[1] <= 4096 && rows >= 8 && R.rows_wavefront) {  // long rows: the coefficient-level wavefront
            const size_t words = rows * ((rsh[1] + 63) / 64) + 1;
            std::shared_ptr<Buf> fl = alloc_doubles((words + 1) / 2 + 1);
            zero_elems(false, fl->p, (words + 1) / 2 + 1);
            return K<E>::rows_wavefront(R.stream, arrival_order ? 2 | 4 : 2, xs.p, xs.plane, xsh, xs.p, xs.plane, xsh, res.p, res.plane, rsh, nullptr, 0,
                                        reinterpret_cast<unsigned*>(fl->p));
        }
        if (rsh[nd - 1] > 64 || rows < 8) return false;  // (the alternative is two launches per slab)
        std::shared_ptr<Buf> fl = alloc_doubles((rows + 2) / 2 + 1);
        zero_elems(false, fl->p, (rows + 2) / 2 + 1);
        return K<E>::exp_wavefront(R.stream, xs.p, xs.plane, xsh, res.p, res.plane, rsh, (int)nd, reinterpret_cast<unsigned*>(fl->p), arrival_order ? 1 : 0);
    }
    static Dims explog_shape(const P& a) {
        Dims rs = a.deg;
        for (size_t i = 0; i < rs.size(); ++i) {
            if (a.shape[i] == 1) rs[i] = 1;
            if (rs[i] == UMAX) throw Error("exp/log: untruncated result shape (degrees_p1 == usize::MAX)");
        }
        return rs;
    }
    static Scalar2 seed_of(const P& a, bool is_exp) {
        double c[2];
        first_value(a, c);
        typename E::V r = is_exp ? E::exp(E::from(hv(c))) : E::log(E::from(hv(c)));
        double o[2] = {0.0, 0.0};
        E::st(o, 1, 0, r);
        return Scalar2{o[0], o[1]};
    }
    static P exp(const P& a) {
        Dims rs = explog_shape(a);
        const bool host = tier_host(prod(rs), a) && est_macs(a.shape, rs, rs) <= R.host_max_macs;
        P out = make(rs, a.deg, host);
        exp_rec(view(a, host), view(out, host), seed_of(a, true));
        return seal(out);
    }

    static void log_rec(const HV& xs, const HV& res, Scalar2 seed) {
        if (xs.numel() == 0) return;
        SideDrain drain_on_unwind;
        const bool host = res.host;
        if (res.shape.empty()) {
            x_set_scalar(res, seed);
            return;
        }
        if (nonunit_axes(xs.shape) == 1) {
            if (nonunit_axes(res.shape) != 1) throw Error("log: called `Option::unwrap()` on a `None` value (mt:1346)");
            if (host) HK<E>::log_1d(xs.p, xs.plane, (unsigned)xs.numel(), res.p, res.plane, (unsigned)res.numel(), seed);
            else if (!line_on_host(xs, res, [&](const double* hx, size_t nxh, double* hr, size_t nrh) {
                         HK<E>::log_1d(hx, nxh, (unsigned)nxh, hr, nrh, (unsigned)nrh, seed);
                     }))
                K<E>::log_1d(R.stream, xs.p, xs.plane, (unsigned)xs.numel(), res.p, res.plane, (unsigned)res.numel(), seed);
            return;
        }
        log_rec(xs.index0(0), res.index0(0), seed);
        size_t n0 = res.shape[0];
        if (n0 <= 1) return;
        if (!host && log_wavefront(xs, res)) return;
        // rs[j] = res[j] * j, filled slab by slab as res becomes known (mt:1362-1365)
        std::shared_ptr<Buf> rsbuf = alloc_tier(host, res.numel() * W);
        HV rs{rsbuf->p, res.numel(), res.shape, host};
        zero_elems(host, rs.p, res.numel() * W);
        Dims sub(res.shape.begin() + 1, res.shape.end());
        HV x0 = xs.index0(0);
        // device tier: right-looking accumulation like div_rec — once rs[k] = res[k] * k is known, one product adds
        // xs[1..m] (*) rs[k] into the m slabs that follow (terms arrive in ascending j, as in the reference)
        const bool right = !host && R.div2d;
        Dims xrest(xs.shape.begin() + 1, xs.shape.end());
        if (right) {
            HV tail{res.p + prod(sub), res.plane, res.shape, false};
            zero_elems(false, tail.p, res.numel() - prod(sub));
            if (W == 2) zero_elems(false, tail.p + res.plane, res.numel() - prod(sub));
        }
        for (size_t k = 1; k < n0; ++k) {
            HV cur = res.index0(k);
            if (!right) conv(xs, rs, res, k, k + 1, false, true, 1, 1, 1);  // sum_{j} xs[k-j] (*) (res[j]*j), j ascending
            HV rk = rs.index0(k);
            bool step_done = false;
            if (right && cur.shape.size() == 2 && x0.numel() > 1) {
                // the whole slab step in one launch (gft_div2d.hip, fused == 2): neg, += k * xs[k], the 2-d division by xs[0]
                // — what Div's dispatcher (mt:1194-1231) comes to for a divisor of more than one coefficient —, / k, and
                // rs[k] = res[k] * k; element for element the sequence below
                const bool have_x = k < xs.shape[0];
                HV xk = have_x ? xs.index0(k) : HV{nullptr, 0, Dims{0, 0}, false};
                step_done = K<E>::div_2d(R.stream, xk.p, xk.plane, have_x ? (unsigned)xk.shape[0] : 0u, have_x ? (unsigned)xk.shape[1] : 0u,
                                         have_x ? xk.shape[1] : 0, x0.p, x0.plane, (unsigned)x0.shape[0], (unsigned)x0.shape[1], cur.p,
                                         cur.plane, (unsigned)cur.shape[0], (unsigned)cur.shape[1], 2, (unsigned)k, rk.p, rk.plane);
            }
            if (!step_done) {
                x_map_inplace(cur, MAP_NEG, 0);
                if (k < xs.shape[0]) x_block_op(cur, xs.index0(k), BLK_ADD_U32_TIMES, (unsigned)k);
                // current = current / xs[0] as full TaylorPoly division with degrees = current.shape (mt:1376-1383)
                P num = make(sub, sub, host), den = make(x0.shape, sub, host);
                copy_planes(host, tp<E>(num, host), num.numel, cur.p, cur.plane, cur.numel());
                copy_planes(host, tp<E>(den, host), den.numel, x0.p, x0.plane, x0.numel());
                P q = div_same_tier(num, den, host);
                if (q.shape != sub) throw Error("log: internal shape mismatch after division");
                copy_planes(host, cur.p, cur.plane, tp<E>(q, host), q.numel, cur.numel());
                x_map_inplace(cur, MAP_DIV_U32, (unsigned)k);
                copy_planes(host, rk.p, rk.plane, cur.p, cur.plane, cur.numel());
                x_map_inplace(rk, MAP_MUL_U32, (unsigned)k);
            }
            if (right && k + 1 < n0 && xs.shape[0] >= 2) {
                const size_t m = std::min(n0 - 1 - k, xs.shape[0] - 1);
                Dims xsm{m}, ys1{1}, zsm{m};
                xsm.insert(xsm.end(), xrest.begin(), xrest.end());
                ys1.insert(ys1.end(), sub.begin(), sub.end());
                zsm.insert(zsm.end(), sub.begin(), sub.end());
                HV xm{xs.p + prod(xrest), xs.plane, xsm, false}, yk{rk.p, rs.plane, ys1, false},
                    zm{res.p + (k + 1) * prod(sub), res.plane, zsm, false};
                right_update(xm, yk, zm, m);
            }
        }
        if (right) join_side();
    }
    // The slabs k0 >= 1 of the log recurrence in one launch (gft_div2d.hip k_div_wavefront, log_mode): every row a task,
    // consumed in the reference's order (mt:1335-1386), same bits.  Shapes: no unit axes in the result, rows of at most 64
    // coefficients, a divisor xs[0] with more than one coefficient (Div's general path, mt:1194-1231), enough rows.
    static bool log_wavefront(const HV& xs, const HV& res) {
        if (!R.div_wavefront || !R.div2d) return false;
        const size_t nd = res.shape.size();
        if (nd < 2 || nd > 4 || xs.shape.size() != nd) return false;
        unsigned xsh[4], rsh[4];
        size_t rows = 1, x0n = 1;
        for (size_t i = 0; i < nd; ++i) {
            if (res.shape[i] < 2 || xs.shape[i] > res.shape[i] || res.shape[i] > 0x7fffffffu) return false;
            xsh[i] = (unsigned)xs.shape[i];
            rsh[i] = (unsigned)res.shape[i];
            if (i + 1 < nd) rows *= res.shape[i];
            if (i > 0) x0n *= xs.shape[i];
        }
        if (nd == 2 && rsh[1] > 64 && rsh[1] <= 4096 && rows >= 8 && x0n >= 2 && nonunit_axes(xs.shape) >= 2 && R.rows_wavefront) {
            // long rows: the coefficient-level wavefront
            std::shared_ptr<Buf> qb = alloc_doubles(res.numel() * W);
            const size_t words = rows * ((rsh[1] + 63) / 64) + 1;
            std::shared_ptr<Buf> fl = alloc_doubles((words + 1) / 2 + 1);
            zero_elems(false, fl->p, (words + 1) / 2 + 1);
            return K<E>::rows_wavefront(R.stream, 1, xs.p, xs.plane, xsh, xs.p, xs.plane, xsh, res.p, res.plane, rsh, qb->p, res.numel(),
                                        reinterpret_cast<unsigned*>(fl->p));
        }
        if (nd >= 3 && rsh[nd - 1] > 64 && rsh[nd - 1] <= 4096 && rows >= 8 && x0n >= 2 && nonunit_axes(xs.shape) >= 2) {
            // long rows, rank 3 / 4 (round 6): the segment wavefront with leading axes
            std::shared_ptr<Buf> qb = alloc_doubles(res.numel() * W);
            const size_t words = rows * ((rsh[nd - 1] + 63) / 64) + 1;
            std::shared_ptr<Buf> fl = alloc_doubles((words + 1) / 2 + 1);
            zero_elems(false, fl->p, (words + 1) / 2 + 1);
            return K<E>::seg_wavefront(R.stream, 1, xs.p, xs.plane, xsh, xs.p, xs.plane, xsh, res.p, res.plane, rsh, (int)nd, qb->p, res.numel(),
                                       reinterpret_cast<unsigned*>(fl->p));
        }
        if (rsh[nd - 1] > 64 || rows < 8 || x0n < 2 || nonunit_axes(xs.shape) < 2) return false;  // (the alternative is 2+ launches per slab)
        std::shared_ptr<Buf> qb = alloc_doubles(res.numel() * W);
        std::shared_ptr<Buf> fl = alloc_doubles((rows + 2) / 2 + 1);
        zero_elems(false, fl->p, (rows + 2) / 2 + 1);
        return K<E>::log_wavefront(R.stream, xs.p, xs.plane, xsh, res.p, res.plane, rsh, (int)nd, qb->p, res.numel(), reinterpret_cast<unsigned*>(fl->p));
    }
    // Div's dispatcher (mt:1194-1231) for the slab division inside log.  tp<E>() serves a device caller whatever side
    // the quotient is on; a host caller needs it in host memory.
    static P div_same_tier(const P& num, const P& den, bool host) {
        P q = div(num, den);
        if (host && !on_host(q)) {  // cannot happen while the dispatch criterion is monotone in the shapes; stay correct anyway
            P h = make(q.shape, q.deg, true);
            HIP_OK(hipMemcpyAsync(hp<E>(h), dp<E>(q), sizeof(double) * q.numel * W, hipMemcpyDeviceToHost, R.stream));
            HIP_OK(hipStreamSynchronize(R.stream));
            return h;
        }
        return q;
    }
    static P log(const P& a) {
        Dims rs = explog_shape(a);
        const bool host = tier_host(prod(rs), a) && est_macs(a.shape, rs, rs) <= R.host_max_macs;
        P out = make(rs, a.deg, host);
        log_rec(view(a, host), view(out, host), seed_of(a, false));
        return seal(out);
    }

    static P pow(const P& a, uint32_t e) {  // mt:433-451
        double one[2] = {1.0, 1.0};
        if (e == 0) return scalar(one);
        if (e == 1) return a;
        P res = scalar(one);
        P base = a;
        while (e > 0) {
            if (e & 1) res = mul(res, base);
            base = mul(base, base);  // includes the reference's redundant final squaring
            e >>= 1;
        }
        return res;
    }

    // Device-resident factor tables for derivative / coefficient expansion / index scaling depend only on
    // (kind, n, len): computed once by k_factor_table (reference operation order) and reused.
    static std::shared_ptr<Buf> cached_table(int table_op, size_t n, size_t len, bool host = false) {
        static std::map<std::tuple<int, size_t, size_t>, std::shared_ptr<Buf>> caches[2];
        auto& cache = caches[host ? 1 : 0];
        auto key = std::make_tuple(table_op, n, len);
        auto it = cache.find(key);
        if (it != cache.end()) {
            use_buf(it->second.get());  // (a table another stream uploaded: ordered before this stream's next launch)
            return it->second;
        }
        if (cache.size() > 4096) cache.clear();
        std::shared_ptr<Buf> tab = alloc_tier(host, len * W);
        if (host) HK<E>::factor_table(table_op, (unsigned)n, (unsigned)len, nullptr, 0, tab->p, len);
        else if (len * W <= 1920 && [] {
                     static const bool on = true;
                     return on;
                 }()) {
            // a data-independent table is a serial chain (a running product): one GPU lane takes 8 us for 200 f64 factors
            // and 90 us for 200 interval ones, a host core well under a microsecond — same functor, same bits; the values
            // travel as kernel arguments (no pinned staging, stream-ordered)
            std::vector<double> h(len * W);
            HK<E>::factor_table(table_op, (unsigned)n, (unsigned)len, nullptr, 0, h.data(), len);
            upload_small(R.stream, tab->p, h.data(), len * W);
        } else
            K<E>::factor_table(R.stream, table_op, (unsigned)n, (unsigned)len, nullptr, 0, tab->p, len);
        cache[key] = tab;
        return tab;
    }

    // ---- derivative-like slab scalings (mt:457-509) ---------------------------------------------------------------
    static P deriv_like(const P& a, size_t v, size_t n, int table_op, const char* what) {
        size_t len_of = v < a.deg.size() ? a.deg[v] : UMAX;
        if (!(v < a.deg.size() && n < len_of)) throw Error(std::string(what) + ": assertion failed: v < num_vars && n < len_of(v)");
        if (v >= a.shape.size()) return n == 0 ? a : zero_with(a.deg);
        Dims d = a.deg;
        d[v] = d[v] > n ? d[v] - n : 0;
        if (n >= a.shape[v]) return zero_with(d);
        size_t len = a.shape[v] - n;
        const bool host = tier_host(a.numel / a.shape[v] * len, a);
        std::shared_ptr<Buf> tab = cached_table(table_op, n, len, host);
        return slab_range(a, v, n, a.shape[v], d, OP_MUL_TAB, (int)v, tab->p, len, host);
    }

    // derivative(a, v, n).truncate_to_degree_p1(d) — what the evaluator does for every Derivative node
    // (generating_function.rs Derivative arm: the operand is evaluated to degree_p1 + n and cut back) — as ONE
    // gather: truncation is pure slicing, so the values are those of the two-step form bit for bit.
    static P derivative_truncated(const P& a, size_t v, size_t n, size_t d) {
        size_t len_of = v < a.deg.size() ? a.deg[v] : UMAX;
        if (!(v < a.deg.size() && n < len_of) || v >= a.shape.size() || n >= a.shape[v])
            return truncate_to_degree_p1(deriv_like(a, v, n, TAB_DERIV, "derivative"), d);  // assertion / zero paths
        Dims deg = a.deg, out = a.shape;
        deg[v] -= n;
        out[v] -= n;
        const size_t len = out[v];
        for (size_t ax = 0; ax < deg.size(); ++ax) {
            deg[ax] = std::min(deg[ax], d);
            out[ax] = std::min(out[ax], deg[ax]);
        }
        const bool host = gather_tier(a, out);
        std::shared_ptr<Buf> tab = cached_table(TAB_DERIV, n, len, host);
        Shifts shift(out.size(), 0);
        shift[v] = (long long)n;
        return gather(a, out, deg, shift, a.shape, OP_MUL_TAB, nullptr, (int)v, tab->p, len, nullptr, host);
    }

    // ---- fused continuous-Poisson observation step -------------------------------------------------------
    // a.derivative(v, 1).truncate_to_degree_p1(d) * from(c) — the body of the reference's loop for observations from a
    // Poisson with a continuous rate (gf.rs:703-706: `gf.derive(param_var, 1) * constant(lambda / k)`) — as ONE gather:
    // element for element  c * (x * ff)  in the reference's order (derivative scaling x * ff, mt:471-479, then the
    // constant on the left, mt:1041-1047).
    static P derive_scale(const P& a, size_t v, const double* c, size_t d) {
        auto generic = [&]() { return mul(derivative_truncated(a, v, 1, d), scalar(c)); };
        size_t len_of = v < a.deg.size() ? a.deg[v] : UMAX;
        if (!(v < a.deg.size() && 1 < len_of) || v >= a.shape.size() || 1 >= a.shape[v]) return generic();  // assertion / zero paths
        if (val_is_zero(c) || val_is_one(c)) return generic();  // Mul's zero / one shortcuts
        for (int i = 0; i < W; ++i)
            if (!(c[i] - c[i] == 0.0)) return generic();  // inf / NaN constant: keep the exact dispatch
        Dims deg = a.deg, out = a.shape;
        deg[v] -= 1;
        out[v] -= 1;
        const size_t len = out[v];
        for (size_t ax = 0; ax < deg.size(); ++ax) {
            deg[ax] = std::min(deg[ax], d);
            out[ax] = std::min(out[ax], deg[ax]);
        }
        if (prod(out) < 2) return generic();  // a 1-element derivative takes Mul's scalar paths
        const bool host = gather_tier(a, out);
        std::shared_ptr<Buf> tab = cached_table(TAB_DERIV, 1, len, host);
        Shifts shift(out.size(), 0);
        shift[v] = 1;
        return gather(a, out, deg, shift, a.shape, OP_MUL_TAB_LMUL_S, c, (int)v, tab->p, len, nullptr, host);
    }

    // ---- fused observation step ------------------------------------------------------------------------
    // (a.derivative(v, 1).truncate_to_degree_p1(d) * var(v, x, d)) * from(c) — the body of the reference's
    // compound-Poisson observation loop (gf.rs:684-689) — in one launch, no dispatch read-backs.
    static P observe_step(const P& a, size_t v, const double* x, const double* c, size_t d) {
        auto generic = [&]() {
            P D = truncate_to_degree_p1(deriv_like(a, v, 1, TAB_DERIV, "derivative"), d);
            P V = var_like(v, x, true, std::min<size_t>(d, 2), d > 1, Dims(v + 1, d));
            return mul(mul(D, V), scalar(c));
        };
        size_t len_of = v < a.deg.size() ? a.deg[v] : UMAX;
        if (!(v < a.deg.size() && 1 < len_of)) return generic();          // reference assertion path
        if (v >= a.shape.size() || a.shape[v] < 2 || d < 2) return generic();  // zero_with / constant var
        if (val_is_zero(c)) return generic();                             // -> zero_with(deg)
        if (tier_host(a.numel, a)) return generic();                      // host tier: the reference's own op sequence
        for (int i = 0; i < W; ++i)
            if (!(x[i] - x[i] == 0.0) || !(c[i] - c[i] == 0.0)) return generic();  // inf/NaN scalars: keep the exact dispatch
        Dims dshape = a.shape, ddeg = a.deg;
        dshape[v] -= 1;
        ddeg[v] -= 1;
        for (size_t ax = 0; ax < ddeg.size(); ++ax) {
            ddeg[ax] = std::min(ddeg[ax], d);
            dshape[ax] = std::min(dshape[ax], d);
        }
        if (prod(dshape) < 2) return generic();                            // scalar shortcuts of Mul
        Dims sh = dshape;
        sh[v] = std::min(ddeg[v], dshape[v] + 1);
        P out = make(sh, ddeg);
        std::shared_ptr<Buf> tab = cached_table(TAB_DERIV, 1, a.shape[v] - 1);
        ObserveArgs g;
        std::memset(&g, 0, sizeof(g));
        Dims ast = c_strides(a.shape);
        int nd = 0;
        g.axis = -1;
        for (size_t ax = 0; ax < sh.size(); ++ax) {
            if (sh[ax] == 1 && ax != v) continue;  // collapsed: index 0 on this axis
            if (nd >= MAXD) return generic();
            g.out.d[nd] = (unsigned)sh[ax];
            g.d_len[nd] = (unsigned)dshape[ax];
            g.a_stride[nd] = ast[ax];
            if (ax == v) g.axis = nd;
            nd++;
        }
        g.out.nd = nd;
        g.x = Scalar2{x[0], W == 2 ? x[1] : 0.0};
        g.c = Scalar2{c[0], W == 2 ? c[1] : 0.0};
        g.x_is_zero = val_is_zero(x);
        g.x_is_one = val_is_one(x);
        g.c_is_one = val_is_one(c);
        g.tab = tab->p;
        g.tab_plane = a.shape[v] - 1;
        K<E>::observe_step(R.stream, dp<E>(a), a.numel, dp<E>(out), out.numel, g);
        return out;
    }

    // ---- a whole observation chain ----------------------------------------------------------------------------
    // n observation steps, innermost first: a <- observe_step(a, v, x, cs[i], d + (n - 1 - i)) — what the evaluator
    // computes for `observe k ~ Poisson(lambda * X)` (gf.rs:678-700: the loop builds n nested derive * var * const
    // nodes, each evaluated one degree lower than the one inside it).  One launch for the chain when every step is an
    // ordinary one (k_observe_chain); otherwise, and on the host tier, the steps one by one.
    static P observe_chain(const P& a, size_t v, const double* x, const double* cs, size_t n, size_t d) {
        auto stepwise = [&]() {
            P r = a;
            for (size_t i = 0; i < n; ++i) r = observe_step(r, v, x, cs + i * W, v_deg(d, n, i));
            return r;
        };
        if (n == 0) return a;
        // (a single step takes the chain kernel too when chains are recorded: as the epilogue-carrying launch of the Add that
        // follows, or as a rider, it costs no launch of its own — k_observe_step is the one-launch form)
        if ((n == 1 && !R.lazy_observe) || tier_host(a.numel, a) || n > 4096) return stepwise();
        if (n > (size_t)OC_MAX) {  // long chains: OC_MAX steps per launch
            P r = a;
            for (size_t i = 0; i < n; i += OC_MAX) {
                const size_t m = std::min<size_t>(OC_MAX, n - i);
                r = observe_chain(r, v, x, cs + i * W, m, v_deg(d, n, i + m - 1));
            }
            return r;
        }
        for (int i = 0; i < W; ++i)
            if (!(x[i] - x[i] == 0.0)) return stepwise();
        // walk the steps on the host: shapes, degrees and the conditions under which observe_step fuses
        Dims S = a.shape, G = a.deg;
        ObserveChainArgs g;
        std::memset(&g, 0, sizeof(g));
        unsigned longest = 0;
        for (size_t i = 0; i < n; ++i) {
            const size_t di = v_deg(d, n, i);
            const double* c = cs + i * W;
            const size_t len_of = v < G.size() ? G[v] : UMAX;
            if (!(v < G.size() && 1 < len_of) || v >= S.size() || S[v] < 2 || di < 2 || val_is_zero(c)) return stepwise();
            for (int k = 0; k < W; ++k)
                if (!(c[k] - c[k] == 0.0)) return stepwise();
            Dims dshape = S, ddeg = G;
            dshape[v] -= 1;
            ddeg[v] -= 1;
            for (size_t ax = 0; ax < ddeg.size(); ++ax) {
                ddeg[ax] = std::min(ddeg[ax], di);
                dshape[ax] = std::min(dshape[ax], di);
            }
            if (prod(dshape) < 2) return stepwise();
            if (i == 0) g.len0 = (unsigned)S[v];
            longest = std::max<unsigned>(longest, (unsigned)std::min<size_t>(S[v], 0xffffffffu));
            S = dshape;
            S[v] = std::min(ddeg[v], dshape[v] + 1);
            G = ddeg;
            g.dl[i] = (unsigned)dshape[v];
            g.lo[i] = (unsigned)S[v];
            g.c[i] = Scalar2{c[0], W == 2 ? c[1] : 0.0};
            if (val_is_one(c)) g.c_one |= 1ull << i;
        }
        if (longest > K<E>::OBSERVE_LINE_MAX || prod(S) / S[v] > 0x7fffffffu) return stepwise();
        std::shared_ptr<Buf> tab = cached_table(TAB_DERIV, 1, a.shape[v] - 1);
        Dims ast = c_strides(a.shape), ost = c_strides(S), okeep;
        int nd = 0;
        g.axis = -1;
        for (size_t ax = 0; ax < S.size(); ++ax) {
            if (S[ax] == 1 && ax != v) continue;  // collapsed: index 0 on this axis
            if (nd >= MAXD) return stepwise();
            g.fs[nd] = (unsigned)S[ax];
            g.a_stride[nd] = ast[ax];
            g.o_stride[nd] = ost[ax];
            if (ax == v) g.axis = nd;
            okeep.push_back(ax);
            nd++;
        }
        g.nd = nd;
        g.nsteps = (unsigned)n;
        g.x = Scalar2{x[0], W == 2 ? x[1] : 0.0};
        g.x_is_zero = val_is_zero(x);
        g.x_is_one = val_is_one(x);
        g.tab = tab->p;
        g.tab_plane = a.shape[v] - 1;
        g.lw_pad = (longest + 8) / 8 * 8;
        const unsigned lines = (unsigned)(prod(S) / S[v]);
        // Where and when (round 5).  The chain is RECORDED, not launched: what usually follows is a scaling or two (deferred
        // stages) and the Add of the two arms, and then the observation kernel runs with that Add as its epilogue (addsub ->
        // fuse_lazy_observe) — one launch instead of two on the critical path of every `if`.  Anything else that wants the
        // values launches the plain kernel through use_buf().
        // (intervals) no exact zero in, none out: every position of every step receives a term src * factor (* x), the
        // derivative factors j + 1 and the constants c are non-zero (checked above), x is not [0,0]
        // (round 6) ... and at x = [0,0] a step is D * eps_v alone: slab 0 along v is exactly zero and nothing else is (Support).
        // Per step the derivative moves the leading zero slabs of axis v down by one, (x + eps_v) * D keeps them where x is not
        // zero (out[k] = D[k-1] + x * D[k]: non-zero iff one of the two is) and moves them up by one where it is.
        Support out_sup = support_of_poly<E>(a);
        if (g_scan_trace.on && W == 2) {
            char key[160];
            snprintf(key, sizeof key, "observe_chain input: kind=%d (buf nz %d, pend %d, from %s)", out_sup.kind, a.buf ? (int)a.buf->nz : -1, (int)(a.pend != nullptr),
                     a.buf && a.buf->origin ? a.buf->origin : "?");
            g_scan_trace.counts[key]++;
        }
        if (out_sup.exact()) {
            if (v < (size_t)Buf::ZAX) {
                out_sup.kind = 3;
                for (size_t i = 0; i < n; ++i) {
                    const unsigned zd = out_sup.z[v] > 0 ? out_sup.z[v] - 1 : 0;
                    out_sup.z[v] = val_is_zero(x) ? zd + 1 : zd;
                }
                for (size_t u = 0; u < (size_t)Buf::ZAX; ++u)
                    if (out_sup.z[u] > 0 && out_sup.z[u] >= (u < S.size() ? S[u] : 1)) out_sup.kind = 5;  // (nothing but zeros)
                out_sup.normalise();
            } else
                out_sup.kind = val_is_zero(x) ? 0 : out_sup.kind;
        }
        // (a recorded SUM as the input is launched now: only an Add could have launched it for free, and a chain whose input is
        // not in memory can neither ride along with another launch nor let the Horner loop behind it do so)
        // (in the launch graph the sum is simply this chain's predecessor)
        if (!R.batch_dag && a.buf && !a.buf->host && a.buf->lazy && a.buf->lazy->sum) use_buf(a.buf.get());
        if (R.lazy_observe && a.buf && !a.buf->host) {
            P out = make_recorded(S, G);
            out_sup.store(out.buf.get());
            auto lo = std::allocate_shared<LazyObs>(gft_small::Alloc<LazyObs>());
            lo->a = a;
            lo->tab = tab;
            lo->g = g;
            lo->lines = lines;
            lo->longest = longest;
            lo->S = S;
            lo->okeep = okeep;
            lo->out_numel = out.numel;
            auto op = std::allocate_shared<LazyOp>(gft_small::Alloc<LazyOp>());
            op->obs = lo;
            op->run = [lo](Buf* b) { launch_obs(*lo, b->p, lo->out_numel, nullptr); };
            auto rec = std::allocate_shared<ObsRec>(gft_small::Alloc<ObsRec>());
            rec->lo = lo;
            op->rec = rec;
            out.buf->lazy = op;
            return out;
        }
        P out = make(S, G);
        out_sup.store(out.buf.get());
        K<E>::observe_chain(R.stream, dp<E>(a), a.numel, dp<E>(out), out.numel, g, lines, longest);
        return out;
    }
    static int nz_of(const P& p) { return nz_of_poly<E>(p); }
    // Asks the device once whether a (plain, interval) tensor holds an exact zero and remembers the answer on its buffer.
    // Programs whose tensors do (triangular supports: hmm) would pay a round trip per tensor for nothing, so a "yes" makes
    // the next candidates go unasked (doubling back-off); their descendants inherit the 1 anyway.
    static int nz_query(const P& p) {
        static unsigned skip = 0, backoff = 0;
        if (W != 2 || !R.nz_proofs || !p.buf || p.buf->host || p.pend || p.numel < 64) return nz_of(p);
        if (p.buf->nz) return p.buf->nz;  // (answered — 1: measured on this buffer, no slab pattern)
        if (skip) {
            --skip;
            return 0;
        }
        // (round 6) where the zeros are, if there are any: whole slabs 0 of some axes are a pattern the proofs can carry (Support)
        Dims keep = collapse_mask({&p.shape}, false);
        bool axes_ok = keep.size() <= 6 && p.numel < 0xffffffffull;
        for (size_t ax : keep) axes_ok = axes_ok && ax < (size_t)Buf::ZAX;
        double has = 1.0;
        if (axes_ok) {
            Shape sh = to_shape(pick(p.shape, keep));
            Mailbox mb = next_mail();
            K<E>::zero_pattern(R.stream, dp<E>(p), p.numel, sh, p.numel, R.d_flag + 32, mb);
            double cnt[7] = {0, 0, 0, 0, 0, 0, 0};
            wait_mail(mb, cnt, 7);
            R.stats[1]++;
            has = cnt[0];
            if (g_scan_trace.on) {
                char key[200];
                std::string shs;
                for (size_t q = 0; q < p.shape.size(); ++q) shs += std::to_string(p.shape[q]) + "x";
                snprintf(key, sizeof key, "zero pattern query: %s zeros=%g per-axis slab0 zeros: %g %g %g (backoff %u)", shs.c_str(), cnt[0], cnt[1], cnt[2], cnt[3], backoff);
                g_scan_trace.counts[key]++;
            }
            if (has != 0.0) {
                Support sp;
                sp.kind = 3;
                double outside = 1.0;  // coefficients in none of the all-zero slabs 0
                for (size_t j = 0; j < keep.size(); ++j) {
                    const double ext = (double)p.shape[keep[j]], slab = (double)p.numel / ext;
                    if (cnt[1 + j] == slab && ext >= 2) {
                        sp.z[keep[j]] = 1;
                        outside *= ext - 1.0;
                    } else
                        outside *= ext;
                }
                if ((double)p.numel - outside == cnt[0]) {  // every zero lies in one of those slabs, and they hold nothing else
                    sp.normalise();
                    if (sp.kind == 3) {
                        sp.store(p.buf.get());
                        backoff = 0;
                        return p.buf->nz;
                    }
                }
            }
        } else {
            Mailbox mb = next_mail();
            K<E>::any_zero(R.stream, dp<E>(p), p.numel, p.numel, R.d_flag + 24, mb);
            wait_mail(mb, &has, 1);
            R.stats[1]++;
        }
        p.buf->nz = has != 0.0 ? 1 : 2;
        if (has != 0.0) {
            backoff = backoff ? std::min(backoff * 2, 256u) : 16;  // (round 6: 4096 -> 256 — a program whose first statements have irregular zeros settles into slabs later)
            skip = backoff;
        } else
            backoff = 0;
        return p.buf->nz;
    }
    // A recorded observation chain (observe_chain above): everything its launch needs.
    struct LazyObs {
        P a;                        // the input (keeps its buffer alive)
        std::shared_ptr<Buf> tab;   // derivative factors (the cache may drop its own reference)
        ObserveChainArgs g;
        unsigned lines = 0, longest = 0;
        Dims S, okeep;              // result shape; the result's axes the kernel indexes (its non-unit axes and v)
        size_t out_numel = 0;
        bool fused = false;         // a consumer has launched it with its Add folded in: the plain values are (so far) nobody's business
    };
    // Launches a recorded chain into `outp` (its own buffer, or a consumer's output with the epilogue `epi`) — the unbatched
    // form ("batch_dag" off; with it on a recording is issued with its level of the launch graph, gft_api_dag.inc).
    static void launch_obs(LazyObs& lo, double* outp, size_t out_numel, const ObsEpi* epi) {
        const double* ap = dp<E>(lo.a);  // (a recorded input is launched first, on its own)
        K<E>::observe_chain_epi(R.stream, ap, lo.a.numel, outp, out_numel, lo.g, lo.lines, lo.longest, epi);
    }
    // addsub(self, other) where one operand is a chain on top of a recorded observation whose result is the whole output:
    // the observation kernel runs with the other operand's chain and the Add as its epilogue (ObsEpi).  false = not this
    // case (nothing launched).
    static bool fuse_lazy_observe(const P& self, const P& other, bool subtract, const Dims& shape, const Dims& rd, P* result) {
        if (!R.lazy_observe) return false;
        auto lazy_of = [&](const P& p) -> LazyObs* {
            if (!p.buf || p.buf->host || !p.buf->lazy || !p.buf->lazy->obs) return nullptr;
            LazyObs* lo = static_cast<LazyObs*>(p.buf->lazy->obs.get());
            if (!same_dims_mod_trailing_ones(lo->S, shape) || !same_dims_mod_trailing_ones(p.shape, shape)) return nullptr;
            if (p.pend) {
                const Pend& q = *p.pend;
                if (q.padded || q.base_off != 0 || q.mat || !same_dims_mod_trailing_ones(q.base_shape, lo->S)) return nullptr;
            }
            return lo;
        };
        // prefer the RIGHT operand (when both are recorded, the left one is launched plain by chain_src below — or has been,
        // on a side stream)
        LazyObs* lo = lazy_of(other);
        const bool x_is_other = lo != nullptr;
        if (!lo) lo = lazy_of(self);
        if (!lo) {
            return false;
        }
        const P& X = x_is_other ? other : self;
        const P& Y = x_is_other ? self : other;
        if (Y.buf && Y.buf.get() == X.buf.get()) return false;  // (both read the same recording: launch it)
        // every axis the other operand's chain indexes must be one of the kernel's axes
        Dims ckeep = chain_keep(shape, {&self, &other});
        auto pos = [&](size_t ax) -> int {
            for (size_t j = 0; j < lo->okeep.size(); ++j)
                if (lo->okeep[j] == ax) return (int)j;
            return -1;
        };
        for (size_t ax : ckeep)
            if (pos(ax) < 0) return false;
        ObsEpi e;
        std::memset(&e, 0, sizeof(e));
        e.mode = x_is_other ? 1 : 2;
        e.subtract = subtract ? 1 : 0;
        if (X.pend) {
            e.npost = X.pend->n;
            for (int i = 0; i < X.pend->n; ++i) {
                const PendStage& s = X.pend->st[i];
                e.post[i].kind = s.kind;
                e.post[i].s = Scalar2{s.s[0], s.s[1]};
                e.post[i].axis = 0;
                if (s.kind == CH_MUL_TAB) {
                    const int j = pos((size_t)s.axis);
                    if (j < 0) return false;
                    e.post[i].axis = j;
                    use_buf(s.tab->dev.get());
                    e.post[i].tab = s.tab->dev->p;
                    e.post[i].tab_plane = s.tab->len;
                }
            }
        }
        if (R.batch_dag) {  // a node of the launch graph: the chain with this Add as its epilogue, issued with its level
            auto rec = std::allocate_shared<ObsAddRec>(gft_small::Alloc<ObsAddRec>());
            rec->xop = X.buf->lazy;
            rec->lo = lo;
            rec->e = e;
            rec->Y = Y;
            P out = make_recorded(shape, rd);
            sum_nz_store(self, other, shape, out.buf.get());
            rec->out_numel = out.numel;
            lo->fused = true;
            out.buf->lazy = op_of(rec);
            *result = out;
            return true;
        }
        // From here on the recording is spoken for: bringing Y into memory may launch OTHER recorded chains (Y's own base, with
        // a rider) and must not pick this one as its rider — its LazyOp would be released under our feet.
        std::shared_ptr<LazyOp> keep = X.buf->lazy;
        lo->fused = true;
        if (!Y.pend) (void)dp<E>(Y);  // a lazy handle / host-tier tensor gets its device buffer
        e.y = chain_src_dev(Y, lo->okeep);
        P out = make(shape, rd);
        sum_nz_store(self, other, shape, out.buf.get());
        launch_obs(*lo, dp<E>(out), out.numel, &e);
        R.stats_side[3]++;
        R.stats_ex[2]++;
        *result = out;
        return true;
    }
    static size_t v_deg(size_t d, size_t n, size_t i) { return d + (n - 1 - i); }  // truncation degree of step i of n

    // ---- shift_down (mt:514-536) -------------------------------------------------------------------------------------
    static void sum_axis_into(const P& a, size_t v, size_t upto, double* out, size_t out_plane, bool host) {
        size_t outer = 1, inner = 1;
        for (size_t i = 0; i < v; ++i) outer *= a.shape[i];
        for (size_t i = v + 1; i < a.shape.size(); ++i) inner *= a.shape[i];
        int mode = SUM_SEQ;
        // ndarray 0.15.6 sum_axis: 2-d array whose summed axis has unit stride => per-lane 8-way fold
        if (a.shape.size() == 2 && inner == 1) mode = SUM_UNROLL8;
        if (host) {
            HK<E>::sum_axis(hp<E>(a), a.numel, (unsigned)outer, (unsigned)upto, (unsigned)inner, a.shape[v] * inner, out,
                            out_plane, mode);
            return;
        }
        if (W == 1 && inner == 1 && upto >= 128) mode = SUM_WAVE;  // long rows: wavefront-shuffle reduction
        K<E>::sum_axis(R.stream, dp<E>(a), a.numel, (unsigned)outer, (unsigned)upto, (unsigned)inner,
                       a.shape[v] * inner, out, out_plane, mode);
    }
    static P shift_down(const P& a, size_t v, size_t n) {
        size_t len_of = v < a.deg.size() ? a.deg[v] : UMAX;
        if (!(v < a.deg.size() && n < len_of)) throw Error("shift_down: assertion failed: v < num_vars && n < len_of(v)");
        if (v >= a.shape.size()) return a;
        Dims d = a.deg;
        d[v] = d[v] > n ? d[v] - n : 0;
        const bool host = tier_host(a.numel, a);
        if (a.shape[v] <= n + 1) {
            Dims rs = a.shape;
            rs[v] = 1;
            P out = make(rs, d, host);
            sum_axis_into(a, v, a.shape[v], tp<E>(out, host), out.numel, host);
            return seal(out);
        }
        Dims os = a.shape;
        os[v] = a.shape[v] - n;
        Shifts sh0(os.size(), 0);
        sh0[v] = (long long)n;
        // (a real copy, not a deferred view: slab 0 of the result is updated in place below, so a view would only be
        // materialised at once — by the chain kernel, which indexes per element and was 30 % slower than the gather's
        // 16-byte path at 384^3: profiles/r03 vs r02 streaming_384.json)
        struct NoDefer {
            bool prev;
            NoDefer() : prev(R.defer) { R.defer = false; }
            ~NoDefer() { R.defer = prev; }
        };
        P out;
        {
            NoDefer nd_;
            out = gather(a, os, d, sh0, a.shape, OP_COPY, nullptr, -1, nullptr, 0, nullptr, host);
        }
        Dims ss = a.shape;
        ss[v] = 1;
        std::shared_ptr<Buf> s = alloc_tier(host, prod(ss) * W);
        sum_axis_into(a, v, n, s->p, prod(ss), host);
        HV vo = view(out, host), vs{s->p, prod(ss), ss, host};
        x_block_op(vo, vs, BLK_ADD, 0);
        return seal(out);
    }


    // ---- subst_var (mt:540-580) -----------------------------------------------------------------------------------------
    static P subst_var(const P& a, size_t v, const P& subst) {
        if (v >= a.shape.size()) return a;
        Dims deg = min_degrees(a, subst);
        if (is_zero(subst)) return slab_range(a, v, 0, 1, deg);
        double c[2], m[2];
        size_t w = 0;
        ScanCtx sc_subst("subst_var.subst");
        // A 2-element tensor is linear by structure; if the host also knows its constant term (see gft_poly::c0_known)
        // the verdict needs no device scan.  m stays on the device (the power table reads it from there).
        bool have_lin = false, m_known = true;
        if (subst.buf && !subst.buf->host && !subst.buf->lin_state && subst.numel == 2 && subst.c0_known && val_is_zero(subst.c0)) {
            for (size_t ax = 0; ax < subst.shape.size(); ++ax)
                if (subst.shape[ax] == 2) w = ax;
            if (v == w) {
                c[0] = subst.c0[0];
                c[1] = subst.c0[1];
                have_lin = true;
                m_known = false;
            }
        }
        if (!have_lin) have_lin = extract_linear(subst, c, m, &w);
        if (have_lin) {
            if (v == w && val_is_zero(c)) {
                Dims lens = a.shape;
                for (size_t i = 0; i < lens.size(); ++i) lens[i] = std::min(lens[i], deg[i]);
                if (m_known && val_is_one(m)) return lead_block(a, lens, deg);  // powers of one: x * 1 == x, nothing to compute
                Dims sst = c_strides(subst.shape);
                Shifts shift(lens.size(), 0);
                if (m_known && gather_tier(a, lens)) {  // host tier: the running product m^k per element (mt:557-565)
                    const double mm[2] = {m[0], m[1]};
                    return gather(a, lens, deg, shift, a.shape, OP_MUL_POW, nullptr, (int)v, mm, 1, nullptr, 1);
                }
                if (m_known && can_defer(a, prod(lens)) && lens.size() <= a.shape.size()) {
                    // deferred: x * m^k along v as a chain stage reading the per-m device table (no launch here)
                    P r = deferred(a, lens, deg, 1);
                    push_stage(r, CH_MUL_TAB, nullptr, (int)v, pow_table(m, lens[v]));
                    return r;
                }
                if (m_known) {
                    // m is known on the host: the powers m^k — the reference's running product ((1*m)*m)*.. (mt:557-565), the
                    // same functor on the host, so the same bits — are formed here.  Up to HTAB_CAP of them travel BY VALUE
                    // with the gather (no table launch, no per-thread running product, no device mirror of a host-resident
                    // subst); longer tables are uploaded.
                    static const bool htab_on = true;
                    if (htab_on) {
                        std::vector<double> pw(lens[v] * W);
                        typename E::V f = E::one();
                        const typename E::V mv = E::from(Scalar2{m[0], W == 2 ? m[1] : 0.0});
                        for (size_t k = 0; k < lens[v]; ++k) {
                            E::st(pw.data(), lens[v], k, f);
                            f = E::mul(f, mv);
                        }
                        if (lens[v] <= (W == 1 ? HTAB_CAP : HTAB_CAP / 2))
                            return gather(a, lens, deg, shift, a.shape, OP_MUL_HTAB, nullptr, (int)v, pw.data(), lens[v], nullptr, 0);
                        std::shared_ptr<Buf> tabh = alloc_doubles(lens[v] * W);
                        upload_small(R.stream, tabh->p, pw.data(), lens[v] * W);
                        return gather(a, lens, deg, shift, a.shape, OP_MUL_TAB, nullptr, (int)v, tabh->p, lens[v], nullptr, 0);
                    }
                    if (lens[v] <= 256) {
                        const double mm[2] = {m[0], m[1]};
                        return gather(a, lens, deg, shift, a.shape, OP_MUL_POW, mm, (int)v, nullptr, 0, nullptr, 0);
                    }
                }
                if (lens[v] <= 256)  // short axis: every thread forms its own m^k (same running product), no table launch
                    return gather(a, lens, deg, shift, a.shape, OP_MUL_POW, nullptr, (int)v, dp<E>(subst) + sst[w], subst.numel, nullptr, 0);
                std::shared_ptr<Buf> tab = alloc_doubles(lens[v] * W);
                K<E>::factor_table(R.stream, TAB_POW, 0, (unsigned)lens[v], dp<E>(subst) + sst[w], subst.numel, tab->p, lens[v]);
                return gather(a, lens, deg, shift, a.shape, OP_MUL_TAB, nullptr, (int)v, tab->p, lens[v], nullptr, 0);
            }
        }
        Dims cshape = a.shape;
        while (cshape.size() < deg.size()) cshape.push_back(1);
        P ca = with_meta_unchecked(a, cshape);
        // linear substitution known from the scan above (memoised on subst's buffer): fused Horner steps
        const bool lin_known = extract_linear(subst, c, m, &w) && w < deg.size() && deg[w] >= 2;
        P res;
        if (R.fuse_horner && cshape[v] <= WIT_SLOTS && horner_speculative(ca, v, subst, deg, lin_known, c, m, w, &res)) return res;
        return horner_exact(ca, v, subst, deg);
    }
    // One Horner coefficient: a[.., i, ..] clipped to deg (mt:571-576)
    static P horner_coeff(const P& ca, size_t v, size_t i, const Dims& deg) {
        Dims out = ca.shape;
        out[v] = 1;
        for (size_t ax = 0; ax < out.size(); ++ax) out[ax] = std::min(out[ax], deg[ax]);
        Shifts shift(out.size(), 0);
        shift[v] = (long long)i;
        return gather(ca, out, deg, shift, ca.shape);
    }
    // The reference's loop, operation for operation (mt:569-579): every product goes through Mul's dispatcher, which asks
    // on every step whether the accumulator is linear (a device scan + host round trip for device tensors, free on the
    // host tier).
    static P horner_exact(const P& ca, size_t v, const P& subst, const Dims& deg) {
        ScanCtx sc_he("subst_var.horner_exact");
        P res = zero_with(deg);
        for (size_t i = ca.shape[v]; i-- > 0;) res = addsub(mul(res, subst), horner_coeff(ca, v, i, deg), false);
        return res;
    }
    // The same loop without a host round trip per step.  The reference asks on every step whether the ACCUMULATOR is
    // linear (and then multiplies the other way round, which compacts its stored shape).  That is the case while the top
    // coefficient slabs are zero or scalar-like; once the accumulator has been seen non-linear it stays so short of an
    // exact cancellation.  So: reference steps (with the scan) until the first "not linear" verdict, then fused /
    // scan-free steps that SPECULATE "still not linear" — and prove it: every accumulator produced on the way is
    // checked on the device for a witness of non-linearity (a non-zero coefficient at an index with two non-zero
    // coordinates or a coordinate >= 2), one sticky word per step, and ONE verdict read-back at the end of the loop.
    // A step without a witness (exact cancellation, or a support on unit positions of two axes) makes the function
    // return false and the caller redoes the loop with horner_exact: the stored shapes are the reference's in every
    // case, the extra cost is one round trip per subst_var instead of one per step.
    // a device tensor copied into a host-tier tensor (the caller has decided that the host is the better machine for it)
    static P to_host_tier(const P& dev) {
        P h = make(dev.shape, dev.deg, true);
        HIP_OK(hipMemcpyAsync(hp<E>(h), dp<E>(dev), sizeof(double) * dev.numel * W, hipMemcpyDeviceToHost, R.stream));
        HIP_OK(hipStreamSynchronize(R.stream));
        return h;
    }
    static constexpr size_t WIT_SLOTS = 8192;
    static bool horner_speculative(const P& ca, size_t v, const P& subst, const Dims& deg, bool lin_known, const double c[2],
                                   const double m[2], size_t w, P* result) {
        ScanCtx sc_hs("subst_var.horner_speculative");
        P res = zero_with(deg);
        bool res_nonlinear_seen = false;
        unsigned slots = 0;  // speculated accumulators so far (sticky witness words R.d_wit[0 .. slots))
        // Interval tensors with a substitution whose constant term c is a non-zero finite interval: interval arithmetic
        // cannot cancel — a product of two non-zero intervals and a sum with a non-zero term are widened, never [0,0]
        // (interval.rs:126-190) — and every step's out[k] contains the term c * res[k], so a non-zero coefficient of the
        // accumulator stays non-zero at its position for the rest of the loop.  The witness the initial scan found
        // therefore persists: the speculation is PROVEN and needs no per-step witnesses and no verdict read-back.
        bool proven = false;
        if (W == 2) {
            double c0[2];
            bool have = false;
            if (lin_known) { c0[0] = c[0]; c0[1] = c[1]; have = true; }
            else if (on_host(subst) || subst.c0_known || (subst.numel == 1 && subst.cached)) { first_value(subst, c0); have = true; }
            proven = have && !val_is_zero(c0) && (c0[0] - c0[0] == 0.0) && (c0[1] - c0[1] == 0.0);
        }
#ifdef GFT_DIAG  // measurement-only builds (make EXTRA=-DGFT_DIAG): skipping the verdict makes the speculation unsound
        static const bool verify = [] {
            const char* e = getenv("GFT_HORNER_VERIFY");
            return e ? atoi(e) != 0 : true;
        }();
        if (!verify) proven = true;
#endif
        // Host phase for LINEAR streaks (round 4).  The top coefficient slabs of a tensor with triangular support hold a
        // coefficient or two each: the accumulator stays linear for several steps, and on the device every one of them is a
        // scan, a host round trip and four launches of a few elements (hmm `--bounds`: 11 918 of its 16 300 scans said
        // "linear").  After the first such verdict the accumulator and the next HBLK coefficient slabs are brought to the
        // host tier (one gather, two copies) and the steps run there — the same reference steps on the same functors, their
        // scans free — until the accumulator is no longer linear or outgrows the tier; then the device takes over again.
        constexpr size_t HBLK = 8;
        P ca_h;                       // host-tier copy of the slabs [ca_h_lo, ca_h_hi) of ca along v
        size_t ca_h_lo = 0, ca_h_hi = 0;
        static const bool host_phase_on = true;
        auto fetch_block = [&](size_t i_top) {
            const size_t lo = i_top + 1 > HBLK ? i_top + 1 - HBLK : 0;
            ca_h = to_host_tier(slab_range(ca, v, lo, i_top + 1, ca.deg, OP_COPY, -1, nullptr, 0, 0));
            ca_h_lo = lo;
            ca_h_hi = i_top + 1;
        };
        // (round 6) a proven loop on a tensor nothing is known about: ask once where its zeros are (nz_query: one launch and round
        // trip, what a single linearity scan costs — and the answer is inherited by everything computed from the tensor)
        if (W == 2 && proven && R.nz_proofs && !on_host(ca) && ca.numel >= 64 && ca.buf->nz == 0) {
            if (!ca.pend) (void)nz_query(ca);
            else if (!ca.pend->mat) {  // a chain: the question goes to its base tensor, the chain's stages carry the answer over
                P base;
                base.width = W;
                base.shape = ca.pend->base_shape;
                base.deg = ca.pend->base_shape;
                base.numel = ca.pend->base_numel;
                base.buf = ca.buf;
                if (base.numel >= 64 && prod(base.shape) == base.numel) (void)nz_query(base);
            }
        }
        for (size_t i = ca.shape[v]; i-- > 0;) {
            bool speculate = false;
            if (!on_host(res) && res.numel > 1) {
                // (round 5, intervals) a proven loop whose coefficient tensor holds no exact zero: the accumulator — a slab of it,
                // at least 3 coefficients — cannot be linear (nz_of_poly): no scan, no round trip; the verdict is memoised as
                // if the scan had delivered it
                if (!res_nonlinear_seen && W == 2 && proven && R.nz_proofs && res.numel >= 3 && res.buf &&
                    (nz_of(res) == 2 || (res.buf.get() == ca.buf.get() && nz_of(ca) == 0 && nz_query(ca) == 2 && nz_of(res) == 2) ||
                     support_proves_nonlinear(support_of_poly<E>(res), res.shape))) {
                    // (the loop reads the slab in place where it is a pure view — res_is_view —, from memory otherwise: dp() in
                    // launch_horner materialises it then)
                    if (!res.pend && res.buf && !res.buf->host && !res.buf->lin_state) res.buf->lin_state = 1;
                    res_nonlinear_seen = true;
                    R.stats_nz++;
                }
                if (!res_nonlinear_seen) {  // device accumulator: scan until the first "not linear" verdict
                    double c_[2], m_[2];
                    size_t u_;
                    ScanCtx sc_res("subst_var.accumulator");
                    const Support sr_pre = g_scan_trace.on ? support_of_poly<E>(res) : Support();
                    const int res_nz_pre = res.buf ? (int)res.buf->nz : -1, res_pend_pre = res.pend ? res.pend->n + 100 * (res.pend->base_off != 0) : -1;
                    const bool res_same_pre = res.buf.get() == ca.buf.get();
                    const bool res_lazy_pre = res.buf && res.buf->lazy != nullptr;
                    ScanToken tok = extract_linear_begin(res);
                    if (g_scan_trace.on && W == 2) {  // why no proof: what is known about the operand's and the accumulator's zeros
                        char key[200];
                        const Support sc = support_of_poly<E>(ca), sr = support_of_poly<E>(res);
                        std::string st;
                        if (ca.pend) {
                            st = ca.pend->padded ? "pad " : "";
                            for (int q = 0; q < ca.pend->n; ++q) st += std::to_string(ca.pend->st[q].kind) + ",";
                            if (ca.pend->base_off) st += "off";
                        }
                        (void)sr;
                        snprintf(key, sizeof key, "no proof: ca buf.nz=%d kind=%d pend=[%s] from %s | res BEFORE the scan: kind=%d buf.nz=%d pend=%d same_buf=%d lazy=%d numel=%s res_seen=%d",
                                 ca.buf ? (int)ca.buf->nz : -1, sc.kind, st.c_str(), ca.buf && ca.buf->origin ? ca.buf->origin : "?", sr_pre.kind, res_nz_pre, res_pend_pre, (int)res_same_pre,
                                 (int)res_lazy_pre, res.numel < 64 ? "<64" : ">=64", (int)res_nonlinear_seen);
                        g_scan_trace.counts[key]++;
                    }
                    if (g_scan_trace.on) {
                        char key[160];
                        snprintf(key, sizeof key, "horner scan: done=%d lin_known=%d rank_ok=%d i=%s proven=%d len_v=%s", (int)tok.done, (int)lin_known,
                                 (int)(res.shape.size() == deg.size()), i == 0 ? "0" : (i == 1 ? "1" : ">1"), (int)proven, ca.shape[v] == 2 ? "2" : ">2");
                        g_scan_trace.counts[key]++;
                    }
                    // The scan is in flight (or answered).  With a linear substitution the steps that follow a "not linear"
                    // verdict are ONE launch (horner_linear_rest): queue it right behind the scan, guarded on the device by
                    // the scan's verdict word, and only then wait for the verdict — the loop runs while the host reads its
                    // mail instead of after a launch latency on top of the round trip (`--bounds` programs: one scan per
                    // subst_var, 16-18 thousand of them).  A "linear" verdict (rare: the reference then multiplies the other
                    // way round) makes the queued launch a no-op; its output is dropped and the reference step is taken.
                    P ahead;
                    bool queued = false;
                    unsigned* ahead_wit = (proven || slots + (unsigned)i > WIT_SLOTS) ? nullptr : R.d_wit + slots;
                    static const bool ahead_on = true;
                    bool queued_step = false;
                    if (ahead_on && !tok.done && R.fuse_horner && lin_known && res.shape.size() == deg.size()) {
                        if (i >= 1 && (proven || ahead_wit)) {
                            if (slots == 0 && !proven) HIP_OK(hipMemsetAsync(R.d_wit, 0, sizeof(unsigned) * WIT_SLOTS, R.stream));
                            // a proven loop on an operand the main chain has long passed: RECORDED (its launch needs no guard —
                            // nobody looks at the handle before the verdict below is in — and is dropped if the verdict is "linear")
                            queued = horner_linear_rest(res, ca, v, i, c, m, w, deg, &ahead, ahead_wit, R.d_flag + 16, proven && R.lazy_horner && R.batch_dag);
                        }
                        // ... or the single fused step where the whole-loop launch does not apply: the last step (nothing is
                        // speculated about its result), or any step of a proven loop (no witness to raise)
                        if (!queued && (i == 0 || proven)) {
                            ahead = horner_linear_step(res, ca, v, i, c, m, w, deg, R.d_flag + 16);
                            queued_step = true;
                        }
                    }
                    if (!extract_linear_end(tok, c_, m_, &u_)) res_nonlinear_seen = true;  // memoised: the generic mul reuses it
                    if (g_scan_trace.on) g_scan_trace.counts[res_nonlinear_seen ? "horner scan verdict: not linear" : "horner scan verdict: LINEAR"]++;
                    if (queued && res_nonlinear_seen) {
                        res = ahead;
                        if (!proven) slots += (unsigned)i;
                        break;
                    }
                    if (queued_step && res_nonlinear_seen) {
                        res = ahead;
                        if (i == 0) break;
                        continue;
                    }
                    // (queued and linear: `ahead` is dropped — its launch returned at the guard)
                    if (!res_nonlinear_seen && host_phase_on && R.host_max_elems && res.numel <= R.host_max_elems && !on_host(ca) &&
                        prod(ca.shape) / ca.shape[v] * HBLK <= 16 * R.host_max_elems) {
                        res = to_host_tier(res);
                        fetch_block(i);
                        ++i;  // this step is taken again, on the host tier
                        continue;
                    }
                }
                speculate = res_nonlinear_seen;
            }
            if (!speculate) {  // reference step; on the host tier the accumulator's scan is free
                // Host-resident accumulator that is not linear, linear substitution: the step as ONE pass of the host tier
                // (gft_host.hpp horner_linear: the element order of mul -> mul_linear -> add -> add, no intermediate
                // tensors) — `--bounds` programs live on this path because subst - constant_term(subst) keeps a widened
                // constant.  Linearity is re-checked on every step (a host scan), so the stored shapes are the reference's.
                // (a host accumulator of a DEVICE tensor: the host phase above — its slabs come from the fetched block)
                const bool in_block = on_host(res) && !on_host(ca) && ca_h_hi > ca_h_lo;
                if (in_block && i < ca_h_lo) fetch_block(i);  // the streak outlasted the block
                if (in_block) {
                    double c_[2], m_[2];
                    size_t u_;
                    // (a PROVEN loop's device steps need not ask again; an unproven one scans the accumulator that arrives on the
                    // device — this verdict is about the accumulator BEFORE the step, and only witnesses cover results of steps)
                    if (res.numel > 1 && !extract_linear(res, c_, m_, &u_) && proven) res_nonlinear_seen = true;
                    res = addsub(mul(res, subst), horner_coeff(ca_h, v, i - ca_h_lo, deg), false);
                    continue;
                }
                if (lin_known && on_host(res) && on_host(ca) && res.numel > 1 && res.shape.size() == deg.size()) {
                    double c_[2], m_[2];
                    size_t u_;
                    if (!extract_linear(res, c_, m_, &u_)) {
                        // (the host scan's verdict carries over when the accumulator outgrows the host tier ONLY for a proven loop
                        // — it stays non-linear.  Otherwise the accumulator this step hands to the device is scanned there once:
                        // the witnesses of the speculative steps cover their own results, not the one they start from)
                        if (proven) res_nonlinear_seen = true;
                        // proven (interval, non-zero finite constant): the accumulator stays non-linear for the rest of the
                        // loop, so every remaining step runs in one host loop — two ping-pong buffers, no per-step scan,
                        // handle or shape vectors (they cost as much as the arithmetic on ~100-element tensors)
                        if (proven && horner_linear_rest_host(res, ca, v, i, c, m, w, deg, &res)) break;
                        res = horner_linear_step(res, ca, v, i, c, m, w, deg);
                        continue;
                    }
                }
                if (g_scan_trace.on && W == 2 && !on_host(ca)) {
                    P hc = horner_coeff(ca, v, i, deg);
                    P pr = mul(res, subst);
                    P nx = addsub(pr, hc, false);
                    char key[200];
                    snprintf(key, sizeof key, "reference step: coeff pend=%d off=%d same=%d | prod numel=%zu cached=%d host=%d | next pend=%d same=%d host=%d numel=%s", hc.pend ? hc.pend->n : -1,
                             hc.pend ? (int)(hc.pend->base_off != 0) : -1, (int)(hc.buf.get() == ca.buf.get()), pr.numel, (int)pr.cached, (int)on_host(pr), nx.pend ? nx.pend->n : -1,
                             (int)(nx.buf.get() == ca.buf.get()), (int)on_host(nx), nx.numel < 64 ? "<64" : ">=64");
                    g_scan_trace.counts[key]++;
                    res = nx;
                    continue;
                }
                res = addsub(mul(res, subst), horner_coeff(ca, v, i, deg), false);
                continue;
            }
            if (slots == 0 && !proven) HIP_OK(hipMemsetAsync(R.d_wit, 0, sizeof(unsigned) * WIT_SLOTS, R.stream));
            bool witnessed = false;
            if (lin_known && res.shape.size() == deg.size()) {
                // every remaining step in one launch (one workgroup per line along w); witnesses are raised in the kernel
                if (i >= 1 && horner_linear_rest(res, ca, v, i, c, m, w, deg, &res, proven ? nullptr : R.d_wit + slots, nullptr, proven && R.lazy_horner && R.batch_dag)) {
                    if (!proven) slots += (unsigned)i;  // the accumulators after the in-kernel steps 0 .. i-1 (the last one is the result)
                    break;
                }
                res = horner_linear_step(res, ca, v, i, c, m, w, deg);
            } else {
                P nxt;
                static const int shallow_diag = 0;
                const bool fuse_wit = !(shallow_diag & 1);
                if (horner_general_step_fused(res, subst, ca, v, i, deg, (fuse_wit && i > 0 && !proven) ? R.d_wit + slots : nullptr, &nxt)) {
                    res = nxt;
                    witnessed = fuse_wit;  // (the step's kernel raised the witness word itself)
                } else {
                    res = addsub(mul_horner(res, subst), horner_coeff(ca, v, i, deg), false);
                }
            }
            if (on_host(res) || res.numel == 1) return false;  // left the speculated regime: take the exact loop
            if (i == 0) break;  // the last accumulator is the result: nothing is speculated about it
            if (!proven) {
                if (!witnessed) {
                    Dims keep = collapse_mask({&res.shape}, false);
                    HV rv = view(res);
                    K<E>::witness(R.stream, dview(rv, &keep), R.d_wit + slots);
                }
                slots++;
            }
        }
        if (slots) {
            Mailbox mb = next_mail();
            witness_verdict(R.stream, R.d_wit, slots, mb);
            R.stats[0]++;
            double missing = 0.0;
            wait_mail(mb, &missing, 1);
            if (missing != 0.0) return false;
        }
        if (g_scan_trace.on && W == 2 && res.buf && !res.buf->host) {
            char key[160];
            const Support sr = support_of_poly<E>(res), sc = support_of_poly<E>(ca);
            snprintf(key, sizeof key, "subst_var (speculative) result: kind=%d from ca kind=%d lin_known=%d proven=%d", sr.kind, sc.kind, (int)lin_known, (int)proven);
            g_scan_trace.counts[key]++;
        }
        *result = res;
        return true;
    }
    // res * subst inside the Horner loop once the accumulator has been seen non-linear.  The generic mul would ask
    // again on every step whether `res` is linear (device scan + host round trip, because res is new each time)
    // before it looks at `subst`, whose verdict is memoised; here only subst's verdict is consulted, and the
    // speculation "res is not linear" is verified by horner_speculative's witnesses.
    static P mul_horner(const P& res, const P& subst) {
        if (res.numel == 1 || subst.numel == 1) return mul(res, subst);
        P self = res, other = subst;
        Dims deg = min_degrees(self, other);
        broadcast(self, other);
        Dims shape = sum_shape(self, other);
        self = truncate_degrees(self, deg);
        other = truncate_degrees(other, deg);
        if (self.numel == 1 || other.numel == 1) return mul(res, subst);
        double c[2], m[2];
        size_t v;
        if (extract_linear(other, c, m, &v)) {
            Dims sh = self.shape;
            sh[v] = std::min(deg[v], sh[v] + 1);
            return mul_linear(self, c, m, v, sh, deg);
        }
        P out = make(shape, deg);
        conv(view(self), view(other), view(out), 0, shape.empty() ? 1 : shape[0], false, false, 0, 0, 0);
        return out;
    }
    // One GENERAL Horner step  res * subst + a[.., i, ..]  (mt:569-579) in ONE launch when the product is shallow (`subst`
    // a stencil of a few coefficients — the compound-distribution substitutions of Genfer's programs): the product's
    // reference-order sums, then Add's operations on each finished sum ((0 + prod) + slab, or element 0 += slab for a
    // 1-element slab), then the witness of non-linearity the speculative loop needs — instead of product (+ prep / reduce /
    // guard launches on the tiled kernel) + slab gather + add + witness.  Operation for operation what
    // addsub(mul_horner(res, subst), horner_coeff(ca, v, i, deg)) computes (the product on the reference-order kernel), so
    // the result carries the reference's bits.  false: not this case — nothing launched, the caller takes that sequence.
    static bool horner_general_step_fused(const P& res, const P& subst, const P& ca, size_t v, size_t i, const Dims& deg,
                                          unsigned* wit, P* result) {
        if (!R.shallow_max_terms || R.conv_mode != 0) return false;
        if (res.numel == 1 || subst.numel == 1 || on_host(res)) return false;
        P self = res, other = subst;
        Dims mdeg = min_degrees(self, other);
        broadcast(self, other);
        Dims shape = sum_shape(self, other);
        const size_t nd = shape.size();
        if (nd != deg.size() || ca.shape.size() != nd) return false;
        for (size_t ax = 0; ax < nd; ++ax)  // nothing to truncate (the loop's accumulators never exceed deg)
            if (self.shape[ax] > mdeg[ax] || other.shape[ax] > mdeg[ax]) return false;
        size_t terms = 1;
        for (size_t ax = 0; ax < nd; ++ax) terms *= std::min(self.shape[ax], other.shape[ax]);
        if (terms > R.shallow_max_terms) return false;
        {
            double c[2], m[2];
            size_t u;
            if (extract_linear(other, c, m, &u)) return false;  // (memoised on subst's buffer) the mul_linear path is the reference's
        }
        // the Add: degrees, the slab's box, the result's shape (mt:854-882)
        Dims rd(nd, UMAX), oc = ca.shape;
        oc[v] = 1;
        for (size_t ax = 0; ax < nd; ++ax) {
            rd[ax] = std::min(mdeg[ax], deg[ax]);
            oc[ax] = std::min(oc[ax], deg[ax]);
            if (shape[ax] > rd[ax] || oc[ax] > rd[ax]) return false;
        }
        const bool slab_scalar = prod(oc) == 1;
        Dims oshape = shape;
        if (!slab_scalar)
            for (size_t ax = 0; ax < nd; ++ax) oshape[ax] = std::max(shape[ax], oc[ax]);
        // the kernel's loop nest is the reference's only if the product's own non-unit axes are the output's
        for (size_t ax = 0; ax < nd; ++ax)
            if ((oshape[ax] > 1) != (shape[ax] > 1)) return false;
        Dims keep = collapse_mask({&shape}, false);
        if (keep.empty() || keep.size() > 6) return false;
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        ConvEpi e;
        std::memset(&e, 0, sizeof(e));
        a.nd = (int)keep.size();
        Dims xs = pick(self.shape, keep), ys = pick(other.shape, keep), zs = pick(shape, keep);
        Dims xst = c_strides(xs), yst = c_strides(ys), zst = c_strides(zs), cst = c_strides(ca.shape);
        for (int j = 0; j < a.nd; ++j) {
            a.xs[j] = (unsigned)xs[j];
            a.ys[j] = (unsigned)ys[j];
            a.zs[j] = (unsigned)zs[j];
            a.xstr[j] = xst[j];
            a.ystr[j] = yst[j];
            a.zstr[j] = zst[j];
            e.os[j] = (unsigned)oshape[keep[j]];
            e.abox[j] = (unsigned)oc[keep[j]];
            e.astr[j] = keep[j] == v ? 0 : cst[keep[j]];
        }
        a.slab_lo = 0;
        a.slab_hi = a.zs[0];
        a.inner_from_zero = 1;  // every kept axis is non-unit: the last one is the reference's 1-d base case (mt:992-1000)
        a.variant = R.conv_variant;
        e.mode = slab_scalar ? 2 : 1;
        const double* cap = dp<E>(ca);  // (a deferred / host-tier coefficient tensor is materialised once for the whole loop)
        e.ap = cap + i * cst[v];
        e.aplane = ca.numel;
        e.wit = wit;
        const double* xp_ = dp<E>(self);
        const double* yp_ = dp<E>(other);
        P out = make(oshape, rd);
        if (!K<E>::conv_shallow(R.stream, xp_, self.numel, yp_, other.numel, out.buf->p, out.numel, a, e)) return false;
        R.stats[5]++;
        R.stats_shallow[0]++;
        R.stats_shallow[1]++;
        *result = out;
        return true;
    }
    // res * (c + m*eps_w) + a[.., i, ..] in one launch (k_horner_linear), element for element the sequence
    // mul -> mul_linear -> mul_var / scale / add -> add that the generic loop above performs.  The generic mul
    // would first ask whether `res` itself is linear (a device scan + host round trip per step) and, if so,
    // multiply the other way round; these kernels speculate that it is not and raise a witness per step
    // (horner_speculative), so a step where the speculation fails is redone by the exact loop.
    // Steps i, i-1, .., 0 in one launch (k_horner_linear_loop) when the final tensor is small enough for a single
    // workgroup to be the faster machine (a launch per step costs ~4 us of host time + ~4 us on the device).
    // The first accumulator of a Horner loop is the top coefficient slab: a sub-box VIEW of the coefficient tensor with the
    // stage "element 0 + [0,0]" from the reference's first step (0 * subst + slab).  For intervals that stage returns its
    // operand unchanged unless the operand is itself zero (iv:126-134), so where the tensor is proven free of exact zeros the
    // view IS the sub-box, bit for bit, and the loop kernel reads it in place (base pointer + the tensor's strides) instead of
    // from a materialised copy — one launch less per subst_var.
    static bool res_is_view(const P& res, const P& ca) {
        if (W != 2 || !R.nz_proofs || !res.pend || !res.buf || res.buf.get() != ca.buf.get() || ca.pend || res.buf->host) return false;
        const Pend& q = *res.pend;
        if (q.padded || q.mat || q.base_shape.size() != res.shape.size() || !(q.base_shape == ca.shape)) return false;
        for (int i = 0; i < q.n; ++i) {
            const PendStage& g = q.st[i];
            if (!((g.kind == CH_FIRST_ADD || g.kind == CH_FIRST_SUB) && g.s[0] == 0.0 && g.s[1] == 0.0)) return false;
        }
        return nz_of(res) == 2;
    }
    // A recorded linear Horner loop (horner_linear_rest with `defer`): everything its launch needs.
    struct LazyHorner {
        P res, ca;                  // incoming accumulator and coefficient tensor (keep their buffers alive)
        HornerLoopArgs g;
        unsigned lines = 0;
        size_t fn = 0;
    };
    // Launches a loop — a recorded one into its own buffer, or a fresh one (the unbatched form; a recording with "batch_dag" on is
    // issued with its level of the launch graph, gft_api_dag.inc HornerRec).
    static void launch_horner(const P& res, const P& ca, double* outp, size_t fn, const HornerLoopArgs& g, unsigned lines, unsigned* wit) {
        const double* cp = dp<E>(ca);
        const bool rview = res_is_view(res, ca);  // (as when the arguments were built: the handle keeps its own chain)
        const double* rp = rview ? cp + res.pend->base_off : dp<E>(res);
        const size_t rplane = rview ? res.pend->base_numel : res.numel;
        K<E>::horner_linear_loop(R.stream, rp, rplane, cp, ca.numel, outp, fn, g, lines, wit);
    }
    static bool horner_linear_rest(const P& res, const P& ca, size_t v, size_t i, const double c[2], const double m[2], size_t w,
                                   const Dims& deg, P* result, unsigned* wit, const unsigned* guard = nullptr, bool defer = false) {
        const size_t nd = deg.size();
        Dims oc = ca.shape;
        oc[v] = 1;
        for (size_t ax = 0; ax < nd; ++ax) oc[ax] = std::min(oc[ax], deg[ax]);
        const bool coeff_scalar = prod(oc) == 1;
        Dims fs = res.shape;  // shape after all remaining steps
        for (size_t t = 0; t <= i; ++t) {
            fs[w] = std::min(deg[w], fs[w] + 1);
            if (!coeff_scalar)
                for (size_t ax = 0; ax < nd; ++ax) fs[ax] = std::max(fs[ax], oc[ax]);
        }
        const size_t fn = prod(fs);
        if (fn > R.horner_loop_max || fs[w] > K<E>::HORNER_LINE_MAX || fn / fs[w] > 0x7fffffffu) return false;
        Dims keep = collapse_mask({&fs}, false);
        if (keep.size() > (size_t)MAXD) return false;
        P out = make_recorded(fs, deg);  // (memory when the loop is launched: here below, or with its level of the launch graph)
        HornerLoopArgs g;
        std::memset(&g, 0, sizeof(g));
        g.nd = (int)keep.size();
        const bool rview = res_is_view(res, ca);
        Dims rst = c_strides(rview ? res.pend->base_shape : res.shape), fst = c_strides(fs), ast = c_strides(ca.shape);
        g.w = -1;
        for (size_t j = 0; j < keep.size(); ++j) {
            size_t ax = keep[j];
            g.deg[j] = (unsigned)std::min<size_t>(deg[ax], 0x7fffffffu);
            g.rs0[j] = (unsigned)res.shape[ax];
            g.fs[j] = (unsigned)fs[ax];
            g.oc[j] = (unsigned)oc[ax];
            g.rstr0[j] = rst[ax];
            g.fstr[j] = fst[ax];
            g.astr[j] = ax == v ? 0 : ast[ax];
            if (ax == w) g.w = (int)j;
        }
        if (g.w < 0) return false;
        g.a_vstride = ast[v];
        g.first_i = (unsigned)i;
        g.nsteps = (unsigned)(i + 1);
        g.c = Scalar2{c[0], W == 2 ? c[1] : 0.0};
        g.m = Scalar2{m[0], W == 2 ? m[1] : 0.0};
        g.c_zero = val_is_zero(c) ? 1 : 0;
        g.c_one = val_is_one(c) ? 1 : 0;
        g.coeff_scalar = coeff_scalar ? 1 : 0;
        g.lw_pad = (unsigned)((fs[w] + 7) / 8 * 8);
        g.diag = 0;
        g.stat = nullptr;
        g.guard = guard;
        // (intervals, c and m not [0,0]) no exact zero among the coefficients, none in the result: position (o, k_w) of the final
        // box receives coeff_i[o, k'] * C(i,j) c^(i-j) m^j for every k' + j = k_w — at least one such term exists, none cancels
        if (W == 2) {
            // (round 6, Support) leading zero slabs on the other axes stay where they are; along the substituted axis every
            // position receives a term from the top coefficient slab
            Support sp = support_of_poly<E>(ca);
            if (sp.exact() && !val_is_zero(c) && !val_is_zero(m)) {
                if (v < (size_t)Buf::ZAX) sp.z[v] = 0;
                sp.kind = 3;
                sp.normalise();
            } else
                sp.kind = 0;
            sp.store(out.buf.get());
        }
        const unsigned lines = (unsigned)(fn / fs[w]);
        if (defer && !wit && res.buf && ca.buf) {
            g.guard = nullptr;
            if (K<E>::horner_can_ride(g)) {
                auto lh = std::allocate_shared<LazyHorner>(gft_small::Alloc<LazyHorner>());
                lh->res = res;
                lh->ca = ca;
                lh->g = g;
                lh->lines = lines;
                lh->fn = fn;
                auto op = std::allocate_shared<LazyOp>(gft_small::Alloc<LazyOp>());
                op->horner = lh;
                op->run = [lh](Buf* b) { launch_horner(lh->res, lh->ca, b->p, lh->fn, lh->g, lh->lines, nullptr); };
                auto rec = std::allocate_shared<HornerRec>(gft_small::Alloc<HornerRec>());
                rec->lh = lh;
                op->rec = rec;
                out.buf->lazy = op;
                *result = out;
                return true;
            }
            g.guard = guard;
        }
        ensure_alloc(out.buf.get());
        launch_horner(res, ca, dp<E>(out), fn, g, lines, wit);
        *result = out;
        return true;
    }
    // Steps i0, i0-1, .., 0 of the linear Horner loop on HOST-resident operands, one pass of HK::horner_linear per step
    // (exactly horner_linear_step's arguments, unit axes kept instead of collapsed), without the per-step handles.
    static bool horner_linear_rest_host(const P& res0, const P& ca, size_t v, size_t i0, const double c[2], const double m[2],
                                        size_t w, const Dims& deg, P* result) {
        const size_t nd = deg.size();
        if (nd == 0 || nd > (size_t)MAXD || res0.shape.size() != nd || ca.shape.size() != nd) return false;
        Dims oc = ca.shape;
        oc[v] = 1;
        for (size_t ax = 0; ax < nd; ++ax) oc[ax] = std::min(oc[ax], deg[ax]);
        const bool coeff_scalar = prod(oc) == 1;
        auto step_shapes = [&](const Dims& rs, Dims& sh, Dims& os) {
            sh = rs;
            sh[w] = std::min(deg[w], sh[w] + 1);
            os = sh;
            if (!coeff_scalar)
                for (size_t ax = 0; ax < nd; ++ax) os[ax] = std::min(std::max(sh[ax], oc[ax]), deg[ax]);
        };
        Dims rs = res0.shape, sh, os, fs = res0.shape;
        size_t cap = 0;
        for (size_t t = 0; t <= i0; ++t) {  // largest intermediate (shapes only grow)
            step_shapes(fs, sh, os);
            fs = os;
            cap = std::max(cap, prod(os));
        }
        if (!tier_host(cap, res0, ca)) return false;
        P out = make(fs, deg, true);
        std::shared_ptr<Buf> ping = alloc_host_doubles(cap * W), pong = alloc_host_doubles(cap * W);
        const double* src = hp<E>(res0);
        size_t src_plane = res0.numel;
        const Dims ast = c_strides(ca.shape);
        HornerArgs g;
        std::memset(&g, 0, sizeof(g));
        g.c = Scalar2{c[0], W == 2 ? c[1] : 0.0};
        g.m = Scalar2{m[0], W == 2 ? m[1] : 0.0};
        g.c_zero = val_is_zero(c) ? 1 : 0;
        g.c_one = val_is_one(c) ? 1 : 0;
        g.coeff_scalar = coeff_scalar ? 1 : 0;
        g.w = (int)w;
        g.out.nd = (int)nd;
        for (size_t ax = 0; ax < nd; ++ax) {
            g.oc[ax] = (unsigned)oc[ax];
            g.astr[ax] = ax == v ? 0 : ast[ax];
        }
        for (size_t i = i0, t = 0;; --i, ++t) {
            step_shapes(rs, sh, os);
            size_t stride = 1;
            for (size_t ax = nd; ax-- > 0;) {
                g.out.d[ax] = (unsigned)os[ax];
                g.rs[ax] = (unsigned)rs[ax];
                g.sh[ax] = (unsigned)sh[ax];
                g.rstr[ax] = stride;
                stride *= rs[ax];
            }
            g.a_base = i * ast[v];
            g.upper = (unsigned)std::min(sh[w] - 1, rs[w]);
            const size_t n_out = prod(os);
            double* dst = i == 0 ? hp<E>(out) : ((t & 1) ? pong->p : ping->p);
            HK<E>::horner_linear(src, src_plane, hp<E>(ca), ca.numel, dst, n_out, g);
            rs = os;
            src = dst;
            src_plane = n_out;
            if (i == 0) break;
        }
        *result = seal(out);
        return true;
    }
    static P horner_linear_step(const P& res, const P& ca, size_t v, size_t i, const double c[2], const double m[2], size_t w,
                                const Dims& deg, const unsigned* guard = nullptr) {
        const size_t nd = deg.size();
        Dims rs = res.shape, sh = res.shape, oc = ca.shape;
        sh[w] = std::min(deg[w], sh[w] + 1);
        oc[v] = 1;
        for (size_t ax = 0; ax < nd; ++ax) oc[ax] = std::min(oc[ax], deg[ax]);
        const bool coeff_scalar = prod(oc) == 1;
        Dims os = sh;
        if (!coeff_scalar)
            for (size_t ax = 0; ax < nd; ++ax) os[ax] = std::min(std::max(sh[ax], oc[ax]), deg[ax]);
        const bool host = tier_host(prod(os), res, ca);
        P out = make(os, deg, host);
        Dims keep = collapse_mask({&os}, false);
        if (keep.size() > (size_t)MAXD) throw Error("tensor rank exceeds GFT MAXD after collapsing");
        HornerArgs g;
        std::memset(&g, 0, sizeof(g));
        g.out = to_shape(pick(os, keep));
        Dims rst = c_strides(rs), ast = c_strides(ca.shape);
        g.w = -1;
        for (size_t j = 0; j < keep.size(); ++j) {
            size_t ax = keep[j];
            g.rs[j] = (unsigned)rs[ax];
            g.sh[j] = (unsigned)sh[ax];
            g.oc[j] = (unsigned)oc[ax];
            g.rstr[j] = rst[ax];
            g.astr[j] = ax == v ? 0 : ast[ax];
            if (ax == w) g.w = (int)j;
        }
        if (g.w < 0) throw Error("horner_linear_step: substitution axis collapsed");  // sh[w] >= 2 => kept
        g.a_base = i * ast[v];
        g.upper = (unsigned)std::min(sh[w] - 1, rs[w]);
        g.c = Scalar2{c[0], W == 2 ? c[1] : 0.0};
        g.m = Scalar2{m[0], W == 2 ? m[1] : 0.0};
        g.c_zero = val_is_zero(c) ? 1 : 0;
        g.c_one = val_is_one(c) ? 1 : 0;
        g.coeff_scalar = coeff_scalar ? 1 : 0;
        if (host) {
            HK<E>::horner_linear(hp<E>(res), res.numel, hp<E>(ca), ca.numel, hp<E>(out), out.numel, g);
            return seal(out);
        }
        g.guard = guard;
        K<E>::horner_linear(R.stream, dp<E>(res), res.numel, dp<E>(ca), ca.numel, dp<E>(out), out.numel, g);
        return out;
    }
    static P with_meta_unchecked(const P& src, const Dims& shape) {
        P r = src;
        r.shape = shape;
        return r;
    }

    // ---- slab extraction (mt:341-404) --------------------------------------------------------------------------------------
    static P coefficients_of_term(const P& a, size_t v, size_t order) {
        if (v >= a.shape.size()) return order == 0 ? a : zero_with(a.deg);
        if (order >= a.shape[v]) return zero_with(a.deg);
        return slab_range(a, v, order, order + 1, a.deg);
    }
    static P taylor_polynomial_terms(const P& a, size_t v, const std::vector<size_t>& orders) {  // a list of orders, not a shape
        size_t max_order_p1 = 1;
        for (size_t o : orders) max_order_p1 = std::max(max_order_p1, o + 1);
        bool has0 = std::find(orders.begin(), orders.end(), (size_t)0) != orders.end();
        if (v >= a.shape.size()) return has0 ? a : zero_with(a.deg);
        size_t upper = std::min(a.shape[v], max_order_p1);
        std::vector<unsigned char> keep(max_order_p1, 0);
        for (size_t o : orders) keep[o] = 1;
        Dims out = a.shape;
        out[v] = upper;
        Shifts shift(out.size(), 0);
        if (gather_tier(a, out)) return gather(a, out, a.deg, shift, a.shape, OP_COPY, nullptr, (int)v, nullptr, 0, keep.data(), 1);
        std::shared_ptr<Buf> kb = alloc_doubles((upper + 7) / 8 + 1);
        HIP_OK(hipMemcpyAsync(kb->p, keep.data(), upper, hipMemcpyHostToDevice, R.stream));
        HIP_OK(hipStreamSynchronize(R.stream));
        return gather(a, out, a.deg, shift, a.shape, OP_COPY, nullptr, (int)v, nullptr, 0, (const unsigned char*)kb->p, 0);
    }

    // ---- metadata ops (mt:81-112, 172-193) -------------------------------------------------------------------------------------
    static P extend_to_dim(const P& a, size_t ndim, size_t degree_p1) {
        if (a.shape.size() > ndim) throw Error("extend_to_dim: ndim smaller than current");
        Dims s = a.shape, d = a.deg;
        while (s.size() < ndim) s.push_back(1);
        d.resize(ndim, degree_p1);
        return with_meta(a, s, d);
    }
    static P extend(const P& a, const Dims& ns) {
        if (a.deg.size() > ns.size()) throw Error("extend: too few dims");
        Dims s = a.shape;
        while (s.size() < ns.size()) s.push_back(1);
        for (size_t v = 0; v < s.size(); ++v)
            if (s[v] > ns[v]) throw Error("extend: shape exceeds new size");
        P src = with_meta_unchecked(a, s);
        Shifts shift(ns.size(), 0);
        return gather(src, ns, ns, shift, s);
    }
    static P remove_last_variable(const P& a) {
        if (a.deg.empty()) throw Error("remove_last_variable: attempt to subtract with overflow (no variables)");
        size_t v = a.deg.size() - 1;
        Dims d = a.deg;
        d.pop_back();
        Dims s = a.shape;
        if (v < s.size()) {
            if (s[v] != 1) {
                Dims lens = s;
                lens[v] = 1;
                P blk = lead_block(with_meta_unchecked(a, s), lens, a.deg);
                lens.pop_back();
                return with_meta(blk, lens, d);
            }
            s.pop_back();
        }
        return with_meta(a, s, d);
    }
    static P truncate_to_degree_p1(const P& a, size_t degree_p1) {
        return truncate_degrees(a, Dims(a.deg.size(), degree_p1));
    }

    // ---- coefficient (mt:314-339) ------------------------------------------------------------------------------------------------
    static void coefficient(const P& a, const Dims& index, double out[2]) {
        size_t consumed = 0, off = 0;
        Dims st = c_strides(a.shape);
        for (size_t v = 0; v < index.size(); ++v) {
            size_t idx = index[v];
            size_t len_of = v < a.deg.size() ? a.deg[v] : UMAX;
            if (!(idx < len_of)) throw Error("index out of bounds");
            if (v >= a.shape.size()) {
                if (idx != 0) {
                    out[0] = out[1] = 0.0;
                    return;
                }
            } else if (idx >= a.shape[v]) {
                out[0] = out[1] = 0.0;
                return;
            } else {
                off += idx * st[v];
                consumed++;
            }
        }
        if (consumed != a.shape.size()) throw Error("index is too short");
        out[1] = 0.0;
        if (on_host(a)) {
            const double* h = hp<E>(a);
            out[0] = h[off];
            if (W == 2) out[1] = h[a.numel + off];
            return;
        }
        const double* d = dp<E>(a);
        Buf* b = a.buf.get();
        if (!b->host_copy && ++b->coef_reads >= 2 && a.numel * W * sizeof(double) <= ((size_t)32 << 20)) {
            std::shared_ptr<Buf> m = alloc_host_doubles(a.numel * W);
            HIP_OK(hipMemcpyAsync(m->p, d, sizeof(double) * a.numel * W, hipMemcpyDeviceToHost, R.stream));
            HIP_OK(hipStreamSynchronize(R.stream));
            b->host_copy = m;
            R.stats[2]++;
        }
        if (b->host_copy) {
            out[0] = b->host_copy->p[off];
            if (W == 2) out[1] = b->host_copy->p[a.numel + off];
            return;
        }
        R.stats[2]++;
        peek(out, d + off, a.numel, W);
    }

    // `impl Display for TaylorPoly` = fmt_polynomial (mt:694-730): non-zero coefficients in row-major order, each
    // followed by its variables ("a".."z", then x_<i>; "^e" above 1), joined by " + "; "0" if there is none.
    // `debug`: `impl Debug` (mt:632-636) = "TaylorPoly({:?}, {})" of degrees_p1 (a Vec<usize>: "[4, 5]", usize::MAX in
    // full) and of the coefficient ARRAY through ndarray's Display: nested brackets with every stored element, zeros
    // included (gft_fmt.hpp fmt_ndarray).
    static std::string format(const P& a, bool debug) {
        std::vector<double> h(a.numel * W);
        if (on_host(a)) std::memcpy(h.data(), hp<E>(a), sizeof(double) * h.size());
        else {
            HIP_OK(hipMemcpyAsync(h.data(), dp<E>(a), sizeof(double) * h.size(), hipMemcpyDeviceToHost, R.stream));
            HIP_OK(hipStreamSynchronize(R.stream));
        }
        auto num = [&](size_t i) {
            if (W == 1) return gftfmt::fmt_f64(h[i]);
            return "[" + gftfmt::fmt_f64(h[i]) + ", " + gftfmt::fmt_f64(h[a.numel + i]) + "]";  // interval.rs:243-247
        };
        if (debug) {
            std::string d = "TaylorPoly([";
            for (size_t i = 0; i < a.deg.size(); ++i) d += (i ? ", " : "") + std::to_string(a.deg[i]);
            return d + "], " + gftfmt::fmt_ndarray(a.shape.begin(), a.shape.size(), num) + ")";
        }
        std::string out;
        bool first = true;
        Dims idx(a.shape.size(), 0);
        for (size_t lin = 0; lin < a.numel; ++lin) {
            const bool zero = W == 1 ? h[lin] == 0.0 : (h[lin] == 0.0 && h[a.numel + lin] == 0.0);
            if (!zero) {
                if (!first) out += " + ";
                first = false;
                out += num(lin);
                for (size_t i = 0; i < idx.size(); ++i) {
                    if (idx[i] == 0) continue;
                    out += i < 26 ? std::string(1, (char)('a' + i)) : "x_" + std::to_string(i);  // ppl.rs:107-117
                    if (idx[i] > 1) out += "^" + std::to_string(idx[i]);
                }
            }
            for (size_t ax = idx.size(); ax-- > 0;) {
                if (++idx[ax] < a.shape[ax]) break;
                idx[ax] = 0;
            }
        }
        if (first) out = "0";
        return out;
    }

#include "gft_api_dag.inc"

    static bool equal(const P& a, const P& b) {
        if (a.deg != b.deg || a.shape != b.shape) return false;
        if (on_host(a) && on_host(b)) return HK<E>::count_neq(hp<E>(a), a.numel, hp<E>(b), b.numel, a.numel) == 0;

        HIP_OK(hipMemsetD32Async((hipDeviceptr_t)(R.d_flag + 1), 0, 1, R.stream));
        K<E>::count_neq(R.stream, dp<E>(a), a.numel, dp<E>(b), b.numel, a.numel, R.d_flag + 1);
        unsigned cnt = 0;
        read_back(&cnt, R.d_flag + 1, sizeof(unsigned));
        return cnt == 0;
    }
};

static Dims dims(const size_t* p, size_t n) { return Dims(p, p + n); }

// GFT_TRACE_API=1: calls per entry point, printed at exit
struct ApiTrace {
    bool on = getenv("GFT_TRACE_API") != nullptr;
    std::map<std::string, size_t> counts;
    std::map<std::string, size_t> tiny;  // results of at most 2 elements that live in DEVICE memory, by entry point
    std::map<std::string, double> secs;  // host wall time inside the entry point (host-tier ops: their compute time)
    std::map<std::string, size_t> settles;  // deferred chains materialised (one launch each), by the entry point that needed the values
    const char* cur = "?";
    void hit(const char* fn) {
        cur = fn;
        if (on) counts[fn]++;
    }
    struct Timer {
        ApiTrace& t;
        const char* fn;
        std::chrono::steady_clock::time_point t0;
        Timer(ApiTrace& tr, const char* f) : t(tr), fn(f) {
            if (t.on) t0 = std::chrono::steady_clock::now();
        }
        ~Timer() {
            if (t.on) t.secs[fn] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    };
    void result(const char* fn, const gft_poly& r) {
        if (on && r.numel <= 2 && r.buf && !r.buf->host) tiny[fn]++;
    }
    ~ApiTrace() {
        if (!on) return;
        for (auto& kv : counts) fprintf(stderr, "[gft api] %-40s %10zu calls %10.4f s\n", kv.first.c_str(), kv.second, secs[kv.first]);
        for (auto& kv : tiny) fprintf(stderr, "[gft api] tiny device result from %-22s %zu\n", kv.first.c_str(), kv.second);
        fprintf(stderr, "[gft api] host-tier Horner steps: positive constants %llu, sign-known c %llu, other %llu; elements of a sign-known step that left its fast path %llu\n",
                gft::g_host_horner_stats[0], gft::g_host_horner_stats[1], gft::g_host_horner_stats[2], gft::g_host_horner_stats[3]);
        for (auto& kv : settles) fprintf(stderr, "[gft api] chains materialised for %-24s %zu\n", kv.first.c_str(), kv.second);
    }
};
static ApiTrace g_api_trace;
static void trace_settle() {
    if (g_api_trace.on) g_api_trace.settles[g_api_trace.cur]++;
}
static void trace_mirror(size_t numel) {
    if (g_api_trace.on)
        g_api_trace.settles[std::string("(host-tier tensor mirrored to the device, ") + (numel <= 2 ? "<= 2" : (numel <= 64 ? "<= 64" : "> 64")) + " elements, in " +
                            (g_scan_trace.ctx ? g_scan_trace.ctx : "-") + ") " + g_api_trace.cur]++;
}

// Genfer-style programs are launch-bound (10^5 dependent kernels of 2-6 us): the HIP runtime places kernel arguments in
// device memory when HIP_FORCE_DEV_KERNARG=1, which shortens every launch (hmm -25 %, mixture -9 % on the same box).  The
// flag is read when the HIP runtime initialises and it is process-wide, so it is the HOST's to set: this library never
// touches the environment (round 2 called setenv from a static constructor).  The entry points of this repo that own
// their process set it before HIP comes up (the `genfer` executable, bench.py, the genfer_amd Python package);
// INTEGRATION.md tells a Rust host to do the same; bench.py records the value its process ran with.

void dist_set_min_macs(double v);  // multi-GPU section below
void dist_set_event_slot(double v);

template <class F>
static gft_poly* guard(F&& f, const char* fn = __builtin_FUNCTION()) {
    try {
        require_ready();
        g_api_trace.hit(fn);
        ApiTrace::Timer timer(g_api_trace, fn);
        gft_poly* r = new gft_poly(f());
        if (r->buf && !r->buf->origin) r->buf->origin = fn;
        g_api_trace.result(fn, *r);
        return r;
    } catch (const std::exception& e) {
        g_err = e.what();
        g_scan_mail_open = 0;
        return nullptr;
    }
}
template <class F>
static int guard_int(F&& f, const char* fn = __builtin_FUNCTION()) {
    try {
        require_ready();
        g_api_trace.hit(fn);
        ApiTrace::Timer timer(g_api_trace, fn);
        return f();
    } catch (const std::exception& e) {
        g_err = e.what();
        g_scan_mail_open = 0;
        return -1;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI — runtime
// ------------------------------------------------------------------------------------------
extern "C" {

int gft_init(int device) {
    if (R.ready) return 0;
    try {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n == 0) throw Error("no HIP device visible");
        if (device < 0) {
            const char* lr = getenv("LOCAL_RANK");
            device = lr ? atoi(lr) % n : 0;
        }
        HIP_OK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_OK(hipGetDeviceProperties(&prop, device));
        if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
            throw Error(std::string("device is ") + prop.gcnArchName + ", but libgftaylor is built for gfx950 only");
        HIP_OK(hipStreamCreateWithFlags(&R.own_stream, hipStreamNonBlocking));
        R.stream = R.own_stream;
        HIP_OK(hipStreamCreateWithFlags(&R.side, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&R.ev_main, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&R.ev_bulk, hipEventDisableTiming));
        // (words 0 .. 63: predicates and counters; from byte 1024 on: the scans' group arrival counters, one per 128 bytes —
        // gft_kernels.hip scan_arrive)
        HIP_OK(hipMalloc((void**)&R.d_flag, 1024 + 128 * 64));
        HIP_OK(hipMemset(R.d_flag, 0, 1024 + 128 * 64));
        {
            unsigned init[64] = {0};
            init[8] = 0xffffffffu;  // linear_scan state: mask word, arrival counter
            HIP_OK(hipMemcpy(R.d_flag, init, sizeof(init), hipMemcpyHostToDevice));
        }
        HIP_OK(hipMalloc((void**)&R.d_scratch, 256));
        HIP_OK(hipMalloc((void**)&R.d_wit, sizeof(unsigned) * 8192));
        HIP_OK(hipHostMalloc((void**)&R.h_pinned, 4096, hipHostMallocDefault));
        HIP_OK(hipHostMalloc((void**)&R.h_mail, 4096, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(R.h_mail, 0, 4096);
        HIP_OK(hipHostGetDevicePointer((void**)&R.d_mail, R.h_mail, 0));
        for (auto& ev : R.events) HIP_OK(hipEventCreate(&ev));
        if (const char* lo = getenv("GFT_LAZY_OBSERVE")) R.lazy_observe = atoi(lo) != 0;
        if (const char* bd = getenv("GFT_BATCH")) R.batch_dag = atoi(bd) != 0;
        if (const char* lh = getenv("GFT_LAZY_HORNER")) R.lazy_horner = atoi(lh) != 0;
        if (const char* np = getenv("GFT_NZ_PROOFS")) R.nz_proofs = atoi(np) != 0;
        if (const char* lsum = getenv("GFT_LAZY_SUM")) R.lazy_sum = atoi(lsum) != 0;
        R.device = device;
        if (const char* tm = getenv("GFT_TILED_MIN_MACS")) {  // tuning knob for the auto-mode crossover
            double v = atof(tm);
            if (v >= 0) R.tiled_min_macs = v;
        }
        if (const char* dw = getenv("GFT_DIV_WAVEFRONT")) R.div_wavefront = atoi(dw) != 0;
        if (const char* er = getenv("GFT_EXP_RIGHT")) R.exp_right = atoi(er) != 0;
        if (const char* df = getenv("GFT_DEFER")) R.defer = atoi(df) != 0;
        {
            const char* al = getenv("GFT_ASYNC_LAUNCH");
            lq_configure(device, al ? atoi(al) != 0 : true);
        }
        if (const char* hm = getenv("GFT_HOST_MAX_ELEMS")) R.host_max_elems = (size_t)atoll(hm);
        if (const char* hm = getenv("GFT_HOST_MAX_MACS")) R.host_max_macs = atof(hm);
        if (const char* hl = getenv("GFT_HORNER_LOOP_MAX")) R.horner_loop_max = (size_t)atoll(hl);
        if (const char* sm = getenv("GFT_SHALLOW_MAX_TERMS")) R.shallow_max_terms = (size_t)atoll(sm);
        if (const char* cm = getenv("GFT_CONV_MODE")) {  // test knob, same meaning as gft_set_conv_mode
            int m = atoi(cm);
            if (m >= 0 && m <= 3) R.conv_mode = m;
        }
        R.ready = true;
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

void gft_shutdown(void) {
    if (!R.ready) return;
    lq_shutdown();
    (void)hipStreamSynchronize(R.stream);
    if (R.side) (void)hipStreamSynchronize(R.side);
    for (auto& c : g_pow_tabs) c.clear();  // device tables of this context: back into the pool before it is freed
    dwf_release_orders();
    staged_release_scratch();
    g_arena.release();
    (void)gft_dist_shutdown();  // the communicator refers to this device and its streams
    for (auto& kv : R.host_blocks)
        for (void* q : kv.second) std::free(q);
    R.host_blocks.clear();
    for (auto& kv : R.free_blocks)
        for (void* q : kv.second) (void)hipFree(q);
    R.free_blocks.clear();
    R.cached = 0;
    if (R.conv_ws) (void)hipFree(R.conv_ws);
    R.conv_ws = nullptr;
    R.conv_ws_bytes = 0;
    (void)hipFree(R.d_flag);
    (void)hipFree(R.d_scratch);
    (void)hipFree(R.d_wit);
    (void)hipHostFree(R.h_pinned);
    (void)hipHostFree(R.h_mail);
    R.h_mail = R.d_mail = nullptr;
    for (auto& ev : R.events) (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(R.own_stream);
    if (R.side) (void)hipStreamDestroy(R.side);
    if (R.ev_main) (void)hipEventDestroy(R.ev_main);
    if (R.ev_bulk) (void)hipEventDestroy(R.ev_bulk);
    R.side = nullptr;
    R.ev_main = R.ev_bulk = nullptr;
    R.side_pending = false;
    R.ready = false;
}

int gft_set_stream(void* s) {
    return guard_int([&] {
        HIP_OK(hipStreamSynchronize(R.stream));
        R.stream = s ? (hipStream_t)s : R.own_stream;
        return 0;
    });
}
void* gft_get_stream(void) { return (void*)R.stream; }
int gft_synchronize(void) {
    return guard_int([&] {
        HIP_OK(hipStreamSynchronize(R.stream));
        return 0;
    });
}
const char* gft_last_error(void) { return g_err.c_str(); }
const char* gfti_last_error(void) { return g_err.c_str(); }
void gft_op_stats(size_t out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = R.stats[i];
}
size_t gft_op_stats_ex(size_t* out, size_t cap) {
    const size_t v[18] = {(size_t)gft::g_launches, R.stats_ex[0], R.stats_ex[1], R.stats_ex[2], (size_t)gft::g_launches_in_place,
                          R.stats_shallow[0], R.stats_shallow[1], R.stats_side[0], R.stats_side[1], R.stats_side[2], R.stats_side[3], R.stats_nz, R.stats_sum,
                          g_dag_stats[0], g_dag_stats[1], g_dag_stats[2], g_dag_stats[3], g_dag_stats[4]};
    for (size_t i = 0; i < 18 && i < cap; ++i) out[i] = v[i];
    return 18;
}
void gft_pool_stats(size_t out[3]) {
    // (the grow-only kernel workspaces — row-pair sums, row flags, the tiled product's — are not pool blocks: counted here so that
    // a host sees what the library holds; pool_alloc's out-of-memory retry releases the staged kernels' ones)
    const size_t ws = staged_scratch_bytes() + R.conv_ws_bytes;
    out[0] = R.in_use + ws;
    out[1] = R.cached;
    R.peak_total = std::max(R.peak_total, R.in_use + ws);  // (sampled at every new pool high and here: a simultaneous figure)
    out[2] = R.peak_total;
}
int gft_event_record(int slot) {
    return guard_int([&] {
        if (slot < 0 || slot >= 64) throw Error("event slot out of range");
        HIP_OK(hipEventRecord(R.events[slot], R.stream));
        return 0;
    });
}
float gft_event_elapsed_ms(int a, int b) {
    try {
        require_ready();
        if (a < 0 || a >= 64 || b < 0 || b >= 64) throw Error("event slot out of range");
        HIP_OK(hipEventSynchronize(R.events[b]));
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, R.events[a], R.events[b]));
        return ms;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1.0f;
    }
}
int gft_set_option(const char* name, double value) {
    std::string n = name ? name : "";
    if (n == "horner_loop_max") R.horner_loop_max = value < 0 ? 0 : (size_t)value;
    else if (n == "div_wavefront") R.div_wavefront = value != 0;
    else if (n == "exp_right") R.exp_right = value != 0;
    else if (n == "recur_overlap") R.recur_overlap = value != 0;
    else if (n == "defer") R.defer = value != 0;
    else if (n == "lazy_observe") R.lazy_observe = value != 0;
    else if (n == "batch_dag") R.batch_dag = value != 0;
    else if (n == "lazy_horner") R.lazy_horner = value != 0;
    else if (n == "nz_proofs") R.nz_proofs = value != 0;
    else if (n == "lazy_sum") R.lazy_sum = value != 0;
    else if (n == "async_launch") lq_configure(R.device, value != 0);
    else if (n == "trace_lq_report") lq_report();  // (GFT_TRACE_LQ=1; measurement aid)
    else if (n == "shallow_max_terms") R.shallow_max_terms = value < 0 ? 256 : (size_t)value;  // < 0: default
    else if (n == "debug_fail_next_launch") g_fail_next_launch.store(value != 0 ? 1 : 0);  // test knob (gft_launch.hpp)
    else if (n == "tiled_min_macs") R.tiled_min_macs = value;
    else if (n == "conv_rb_min_macs") staged_set_rb_min_macs(value);
    else if (n == "conv_rb_pairs") staged_set_rb_pairs(value);
    else if (n == "conv_rb_pairs_cap") staged_set_rb_pairs_cap(value);
    else if (n == "conv_rb_pairs_lanes") staged_set_rb_pairs_lanes(value);
    else if (n == "shallow_pair_min") shallow_set_pair_min(value == -1.0 ? 4096.0 : value);
    else if (n == "tiled_tile") tiled_set_lane_tile((int)value);
    else if (n == "host_max_elems") R.host_max_elems = value < 0 ? Runtime::HOST_MAX_ELEMS_DEFAULT : (size_t)value;  // < 0: default
    else if (n == "host_max_macs") R.host_max_macs = value < 0 ? Runtime::HOST_MAX_MACS_DEFAULT : value;
    else if (n == "dist_min_macs") dist_set_min_macs(value);
    else if (n == "dist_event_slot") dist_set_event_slot(value);
    else return -1;
    return 0;
}
int gft_set_conv_variant(int v) {
    R.conv_variant = v;
    return 0;
}
int gft_set_conv_mode(int mode) {
    if (mode < 0 || mode > 3) return -1;
    R.conv_mode = mode;
    return 0;
}

// ---- raw entry points ------------------------------------------------------------------------
int gft_conv_raw(const double* x, const size_t* xshape, const double* y, const size_t* yshape, double* res,
                 const size_t* rshape, size_t ndim, size_t slab_lo, size_t slab_hi, int accumulate) {
    return guard_int([&] {
        typedef Ops<EF64> O;
        O::HV xv{const_cast<double*>(x), 0, dims(xshape, ndim)};
        O::HV yv{const_cast<double*>(y), 0, dims(yshape, ndim)};
        O::HV zv{res, 0, dims(rshape, ndim)};
        for (size_t i = 0; i < ndim; ++i)
            if (xshape[i] > rshape[i] || yshape[i] > rshape[i] || xshape[i] == 0 || yshape[i] == 0)
                throw Error("conv_raw: operand shapes must be non-empty and not exceed the result shape");
        if (ndim > 0 && (slab_lo > slab_hi || slab_hi > rshape[0])) throw Error("conv_raw: bad slab range");
        // Whole-product semantics on the selected slabs: same summation structure as Mul's general path.
        O::conv(xv, yv, zv, slab_lo, slab_hi, accumulate != 0, false, 0, 0, 0);
        launch_drain();  // raw entry point on the caller's stream: every launch is in the stream when this returns
        return 0;
    });
}

double gft_conv_macs(const size_t* xs, const size_t* ys, const size_t* rs, size_t ndim, size_t slab_lo,
                     size_t slab_hi) {
    auto pairs = [](size_t sx, size_t sy, size_t k) -> double {
        size_t lo = k + 1 > sy ? k + 1 - sy : 0, hi = std::min(k + 1, sx);
        return hi > lo ? (double)(hi - lo) : 0.0;
    };
    double inner = 1.0;
    for (size_t a = 1; a < ndim; ++a) {
        double s = 0;
        for (size_t k = 0; k < rs[a]; ++k) s += pairs(xs[a], ys[a], k);
        inner *= s;
    }
    if (ndim == 0) return 1.0;
    double total = 0;
    for (size_t k = slab_lo; k < slab_hi && k < rs[0]; ++k) total += pairs(xs[0], ys[0], k) * inner;
    return total;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// multi-GPU behind the C ABI: one process per GPU, RCCL over xGMI, no torch on the path
// ------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double min_macs = 1.0e10;  // gft_mul shards a general product at or above this many multiply-adds ("dist_min_macs")
    int ev_slot = -1;          // "dist_event_slot": the next sharded products record the event slots s (start), s + 1 (local
                               // kernels done), s + 2 (exchange done), then s += 3: kernel-only and exchange-only time (bench.py)
};
Rccl D;
void dist_set_min_macs(double v) { D.min_macs = v; }
void dist_set_event_slot(double v) { D.ev_slot = (v >= 0 && v <= 61) ? (int)v : -1; }

static void rccl_load() {
    if (D.lib) return;
    // the copy a host process already loaded (PyTorch-ROCm bundles one) wins; else ROCm's
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        D.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (D.lib) break;
    }
    if (!D.lib) throw Error(std::string("cannot load RCCL: ") + dlerror());
#define GFT_NCCL(f)                                                          \
    D.f = reinterpret_cast<decltype(D.f)>(dlsym(D.lib, "nccl" #f));          \
    if (!D.f) throw Error("RCCL does not export nccl" #f)
    GFT_NCCL(GetUniqueId); GFT_NCCL(CommInitRank); GFT_NCCL(CommCount); GFT_NCCL(CommDestroy); GFT_NCCL(AllGather);
    GFT_NCCL(AllReduce); GFT_NCCL(Broadcast); GFT_NCCL(Send); GFT_NCCL(Recv); GFT_NCCL(GroupStart); GFT_NCCL(GroupEnd);
    GFT_NCCL(GetErrorString);
#undef GFT_NCCL
}
#define NCCL_OK(call)                                                                                       \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) throw Error(std::string("RCCL error: ") + D.GetErrorString(r_) + " in " #call); \
    } while (0)

// Exchange of a sharded result whose leading axis has n0 slabs of `slab` doubles each: every rank has computed the
// slab groups gft_plan_slabs gives it and ends up with all of them.  Even split: the low groups lie in rank order (an
// in-place all-gather); the mirrored high groups lie in REVERSE rank order, exchanged as grouped point-to-point
// sends/receives (xGMI is point-to-point; each rank pushes its group to its 7 peers on 7 links).  Uneven split:
// zero-filled all-reduce (adding zeros is exact).  `even` and the ranges come from gft_plan_slabs.
static void dist_exchange(double* z, size_t n0, size_t slab, bool zeroed_outside) {
    if (D.world <= 1) return;
    launch_drain();  // the collectives go to the stream from this thread: after every queued launch
    size_t mine[4];
    const bool even = gft_plan_slabs(n0, D.world, D.rank, mine) != 0;
    if (!even) {
        if (!zeroed_outside) throw Error("internal: uneven sharded product without a zeroed result");
        NCCL_OK(D.AllReduce(z, z, n0 * slab, ncclDouble, ncclSum, D.comm, R.stream));
        return;
    }
    const size_t b0 = mine[1] - mine[0], b1 = mine[3] - mine[2];
    if (b0) NCCL_OK(D.AllGather(z + mine[0] * slab, z, b0 * slab, ncclDouble, D.comm, R.stream));  // in place: rank r's group at r * b0
    if (b1) {
        NCCL_OK(D.GroupStart());
        for (int peer = 0; peer < D.world; ++peer) {
            if (peer == D.rank) continue;
            size_t theirs[4];
            gft_plan_slabs(n0, D.world, peer, theirs);
            NCCL_OK(D.Send(z + mine[2] * slab, b1 * slab, ncclDouble, peer, D.comm, R.stream));
            NCCL_OK(D.Recv(z + theirs[2] * slab, (theirs[3] - theirs[2]) * slab, ncclDouble, peer, D.comm, R.stream));
        }
        NCCL_OK(D.GroupEnd());
    }
}

// z = x (*) y, leading output axis sharded over the communicator (operands replicated on every rank)
template <class O>
static void dist_conv(const typename O::HV& x, const typename O::HV& y, const typename O::HV& z) {
    const size_t n0 = z.shape[0];
    size_t slab = 1;
    for (size_t i = 1; i < z.shape.size(); ++i) slab *= z.shape[i];
    size_t mine[4];
    const bool even = gft_plan_slabs(n0, D.world, D.rank, mine) != 0;
    if (!even) HIP_OK(hipMemsetAsync(z.p, 0, sizeof(double) * n0 * slab, R.stream));
    const int ev = D.ev_slot;
    if (ev >= 0) HIP_OK(hipEventRecord(R.events[ev], R.stream));
    struct AfterLocal {
        int ev;
        ~AfterLocal() {
            if (ev >= 0) {
                launch_drain_nothrow();  // (destructor: a latched launch failure is raised by the next throwing drain)
                (void)(hipEventRecord)(R.events[ev + 1], R.stream);
                D.ev_slot = ev + 2 <= 62 ? ev + 2 : -1;
            }
        }
    };
    {
    AfterLocal after{ev};
    if (mine[1] == mine[2]) {  // the two groups touch: one launch
        O::conv(x, y, z, mine[0], mine[3], false, false, 0, 0, 0);
    } else {
        O::conv(x, y, z, mine[0], mine[1], false, false, 0, 0, 0);
        O::conv(x, y, z, mine[2], mine[3], false, false, 0, 0, 0);
    }
    }
    dist_exchange(z.p, n0, slab, !even);
    if (ev >= 0 && ev + 2 <= 63) {  // slot s + 2: after the exchange (exchange-only time = [s + 1, s + 2])
        (void)hipEventRecord(R.events[ev + 2], R.stream);
        D.ev_slot = ev + 3 <= 61 ? ev + 3 : -1;
    }
}
}  // namespace

template <class E>
bool Ops<E>::dist_shard(const P& self, const P& other, const P& out) {
    // Every rank runs the same program on replicated operands ("replicas" outside this one operation), so
    // every rank reaches this product with the same shapes and takes the same decision.
    if (!D.comm || D.world <= 1 || out.shape.empty() || out.shape[0] < (size_t)(2 * D.world)) return false;
    if (gft_conv_macs(self.shape.begin(), other.shape.begin(), out.shape.begin(), out.shape.size(), 0, out.shape[0]) < D.min_macs) return false;
    dist_conv<Ops<E>>(view(self), view(other), view(out));
    return true;
}

extern "C" {

int gft_dist_unique_id(void* out128) {
    return guard_int([&] {
        rccl_load();
        ncclUniqueId id;
        NCCL_OK(D.GetUniqueId(&id));
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
        std::memcpy(out128, &id, sizeof id);
        return 0;
    });
}
int gft_dist_init(int rank, int world, const void* unique_id128) {
    return guard_int([&] {
        if (world < 1 || rank < 0 || rank >= world) throw Error("gft_dist_init: bad rank / world");
        if (D.comm) throw Error("gft_dist_init: already initialised");
        rccl_load();
        ncclUniqueId id;
        std::memcpy(&id, unique_id128, sizeof id);
        NCCL_OK(D.CommInitRank(&D.comm, world, id, rank));
        D.rank = rank;
        D.world = world;
        if (const char* st = getenv("GFT_DIST_SELFTEST"))  // every rank proves the exchange before anything is computed with it
            if (atoi(st) != 0 && gft_dist_selftest() != 0) throw Error(g_err);
        return 0;
    });
}
// Every rank: two small sharded products (an even split -> all-gather + point-to-point, an uneven one -> zero-filled
// all-reduce) through BOTH entries — the raw sharded product and gft_mul's auto-shard — compared with the rank's own
// full product, computed by the reference-order kernel (bit-identical whatever the slab ranges, so any difference is
// the exchange's).  0 = every slab arrived where it belongs; -1 (and gft_last_error) otherwise.
int gft_dist_selftest(void) {
    return guard_int([&] {
        typedef Ops<EF64> O;
        if (!D.comm) throw Error("gft_dist_selftest: gft_dist_init has not been called");
        const int saved_mode = R.conv_mode;
        const double saved_min = D.min_macs;
        struct Restore {
            int m;
            double d;
            ~Restore() {
                R.conv_mode = m;
                D.min_macs = d;
            }
        } restore{saved_mode, saved_min};
        R.conv_mode = 1;  // reference order: the same bits from any slab range
        D.min_macs = 0.0;
        for (size_t n0 : {(size_t)16 * (size_t)std::max(1, D.world / 8 + (D.world % 8 ? 1 : 0)), (size_t)(2 * D.world + 1)}) {
            const Dims shape{n0, 12, 10};
            const size_t n = prod(shape);
            std::vector<double> hx(n), hy(n);
            unsigned long long sx = 0x9E3779B97F4A7C15ull * 7, sy = 0x9E3779B97F4A7C15ull * 11;
            auto next = [](unsigned long long& st) {
                st += 0x9E3779B97F4A7C15ull;
                unsigned long long zz = st;
                zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
                zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
                zz ^= zz >> 31;
                return (double)(zz >> 11) * (1.0 / 9007199254740992.0) - 0.25;
            };
            for (size_t i = 0; i < n; ++i) {
                hx[i] = next(sx);
                hy[i] = next(sy);
            }
            gft_poly px = O::make(shape, shape), py = O::make(shape, shape);
            HIP_OK(hipMemcpyAsync(dp<EF64>(px), hx.data(), sizeof(double) * n, hipMemcpyHostToDevice, R.stream));
            HIP_OK(hipMemcpyAsync(dp<EF64>(py), hy.data(), sizeof(double) * n, hipMemcpyHostToDevice, R.stream));
            HIP_OK(hipStreamSynchronize(R.stream));
            gft_poly full = O::make(shape, shape), raw = O::make(shape, shape);
            O::conv(O::view(px), O::view(py), O::view(full), 0, n0, false, false, 0, 0, 0);   // local, all slabs
            dist_conv<O>(O::view(px), O::view(py), O::view(raw));                             // sharded, raw entry
            gft_poly handle = O::mul(px, py);                                                 // sharded inside gft_mul
            if (!O::equal(full, raw)) throw Error("gft_dist_selftest: the sharded raw product differs from the local product (n0 = " + std::to_string(n0) + ")");
            if (!O::equal(full, handle)) throw Error("gft_dist_selftest: gft_mul's sharded product differs from the local product (n0 = " + std::to_string(n0) + ")");
        }
        return 0;
    });
}
int gft_dist_world(void) { return D.world; }
int gft_dist_rank(void) { return D.rank; }
int gft_dist_comm_count(void) {
    int n = 0;
    if (D.comm && D.CommCount(D.comm, &n) != ncclSuccess) return -1;
    return D.comm ? n : 0;
}
int gft_dist_shutdown(void) {
    return guard_int([&] {
        if (D.comm) {
            HIP_OK(hipStreamSynchronize(R.stream));
            NCCL_OK(D.CommDestroy(D.comm));
        }
        D.comm = nullptr;
        D.rank = 0;
        D.world = 1;
        return 0;
    });
}
int gft_dist_broadcast(double* buf, size_t count, int root) {
    return guard_int([&] {
        if (!D.comm) throw Error("gft_dist_broadcast: gft_dist_init has not been called");
        launch_drain();
        NCCL_OK(D.Broadcast(buf, buf, count, ncclDouble, root, D.comm, R.stream));
        return 0;
    });
}
int gft_conv_raw_sharded(const double* x, const size_t* xshape, const double* y, const size_t* yshape, double* res,
                         const size_t* rshape, size_t ndim) {
    return guard_int([&] {
        typedef Ops<EF64> O;
        if (ndim == 0) throw Error("conv_raw_sharded: a scalar product does not shard");
        for (size_t i = 0; i < ndim; ++i)
            if (xshape[i] > rshape[i] || yshape[i] > rshape[i] || xshape[i] == 0 || yshape[i] == 0)
                throw Error("conv_raw_sharded: operand shapes must be non-empty and not exceed the result shape");
        O::HV xv{const_cast<double*>(x), 0, dims(xshape, ndim)};
        O::HV yv{const_cast<double*>(y), 0, dims(yshape, ndim)};
        O::HV zv{res, 0, dims(rshape, ndim)};
        dist_conv<O>(xv, yv, zv);
        launch_drain();
        return 0;
    });
}

int gft_plan_slabs(size_t n0, int world, int rank, size_t out[4]) {
    // Folded assignment: work(k) ~ k+1, so pair low slab group r with the mirrored high
    // group.  Boundaries are ceil-split so any n0 / world works; groups may be empty.
    if (world < 1 || rank < 0 || rank >= world) {
        out[0] = out[1] = out[2] = out[3] = 0;
        return 0;
    }
    size_t half = n0 / 2;            // low half [0, half), high half [half, n0) mirrored
    size_t G = (size_t)world;
    auto cut = [&](size_t len, size_t i) { return (len * i) / G; };
    size_t lo_len = half, hi_len = n0 - half;
    out[0] = cut(lo_len, (size_t)rank);
    out[1] = cut(lo_len, (size_t)rank + 1);
    // mirrored: rank r takes the r-th chunk counted from the top
    out[2] = n0 - cut(hi_len, (size_t)rank + 1);
    out[3] = n0 - cut(hi_len, (size_t)rank);
    return (lo_len % G == 0 && hi_len % G == 0) ? 1 : 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// C ABI — handle API, generated for both element types
// ------------------------------------------------------------------------------------------
#define GFT_API(PFX, E)                                                                                       \
    extern "C" {                                                                                              \
    int PFX##width(void) { return E::W; }                                                                     \
    gft_poly* PFX##from_host(const double* c, const size_t* sh, const size_t* dg, size_t nd) {                \
        return guard([&] {                                                                                    \
            Dims shape = dims(sh, nd);                                                                        \
            const size_t n = prod(shape);                                                                     \
            if (n == 2 && nd >= 1) { /* [e0, e1] along one axis: a host value like var()'s (no buffer until a kernel reads it) */ \
                size_t ax = 0;                                                                                \
                for (size_t i = 0; i < nd; ++i)                                                               \
                    if (shape[i] == 2) ax = i;                                                                \
                return Ops<E>::affine_like(ax, c, shape, dims(dg, nd));                                       \
            }                                                                                                 \
            const bool host = R.host_max_elems && n <= R.host_max_elems;  /* small: stays host-resident */    \
            gft_poly r = Ops<E>::make(shape, dims(dg, nd), host);                                             \
            if (host) {                                                                                       \
                std::memcpy(hp<E>(r), c, sizeof(double) * n * E::W);                                          \
                return Ops<E>::seal(r);                                                                       \
            }                                                                                                 \
            HIP_OK(hipMemcpyAsync(dp<E>(r), c, sizeof(double) * r.numel * E::W, hipMemcpyHostToDevice, R.stream)); \
            HIP_OK(hipStreamSynchronize(R.stream));                                                           \
            return r;                                                                                         \
        });                                                                                                   \
    }                                                                                                         \
    gft_poly* PFX##scalar(const double* x) { return guard([&] { return Ops<E>::scalar(x); }); }               \
    gft_poly* PFX##from_u32(uint32_t c) {                                                                     \
        return guard([&] {                                                                                    \
            double v[2] = {(double)c, (double)c};                                                             \
            return Ops<E>::scalar(v);                                                                         \
        });                                                                                                   \
    }                                                                                                         \
    gft_poly* PFX##zero_with(const size_t* dg, size_t nd) { return guard([&] { return Ops<E>::zero_with(dims(dg, nd)); }); } \
    gft_poly* PFX##var(size_t v, const double* x, size_t len) {                                               \
        return guard([&] { return Ops<E>::var_like(v, x, true, std::min<size_t>(len, 2), len > 1, Dims(v + 1, len)); }); \
    }                                                                                                         \
    gft_poly* PFX##var_at_zero(size_t v, size_t len) {                                                        \
        return guard([&] { return Ops<E>::var_like(v, nullptr, false, 2, len > 1, Dims(v + 1, len)); });      \
    }                                                                                                         \
    gft_poly* PFX##var_with_degrees_p1(size_t v, const double* x, const size_t* dg, size_t nd) {              \
        return guard([&] {                                                                                    \
            Dims d = dims(dg, nd);                                                                            \
            if (v >= nd) throw Error("var_with_degrees_p1: index out of bounds");                             \
            return Ops<E>::var_like(v, x, true, 2, d[v] > 1, d);                                              \
        });                                                                                                   \
    }                                                                                                         \
    gft_poly* PFX##clone(const gft_poly* p) { return guard([&] { return *p; }); }                             \
    void PFX##free(gft_poly* p) { delete p; }                                                                 \
    size_t PFX##num_vars(const gft_poly* p) { return p->deg.size(); }                                         \
    size_t PFX##numel(const gft_poly* p) { return p->numel; }                                                 \
    void PFX##shape(const gft_poly* p, size_t* out) { std::copy(p->shape.begin(), p->shape.end(), out); }     \
    void PFX##degrees_p1(const gft_poly* p, size_t* out) { std::copy(p->deg.begin(), p->deg.end(), out); }    \
    int PFX##to_host(const gft_poly* p, double* out) {                                                        \
        return guard_int([&] {                                                                                \
            if (on_host(*p)) {                                                                                \
                std::memcpy(out, hp<E>(*p), sizeof(double) * p->numel * E::W);                                \
                return 0;                                                                                     \
            }                                                                                                 \
            HIP_OK(hipMemcpyAsync(out, dp<E>(*p), sizeof(double) * p->numel * E::W, hipMemcpyDeviceToHost, R.stream)); \
            HIP_OK(hipStreamSynchronize(R.stream));                                                           \
            return 0;                                                                                         \
        });                                                                                                   \
    }                                                                                                         \
    size_t PFX##len_of(const gft_poly* p, size_t v) { return v < p->deg.size() ? p->deg[v] : UMAX; }          \
    int PFX##is_constant(const gft_poly* p) { return p->numel == 1; }                                         \
    int PFX##is_zero(const gft_poly* p) { return guard_int([&] { return (int)Ops<E>::is_zero(*p); }); }       \
    int PFX##is_one(const gft_poly* p) { return guard_int([&] { return (int)Ops<E>::is_one(*p); }); }         \
    int PFX##equal(const gft_poly* a, const gft_poly* b) { return guard_int([&] { return (int)Ops<E>::equal(*a, *b); }); } \
    long PFX##format(const gft_poly* p, int debug, char* out, size_t cap) {                                   \
        long need = -1;                                                                                       \
        (void)guard_int([&] {                                                                                 \
            std::string s = Ops<E>::format(*p, debug != 0);                                                   \
            need = (long)s.size();                                                                            \
            if (out && cap) {                                                                                 \
                size_t n = std::min(cap - 1, s.size());                                                       \
                std::memcpy(out, s.data(), n);                                                                \
                out[n] = 0;                                                                                   \
            }                                                                                                 \
            return 0;                                                                                         \
        });                                                                                                   \
        return need;                                                                                          \
    }                                                                                                         \
    int PFX##constant_term(const gft_poly* p, double* out) {                                                  \
        return guard_int([&] {                                                                                \
            double v[2];                                                                                      \
            Ops<E>::first_value(*p, v);                                                                       \
            for (int i = 0; i < E::W; ++i) out[i] = v[i];                                                     \
            return 0;                                                                                         \
        });                                                                                                   \
    }                                                                                                         \
    int PFX##extract_constant(const gft_poly* p, double* out) {                                               \
        return guard_int([&] {                                                                                \
            if (p->numel != 1) return 0;                                                                      \
            double v[2];                                                                                      \
            Ops<E>::first_value(*p, v);                                                                       \
            for (int i = 0; i < E::W; ++i) out[i] = v[i];                                                     \
            return 1;                                                                                         \
        });                                                                                                   \
    }                                                                                                         \
    int PFX##extract_linear(const gft_poly* p, double* c, double* m, size_t* v) {                             \
        return guard_int([&] {                                                                                \
            double cc[2], mm[2];                                                                              \
            if (!Ops<E>::extract_linear(*p, cc, mm, v)) return 0;                                             \
            for (int i = 0; i < E::W; ++i) {                                                                  \
                c[i] = cc[i];                                                                                 \
                m[i] = mm[i];                                                                                 \
            }                                                                                                 \
            return 1;                                                                                         \
        });                                                                                                   \
    }                                                                                                         \
    int PFX##coefficient(const gft_poly* p, const size_t* idx, size_t n, double* out) {                       \
        return guard_int([&] {                                                                                \
            double v[2];                                                                                      \
            Ops<E>::coefficient(*p, dims(idx, n), v);                                                         \
            for (int i = 0; i < E::W; ++i) out[i] = v[i];                                                     \
            return 0;                                                                                         \
        });                                                                                                   \
    }                                                                                                         \
    gft_poly* PFX##add(const gft_poly* a, const gft_poly* b) { return guard([&] { return Ops<E>::addsub(*a, *b, false); }); } \
    gft_poly* PFX##sub(const gft_poly* a, const gft_poly* b) { return guard([&] { return Ops<E>::addsub(*a, *b, true); }); } \
    gft_poly* PFX##neg(const gft_poly* a) { return guard([&] { return Ops<E>::neg(*a); }); }                  \
    gft_poly* PFX##add_scaled(const gft_poly* a, const gft_poly* b, const double* c) {                        \
        return guard([&] { return Ops<E>::add_scaled(*a, *b, c); });                                          \
    }                                                                                                         \
    gft_poly* PFX##mul(const gft_poly* a, const gft_poly* b) { return guard([&] { return Ops<E>::mul(*a, *b); }); } \
    gft_poly* PFX##div(const gft_poly* a, const gft_poly* b) { return guard([&] { return Ops<E>::div(*a, *b); }); } \
    gft_poly* PFX##exp(const gft_poly* a) { return guard([&] { return Ops<E>::exp(*a); }); }                  \
    gft_poly* PFX##log(const gft_poly* a) { return guard([&] { return Ops<E>::log(*a); }); }                  \
    gft_poly* PFX##pow(const gft_poly* a, uint32_t e) { return guard([&] { return Ops<E>::pow(*a, e); }); }   \
    gft_poly* PFX##derivative(const gft_poly* a, size_t v, size_t n) {                                        \
        return guard([&] { return Ops<E>::deriv_like(*a, v, n, TAB_DERIV, "derivative"); });                  \
    }                                                                                                         \
    gft_poly* PFX##taylor_expansion_of_coeff(const gft_poly* a, size_t v, size_t n) {                         \
        return guard([&] { return Ops<E>::deriv_like(*a, v, n, TAB_COEFF, "taylor_expansion_of_coeff"); });   \
    }                                                                                                         \
    gft_poly* PFX##shift_down(const gft_poly* a, size_t v, size_t n) { return guard([&] { return Ops<E>::shift_down(*a, v, n); }); } \
    gft_poly* PFX##derivative_truncated(const gft_poly* a, size_t v, size_t n, size_t d) {                    \
        return guard([&] { return Ops<E>::derivative_truncated(*a, v, n, d); });                              \
    }                                                                                                         \
    gft_poly* PFX##observe_step(const gft_poly* a, size_t v, const double* x, const double* c, size_t d) {    \
        return guard([&] { return R.lazy_observe ? Ops<E>::observe_chain(*a, v, x, c, 1, d) : Ops<E>::observe_step(*a, v, x, c, d); }); \
    }                                                                                                         \
    gft_poly* PFX##observe_chain(const gft_poly* a, size_t v, const double* x, const double* cs, size_t n, size_t d) { \
        return guard([&] { return Ops<E>::observe_chain(*a, v, x, cs, n, d); });                              \
    }                                                                                                         \
    gft_poly* PFX##derive_scale(const gft_poly* a, size_t v, const double* c, size_t d) {                     \
        return guard([&] { return Ops<E>::derive_scale(*a, v, c, d); });                                      \
    }                                                                                                         \
    gft_poly* PFX##subst_var(const gft_poly* a, size_t v, const gft_poly* s) {                                \
        return guard([&] { return Ops<E>::subst_var(*a, v, *s); });                                           \
    }                                                                                                         \
    gft_poly* PFX##coefficients_of_term(const gft_poly* a, size_t v, size_t o) {                              \
        return guard([&] { return Ops<E>::coefficients_of_term(*a, v, o); });                                 \
    }                                                                                                         \
    gft_poly* PFX##taylor_polynomial_terms(const gft_poly* a, size_t v, const size_t* orders, size_t n) {     \
        return guard([&] { return Ops<E>::taylor_polynomial_terms(*a, v, std::vector<size_t>(orders, orders + n)); });                \
    }                                                                                                         \
    gft_poly* PFX##truncate_to_degree_p1(const gft_poly* a, size_t d) {                                       \
        return guard([&] { return Ops<E>::truncate_to_degree_p1(*a, d); });                                   \
    }                                                                                                         \
    gft_poly* PFX##remove_last_variable(const gft_poly* a) { return guard([&] { return Ops<E>::remove_last_variable(*a); }); } \
    gft_poly* PFX##extend_to_dim(const gft_poly* a, size_t nd, size_t d) {                                    \
        return guard([&] { return Ops<E>::extend_to_dim(*a, nd, d); });                                       \
    }                                                                                                         \
    gft_poly* PFX##extend(const gft_poly* a, const size_t* ns, size_t n) {                                    \
        return guard([&] { return Ops<E>::extend(*a, dims(ns, n)); });                                        \
    }                                                                                                         \
    gft_poly* PFX##mul_var(const gft_poly* a, const double* m, size_t v, const size_t* sh, const size_t* dg, size_t n) { \
        return guard([&] { return Ops<E>::mul_var(*a, m, v, dims(sh, n), dims(dg, n)); });                    \
    }                                                                                                         \
    gft_poly* PFX##mul_linear(const gft_poly* a, const double* c, const double* m, size_t v, const size_t* sh, \
                              const size_t* dg, size_t n) {                                                   \
        return guard([&] { return Ops<E>::mul_linear(*a, c, m, v, dims(sh, n), dims(dg, n)); });              \
    }                                                                                                         \
    }

GFT_API(gft_, EF64)
GFT_API(gfti_, EIv)
