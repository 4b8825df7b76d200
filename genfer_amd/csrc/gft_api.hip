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

// Reference counts inside the library are NOT atomic: the header's contract is one calling thread, the launch thread's closures
// capture raw device pointers (never a handle's buffer object), and a locked increment / decrement per handle copy was 7 % of the
// calling thread on mixture.  libstdc++'s shared_ptr with the single-threaded lock policy is that type.
template <class U>
using Rc = std::__shared_ptr<U, __gnu_cxx::_S_single>;
template <class U, class A, class... Args>
static inline Rc<U> rc_allocate(const A& a, Args&&... args) {
    return std::__allocate_shared<U, __gnu_cxx::_S_single>(a, std::forward<Args>(args)...);
}
template <class U, class... Args>
static inline Rc<U> rc_make(Args&&... args) {
    return std::__make_shared<U, __gnu_cxx::_S_single>(std::forward<Args>(args)...);
}
struct LazyOp;
struct Buf {
    double* p = nullptr;
    size_t cls = 0;
    bool borrowed = false;
    bool host = false;           // p is host memory (host tier); `dev` is its device mirror once a kernel needed it
    // the contents have not been launched yet (a recorded observation chain): use_buf() launches, or the consumer fuses
    Rc<LazyOp> lazy;
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
    Rc<Buf> dev;
    // device tensors whose coefficients are read one by one (probs_taylor / moments_taylor read `limit` of them,
    // generating_function.rs:963,992): the second read mirrors the whole (immutable) buffer to the host once
    Rc<Buf> host_copy;
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
    Rc<DagRec> rec;   // (round 6) the recording as a node of the deferred launch graph (gft_batch.hpp); null: launched by run() only
    Rc<void> obs;     // Ops<E>::LazyObs for the fused form (typed by the element class that recorded it)
    Rc<void> horner;  // Ops<E>::LazyHorner: a recorded linear Horner loop (rides along with another loop's launch)
    Rc<void> sum;     // Ops<E>::LazySum: a recorded Add / Sub of two chains (an Add that consumes it launches both: K<E>::chain_nest)
};

// (+ 8 doubles of slack: the tiled product reads operands in place and its pipelined x loads request one 64-byte chunk
// beyond the last one they use — gft_conv_tiled.hip, ConvArgs::operands_slack)
static Rc<Buf> alloc_doubles(size_t n) {
    auto b = rc_allocate<Buf>(gft_small::Alloc<Buf>());
    b->p = (double*)pool_alloc((std::max<size_t>(n, 1) + 8) * sizeof(double), &b->cls);
    return b;
}
// a recording's result: no memory yet (ensure_alloc, when the recording is launched)
static Rc<Buf> alloc_recorded(size_t n) {
    auto b = rc_allocate<Buf>(gft_small::Alloc<Buf>());
    b->want = std::max<size_t>(n, 1);
    return b;
}
static void ensure_alloc(Buf* b) {
    if (b->p || b->host) return;
    b->p = (double*)pool_alloc((std::max<size_t>(b->want, 1) + 8) * sizeof(double), &b->cls);
}
static Rc<Buf> alloc_host_doubles(size_t n) {
    auto b = rc_allocate<Buf>(gft_small::Alloc<Buf>());
    b->host = true;
    b->p = (double*)host_alloc(std::max<size_t>(n, 1) * sizeof(double), &b->cls);
    return b;
}
static Rc<Buf> alloc_tier(bool host, size_t n) { return host ? alloc_host_doubles(n) : alloc_doubles(n); }

static void force_buf(Buf* b);
static void ensure_alloc(Buf* b);
#include "gft_batch.hpp"
// Every access to a device buffer's contents on behalf of work about to be issued on the stream.
static inline void use_buf(Buf* b) {
    if (!b->host && b->lazy) force_buf(b);
}
static void force_buf(Buf* b) {
    Rc<LazyOp> op = b->lazy;
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
    Rc<Buf> dev;   // W planes of `len` doubles
    std::vector<double> host;   // the same values (element 0 of a chain is host-computable)
    size_t len = 0;
    int nz = -1;                // (intervals) 1: no entry is exactly [0,0]; -1: not looked at yet
};
struct PendStage {
    int kind = 0, axis = 0;
    double s[2] = {0, 0};
    Rc<TabEntry> tab;
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
    Rc<Buf> mat;   // the materialised tensor once some consumer needed it (shared by all copies of the handle)
    Dims mat_shape;
};

std::map<std::tuple<unsigned long long, unsigned long long>, Rc<TabEntry>> g_pow_tabs[2];  // Ops<E>::pow_table

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
    mutable Rc<Buf> buf;   // width * numel doubles, plane stride == numel (null: lazy host-cached scalar)
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
    mutable Rc<Pend> pend;
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
        Rc<Buf> out = alloc_doubles(p.numel * E::W);
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

#include "gft_ops_core.inc"

#include "gft_ops_product.inc"

#include "gft_ops_recur.inc"

#include "gft_ops_observe.inc"

#include "gft_ops_horner.inc"

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
        Rc<Buf> kb = alloc_doubles((upper + 7) / 8 + 1);
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
            Rc<Buf> m = alloc_host_doubles(a.numel * W);
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
        fprintf(stderr, "[gft api] host-tier Horner elements: sign-known runs %llu held / %llu failed, finite runs %llu / %llu, lines without runs %llu (sign-known c) + %llu; element form stored by: sign-known %llu, finite %llu, positive %llu, general %llu\n",
                gft::g_host_horner_stats[4], gft::g_host_horner_stats[5], gft::g_host_horner_stats[6], gft::g_host_horner_stats[7], gft::g_host_horner_stats[8], gft::g_host_horner_stats[9],
                gft::g_host_horner_stats[10], gft::g_host_horner_stats[11], gft::g_host_horner_stats[12], gft::g_host_horner_stats[13]);
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
