// Structural / streaming / reduction kernels and the reference-order convolution for gfx950.
// All of these are HBM- or latency-bound index/copy work: coalesced along the innermost axis,
// grid-stride loops capped at 2048 workgroups of 256 threads (4 waves), no LDS needed.  The
// compute-bound product lives in gft_conv_tiled.hip.
#include "gft_kernels.hpp"
#include "gft_kernels_int.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>

namespace gft {

unsigned long long g_launches = 0;
unsigned long long g_host_horner_stats[16] = {0};  // gft_host.hpp (GFT_TRACE_API)
bool g_host_horner_runs = true;                             // gft_host.hpp: the finite regime of the host Horner step in runs (tests: false)
int g_host_simd = -1;                                          // gft_host.hpp: the AVX2 clones of those runs (-1: when the CPU has AVX2, 0: never)
unsigned long long g_launches_in_place = 0;
unsigned long long g_stream_ops = 0;

// ------------------------------------------------------------------------------------------
// the launch thread (gft_launch.hpp): single-producer / single-consumer ring of launch closures
// ------------------------------------------------------------------------------------------
namespace {
constexpr uint64_t LQ_SLOTS = 1024;  // power of two; 4 MB of slots
struct LaunchQueue {
    std::atomic<uint64_t> head{0};   // consumer: slots [0, head) have been issued
    std::atomic<uint64_t> tail{0};   // producer: slots [0, tail) have been written
    LaunchSlot* slots = nullptr;
    std::thread worker;
    std::atomic<bool> stop{false}, sleeping{false};
    std::mutex m;
    std::condition_variable cv;
    int device = -1;
    bool enabled = true;
    std::thread::id worker_id;

    // GFT_TRACE_LQ=1: where the two threads' time goes — the worker inside the queued closures (hipLaunchKernel and the
    // like) vs waiting for the producer, the producer waiting for a free slot / for the worker to catch up (drain); printed
    // by gft_shutdown / at exit.  Off: one predictable branch per item.
    bool trace = getenv("GFT_TRACE_LQ") != nullptr;
    uint64_t tr_items = 0, tr_issue_ns = 0, tr_backlog_sum = 0, tr_empty = 0, tr_max_ns = 0;
    uint64_t tr_full_ns = 0, tr_full_n = 0, tr_drain_ns = 0, tr_drain_n = 0;
    std::chrono::steady_clock::time_point tr_first, tr_last;
    static uint64_t ns_since(std::chrono::steady_clock::time_point t0) {
        return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    void report() {
        if (!trace || !tr_items) return;
        const double span = std::chrono::duration<double>(tr_last - tr_first).count();
        fprintf(stderr,
                "[GFT_TRACE_LQ] worker: %llu items, %.3f ms inside them (mean %.2f us, max %.1f us) over %.3f ms first-to-last; "
                "found the ring empty %llu times, mean backlog when taking an item %.1f\n"
                "[GFT_TRACE_LQ] producer: waited %.3f ms for a free slot (%llu times), %.3f ms in %llu drains\n",
                (unsigned long long)tr_items, tr_issue_ns / 1e6, tr_issue_ns / 1e3 / tr_items, tr_max_ns / 1e3, span * 1e3,
                (unsigned long long)tr_empty, (double)tr_backlog_sum / tr_items, tr_full_ns / 1e6, (unsigned long long)tr_full_n,
                tr_drain_ns / 1e6, (unsigned long long)tr_drain_n);
        tr_items = tr_issue_ns = tr_backlog_sum = tr_empty = tr_max_ns = tr_full_ns = tr_full_n = tr_drain_ns = tr_drain_n = 0;
    }

    void run() {
        if (device >= 0) (void)hipSetDevice(device);
        uint64_t h = head.load(std::memory_order_relaxed);
        unsigned idle = 0;
        for (;;) {
            const uint64_t t_now = tail.load(std::memory_order_acquire);
            if (h != t_now) {
                LaunchSlot& s = slots[h & (LQ_SLOTS - 1)];
                if (trace) {
                    const auto t0 = std::chrono::steady_clock::now();
                    if (!tr_items) tr_first = t0;
                    s.run(s.payload);
                    tr_last = std::chrono::steady_clock::now();
                    const uint64_t d = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(tr_last - t0).count();
                    tr_issue_ns += d;
                    if (d > tr_max_ns) tr_max_ns = d;
                    tr_backlog_sum += t_now - h;
                    if (idle) tr_empty++;
                    tr_items++;
                } else
                    s.run(s.payload);
                head.store(++h, std::memory_order_release);
                idle = 0;
                continue;
            }
            if (stop.load(std::memory_order_acquire)) return;
            if (++idle < 20000) {  // ~100 us of polling: the producer is rarely further away than that inside a program
                __builtin_ia32_pause();
                continue;
            }
            std::unique_lock<std::mutex> lk(m);
            sleeping.store(true, std::memory_order_seq_cst);
            cv.wait(lk, [&] { return h != tail.load(std::memory_order_seq_cst) || stop.load(std::memory_order_seq_cst); });
            sleeping.store(false, std::memory_order_seq_cst);
            idle = 0;
        }
    }
    void start() {
        if (!slots) slots = new LaunchSlot[LQ_SLOTS];
        stop.store(false);
        worker = std::thread([this] { run(); });
        worker_id = worker.get_id();
    }
    void shutdown() {
        if (!worker.joinable()) return;
        drain();
        {
            std::lock_guard<std::mutex> lk(m);
            stop.store(true, std::memory_order_seq_cst);
        }
        cv.notify_all();
        worker.join();
    }
    void drain() {
        if (!worker.joinable() || std::this_thread::get_id() == worker_id) return;
        const uint64_t t = tail.load(std::memory_order_relaxed);
        if (trace && head.load(std::memory_order_acquire) < t) {
            const auto t0 = std::chrono::steady_clock::now();
            while (head.load(std::memory_order_acquire) < t) __builtin_ia32_pause();
            tr_drain_ns += ns_since(t0);
            tr_drain_n++;
            return;
        }
        while (head.load(std::memory_order_acquire) < t) __builtin_ia32_pause();
    }
    ~LaunchQueue() {
        shutdown();
        report();
    }
};
LaunchQueue g_lq;
}  // namespace

std::atomic<int> g_fail_next_launch{0};
namespace {
std::mutex g_lq_err_m;
std::atomic<bool> g_lq_err_set{false};
hipError_t g_lq_err = hipSuccess;
const void* g_lq_err_kernel = nullptr;
const char* g_lq_err_what = nullptr;
}  // namespace
void lq_note(hipError_t e, const void* kernel, const char* what) {
    if (e == hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_lq_err_m);
    if (g_lq_err_set.load(std::memory_order_relaxed)) return;  // the first failure is the one that explains the rest
    g_lq_err = e;
    g_lq_err_kernel = kernel;
    g_lq_err_what = what;
    g_lq_err_set.store(true, std::memory_order_release);
}
static void lq_raise() {
    if (!g_lq_err_set.load(std::memory_order_acquire)) return;
    std::string msg;
    {
        std::lock_guard<std::mutex> lk(g_lq_err_m);
        msg = std::string("HIP error: ") + hipGetErrorString(g_lq_err);
        if (g_lq_err_kernel) {
            const char* name = hipKernelNameRefByPtr(g_lq_err_kernel, nullptr);
            msg += std::string(" launching ") + (name ? name : "a kernel");
        } else if (g_lq_err_what) {
            msg += std::string(" in queued ") + g_lq_err_what;
        }
        msg += " (issued by the launch thread; results computed since are invalid)";
        g_lq_err_set.store(false, std::memory_order_relaxed);
    }
    throw std::runtime_error(msg);
}

bool lq_enabled() { return g_lq.enabled; }
int lq_debug() {
    static const int d = 0;
    return d;
}
void lq_configure(int device, bool enabled) {
    g_lq.drain();
    g_lq.device = device;
    g_lq.enabled = enabled;
}
void lq_shutdown() {
    g_lq.shutdown();
    g_lq.report();
    g_lq_err_set.store(false);  // a failure nobody asked about dies with the runtime
}
void lq_report() {
    g_lq.drain();
    g_lq.report();
}
void launch_drain_nothrow() { g_lq.drain(); }
void launch_drain() {
    g_lq.drain();
    lq_raise();
}
LaunchSlot* lq_begin() {
    if (!g_lq.worker.joinable()) g_lq.start();
    const uint64_t t = g_lq.tail.load(std::memory_order_relaxed);
    if (g_lq.trace && t - g_lq.head.load(std::memory_order_acquire) >= LQ_SLOTS) {
        const auto t0 = std::chrono::steady_clock::now();
        while (t - g_lq.head.load(std::memory_order_acquire) >= LQ_SLOTS) __builtin_ia32_pause();
        g_lq.tr_full_ns += LaunchQueue::ns_since(t0);
        g_lq.tr_full_n++;
    }
    while (t - g_lq.head.load(std::memory_order_acquire) >= LQ_SLOTS) __builtin_ia32_pause();  // ring full: the worker is behind
    return &g_lq.slots[t & (LQ_SLOTS - 1)];
}
void lq_commit() {
    g_lq.tail.store(g_lq.tail.load(std::memory_order_relaxed) + 1, std::memory_order_seq_cst);
    if (g_lq.sleeping.load(std::memory_order_seq_cst)) {
        std::lock_guard<std::mutex> lk(g_lq.m);
        g_lq.cv.notify_one();
    }
}


// ------------------------------------------------------------------------------------------
// gather: structured strided copy with optional scaling / per-slab table / keep-mask
// ------------------------------------------------------------------------------------------
template <class E>
__global__ void __launch_bounds__(256) k_gather(const double* __restrict__ src, size_t src_plane,
                                                double* __restrict__ out, size_t out_plane, GatherArgs a,
                                                size_t total) {
    typedef typename E::V V;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        size_t r = lin;
        size_t soff = 0;
        bool valid = true;
        unsigned kaxis = 0;
#pragma unroll 1
        for (int ax = a.out.nd - 1; ax >= 0; --ax) {
            unsigned d = a.out.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            long long si = (long long)k + a.shift[ax];
            if (si < 0 || si >= (long long)a.src_len[ax]) valid = false;
            soff += (size_t)(si < 0 ? 0 : si) * a.src_stride[ax];
            if (ax == a.tab_axis) kaxis = k;
        }
        if (valid && a.keep && !a.keep[kaxis]) valid = false;
        V v = E::zero();
        if (valid) {
            v = E::ld(src, src_plane, soff);
            switch (a.op) {
                case OP_MUL_S: v = E::mul(v, E::from(a.s)); break;
                case OP_DIV_S: v = E::div(v, E::from(a.s)); break;
                case OP_LMUL_S: v = E::mul(E::from(a.s), v); break;
                case OP_NEG: v = E::neg(v); break;
                case OP_MUL_TAB: v = E::mul(v, E::ld(a.tab, a.tab_plane, kaxis)); break;
                case OP_MUL_TAB_LMUL_S: v = E::mul(E::from(a.s), E::mul(v, E::ld(a.tab, a.tab_plane, kaxis))); break;
                case OP_MUL_POW: {
                    const V mv = a.tab ? E::ld(a.tab, a.tab_plane, 0) : E::from(a.s);  // m in memory, or by value
                    V f = E::one();
                    for (unsigned i = 0; i < kaxis; ++i) f = E::mul(f, mv);
                    v = E::mul(v, f);
                    break;
                }
                default: break;
            }
        }
        E::st(out, out_plane, lin, v);
    }
}

// f64 fast path: the innermost (collapsed) axis is unit-stride, unshifted, fully valid and even, all outer
// strides are even and both bases 16-byte aligned => every thread moves one double2 (global_load_dwordx4).
struct NoTab {};
template <class TAB>
__global__ void __launch_bounds__(256) k_gather_f64x2(const double* __restrict__ src, double* __restrict__ out,
                                                      GatherArgs a, TAB t, size_t total_pairs) {
    const int last = a.out.nd - 1;
    const unsigned half = a.out.d[last] >> 1;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total_pairs;
         lin += (size_t)gridDim.x * blockDim.x) {
        size_t r = lin / half;
        unsigned kp = (unsigned)(lin - r * half);
        size_t soff = 2 * (size_t)kp;
        bool valid = true;
        unsigned kaxis = 0;
#pragma unroll 1
        for (int ax = last - 1; ax >= 0; --ax) {
            unsigned d = a.out.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            long long si = (long long)k + a.shift[ax];
            if (si < 0 || si >= (long long)a.src_len[ax]) valid = false;
            soff += (size_t)(si < 0 ? 0 : si) * a.src_stride[ax];
            if (ax == a.tab_axis) kaxis = k;
        }
        if (valid && a.keep && !a.keep[kaxis]) valid = false;
        double2 v = make_double2(0.0, 0.0);
        if (valid) {
            v = *reinterpret_cast<const double2*>(src + soff);
            switch (a.op) {
                case OP_MUL_S: v.x = v.x * a.s.a; v.y = v.y * a.s.a; break;
                case OP_DIV_S: v.x = v.x / a.s.a; v.y = v.y / a.s.a; break;
                case OP_LMUL_S: v.x = a.s.a * v.x; v.y = a.s.a * v.y; break;
                case OP_NEG: v.x = -v.x; v.y = -v.y; break;
                case OP_MUL_TAB: { double f = a.tab[kaxis]; v.x = v.x * f; v.y = v.y * f; break; }
                case OP_MUL_HTAB:
                    if constexpr (sizeof(TAB) > 1) { double f = t.v[kaxis]; v.x = v.x * f; v.y = v.y * f; }
                    break;
                case OP_MUL_TAB_LMUL_S: { double f = a.tab[kaxis]; v.x = a.s.a * (v.x * f); v.y = a.s.a * (v.y * f); break; }
                case OP_MUL_POW: {
                    const double m = a.tab ? a.tab[0] : a.s.a;
                    double f = 1.0;
                    for (unsigned i = 0; i < kaxis; ++i) f = f * m;
                    v.x = v.x * f;
                    v.y = v.y * f;
                    break;
                }
                default: break;
            }
        }
        *reinterpret_cast<double2*>(out + 2 * lin) = v;
    }
}

// k_gather with the per-slab factors passed BY VALUE (OP_MUL_HTAB): x * t[k_axis]
template <class E>
__global__ void __launch_bounds__(256) k_gather_htab(const double* __restrict__ src, size_t src_plane,
                                                     double* __restrict__ out, size_t out_plane, GatherArgs a, HostTab t,
                                                     size_t total) {
    typedef typename E::V V;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        size_t r = lin;
        size_t soff = 0;
        bool valid = true;
        unsigned kaxis = 0;
#pragma unroll 1
        for (int ax = a.out.nd - 1; ax >= 0; --ax) {
            unsigned d = a.out.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            long long si = (long long)k + a.shift[ax];
            if (si < 0 || si >= (long long)a.src_len[ax]) valid = false;
            soff += (size_t)(si < 0 ? 0 : si) * a.src_stride[ax];
            if (ax == a.tab_axis) kaxis = k;
        }
        V v = E::zero();
        if (valid && a.keep && !a.keep[kaxis]) valid = false;
        if (valid) v = E::mul(E::ld(src, src_plane, soff), E::ld(t.v, HTAB_CAP / 2, kaxis));
        E::st(out, out_plane, lin, v);
    }
}

// Row-oriented gather: one WAVE per output row (all axes but the last), lanes along the row.  The row's multi-index,
// its validity and its source offset are wave-uniform and formed once per row; the lanes only add their position on the
// last axis — no per-element division (the generic kernel spends ~100 instructions per axis and element on the 64-bit
// index decomposition, which shows as soon as the 16-byte fast path does not apply: rows shifted by one element along
// the last axis, interval tensors).  Same per-element operations as k_gather.
template <class E>
__global__ void __launch_bounds__(256) k_gather_rows(const double* __restrict__ src, size_t src_plane,
                                                     double* __restrict__ out, size_t out_plane, GatherArgs a, size_t rows) {
    typedef typename E::V V;
    const int last = a.out.nd - 1;
    const unsigned inner = a.out.d[last];
    const unsigned lane = threadIdx.x & 63u;
    const size_t wave0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    const long long shl = a.shift[last], lenl = a.src_len[last];
    const size_t strl = a.src_stride[last];
    const bool tab_last = a.tab_axis == last;
    for (size_t row = wave0; row < rows; row += nwaves) {
        size_t r = row, soff = 0;
        bool valid_row = true;
        unsigned kaxis_row = 0;
#pragma unroll 1
        for (int ax = last - 1; ax >= 0; --ax) {
            unsigned d = a.out.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            long long si = (long long)k + a.shift[ax];
            if (si < 0 || si >= (long long)a.src_len[ax]) valid_row = false;
            soff += (size_t)(si < 0 ? 0 : si) * a.src_stride[ax];
            if (ax == a.tab_axis) kaxis_row = k;
        }
        if (valid_row && a.keep && !tab_last && !a.keep[kaxis_row]) valid_row = false;
        const size_t obase = row * inner;
        for (unsigned k = lane; k < inner; k += 64) {
            const long long si = (long long)k + shl;
            const unsigned kaxis = tab_last ? k : kaxis_row;
            bool valid = valid_row && si >= 0 && si < lenl;
            if (valid && a.keep && tab_last && !a.keep[kaxis]) valid = false;
            V v = E::zero();
            if (valid) {
                v = E::ld(src, src_plane, soff + (size_t)si * strl);
                switch (a.op) {
                    case OP_MUL_S: v = E::mul(v, E::from(a.s)); break;
                    case OP_DIV_S: v = E::div(v, E::from(a.s)); break;
                    case OP_LMUL_S: v = E::mul(E::from(a.s), v); break;
                    case OP_NEG: v = E::neg(v); break;
                    case OP_MUL_TAB: v = E::mul(v, E::ld(a.tab, a.tab_plane, kaxis)); break;
                    case OP_MUL_TAB_LMUL_S: v = E::mul(E::from(a.s), E::mul(v, E::ld(a.tab, a.tab_plane, kaxis))); break;
                    case OP_MUL_POW: {
                        const V mv = a.tab ? E::ld(a.tab, a.tab_plane, 0) : E::from(a.s);
                        V f = E::one();
                        for (unsigned i = 0; i < kaxis; ++i) f = E::mul(f, mv);
                        v = E::mul(v, f);
                        break;
                    }
                    default: break;
                }
            }
            E::st(out, out_plane, obase + k, v);
        }
    }
}

template <class E>
void K<E>::gather(hipStream_t st, const double* src, size_t src_plane, double* out, size_t out_plane,
                  const GatherArgs& a) {
    size_t total = 1;
    for (int i = 0; i < a.out.nd; ++i) total *= a.out.d[i];
    if (total == 0) return;
    HostTab t;
    GatherArgs b = a;
    if (a.op == OP_MUL_HTAB) {  // a.tab is a HOST array here: [plane][tab_plane] -> by-value kernel argument
        std::memset(&t, 0, sizeof(t));
        const unsigned len = a.tab_axis >= 0 ? a.out.d[a.tab_axis] : 0;
        for (unsigned pl = 0; pl < (unsigned)E::W; ++pl)
            std::memcpy(t.v + pl * (HTAB_CAP / 2), a.tab + pl * a.tab_plane, sizeof(double) * len);
        b.tab = nullptr;
    }
    if (E::W == 1 && a.out.nd >= 1) {
        const int last = a.out.nd - 1;
        bool ok = a.src_stride[last] == 1 && a.shift[last] == 0 && a.out.d[last] <= a.src_len[last] &&
                  (a.out.d[last] & 1u) == 0 && a.tab_axis != last && (((uintptr_t)src | (uintptr_t)out) & 15) == 0;
        for (int i = 0; i < last && ok; ++i)
            if (a.src_stride[i] & 1) ok = false;
        if (ok) {
            size_t pairs = total / 2;
            if (a.op == OP_MUL_HTAB)
                GFT_LAUNCH(k_gather_f64x2<HostTab>, dim3(grid_for(pairs)), dim3(256), 0, st, src, out, b, t, pairs);
            else
                GFT_LAUNCH(k_gather_f64x2<NoTab>, dim3(grid_for(pairs)), dim3(256), 0, st, src, out, a, NoTab{}, pairs);
            return;
        }
    }
    if (a.op == OP_MUL_HTAB) {
        GFT_LAUNCH(k_gather_htab<E>, dim3(grid_for(total)), dim3(256), 0, st, src, src_plane, out, out_plane, b, t, total);
        return;
    }
    // rows of at least a wave's width that missed the 16-byte path: one wave per row (no per-element index arithmetic)
    static const bool rows_on = true;
    // (only where bandwidth is the issue: on a 180 x 180 tensor one wave per row is 180 waves walking their rows serially
    // where the per-element kernel has 32 000 threads in flight — mixture --bounds 3.0 -> 3.7 s when it was unconditional)
    // (and only with rows enough to fill the chip: four_populations gathers 1e6 elements in a few dozen rows of tens of
    // thousands — 8 workgroups walking them took 337 us where the per-element kernel takes 19)
    if (rows_on && a.out.nd >= 2 && a.out.d[a.out.nd - 1] >= 48 && total >= ((size_t)1 << 20) && total / a.out.d[a.out.nd - 1] >= 4096) {
        const size_t rows = total / a.out.d[a.out.nd - 1];
        const size_t blocks = std::min<size_t>((rows + 3) / 4, 256 * 16);
        GFT_LAUNCH(k_gather_rows<E>, dim3((unsigned)blocks), dim3(256), 0, st, src, src_plane, out, out_plane, a, rows);
        return;
    }
    GFT_LAUNCH(k_gather<E>, dim3(grid_for(total)), dim3(256), 0, st, src, src_plane, out, out_plane, a,
                       total);
}

// ------------------------------------------------------------------------------------------
// add / sub of two leading blocks into a zero tensor (mt:873-880, 927-934)
// ------------------------------------------------------------------------------------------
template <class E>
__global__ void __launch_bounds__(256) k_addsub_padded(DView out, DView a, DView b, int subtract, size_t total) {
    typedef typename E::V V;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        size_t r = lin, aoff = 0, boff = 0, astr = 1, bstr = 1;
        bool ina = true, inb = true;
#pragma unroll 1
        for (int ax = out.sh.nd - 1; ax >= 0; --ax) {
            unsigned d = out.sh.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            if (k >= a.sh.d[ax]) ina = false;
            if (k >= b.sh.d[ax]) inb = false;
            aoff += k * astr;
            boff += k * bstr;
            astr *= a.sh.d[ax];
            bstr *= b.sh.d[ax];
        }
        V v = E::zero();
        if (ina) v = E::add(v, E::ld(a.p, a.plane, aoff));
        if (inb) {
            V w = E::ld(b.p, b.plane, boff);
            v = subtract ? E::sub(v, w) : E::add(v, w);
        }
        E::st(out.p, out.plane, lin, v);
    }
}

template <class E>
__global__ void __launch_bounds__(256) k_add_scaled_padded(DView out, DView a, DView b, Scalar2 c, size_t total) {
    typedef typename E::V V;
    const V cv = E::from(c);
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        size_t r = lin, aoff = 0, boff = 0, astr = 1, bstr = 1;
        bool ina = true, inb = true;
#pragma unroll 1
        for (int ax = out.sh.nd - 1; ax >= 0; --ax) {
            unsigned d = out.sh.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            if (k >= a.sh.d[ax]) ina = false;
            if (k >= b.sh.d[ax]) inb = false;
            aoff += k * astr;
            boff += k * bstr;
            astr *= a.sh.d[ax];
            bstr *= b.sh.d[ax];
        }
        V v = E::zero();
        if (ina) v = E::add(v, E::ld(a.p, a.plane, aoff));
        if (inb) v = E::add(v, E::mul(cv, E::ld(b.p, b.plane, boff)));
        E::st(out.p, out.plane, lin, v);
    }
}
template <class E>
void K<E>::add_scaled_padded(hipStream_t st, const DView& out, const DView& a, const DView& b, Scalar2 c) {
    size_t total = 1;
    for (int i = 0; i < out.sh.nd; ++i) total *= out.sh.d[i];
    if (total == 0) return;
    GFT_LAUNCH(k_add_scaled_padded<E>, dim3(grid_for(total)), dim3(256), 0, st, out, a, b, c, total);
}

// Equal shapes (the common case): 1-D, two f64 per thread.  Same per-element order: (0 + a) (+|-) b.
__global__ void __launch_bounds__(256) k_addsub_f64x2(const double* __restrict__ a, const double* __restrict__ b,
                                                      double* __restrict__ out, size_t pairs, int subtract) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
        double2 x = reinterpret_cast<const double2*>(a)[i], y = reinterpret_cast<const double2*>(b)[i], r;
        r.x = 0.0 + x.x;
        r.y = 0.0 + x.y;
        r.x = subtract ? r.x - y.x : r.x + y.x;
        r.y = subtract ? r.y - y.y : r.y + y.y;
        reinterpret_cast<double2*>(out)[i] = r;
    }
}

template <class E>
void K<E>::addsub_padded(hipStream_t st, const DView& out, const DView& a, const DView& b, int subtract) {
    size_t total = 1;
    for (int i = 0; i < out.sh.nd; ++i) total *= out.sh.d[i];
    if (total == 0) return;
    bool same = E::W == 1 && (total & 1) == 0 && (((uintptr_t)out.p | (uintptr_t)a.p | (uintptr_t)b.p) & 15) == 0;
    for (int i = 0; i < out.sh.nd && same; ++i)
        if (a.sh.d[i] != out.sh.d[i] || b.sh.d[i] != out.sh.d[i]) same = false;
    if (same) {
        GFT_LAUNCH(k_addsub_f64x2, dim3(grid_for(total / 2)), dim3(256), 0, st, a.p, b.p, out.p, total / 2, subtract);
        return;
    }
    GFT_LAUNCH(k_addsub_padded<E>, dim3(grid_for(total)), dim3(256), 0, st, out, a, b, subtract, total);
}

// ------------------------------------------------------------------------------------------
// deferred elementwise chains (gft_kernels.hpp ChainSrc): materialise one, or add / subtract two on the fly
// ------------------------------------------------------------------------------------------
template <class E>
__device__ __forceinline__ typename E::V chain_apply(const ChainStage* st, int nstages, const int* pad, typename E::V x, const unsigned* k, bool first);
template <class E>
__device__ __forceinline__ typename E::V chain_eval(const ChainSrc& c, size_t off, const unsigned* k, bool first) {
    return chain_apply<E>(c.st, c.nstages, c.pad, E::ld(c.p, c.plane, off), k, first);
}
// the recorded stages on a value that is already in a register (`pad`: the view's front pad per axis, or null)
template <class E>
__device__ __forceinline__ typename E::V chain_apply(const ChainStage* st, int nstages, const int* pad, typename E::V x, const unsigned* k, bool first) {
#pragma unroll 1
    for (int i = 0; i < nstages; ++i) {
        const ChainStage& g = st[i];
        switch (g.kind) {
            case CH_LMUL_S: x = E::mul(E::from(g.s), x); break;
            case CH_MUL_S: x = E::mul(x, E::from(g.s)); break;
            case CH_DIV_S: x = E::div(x, E::from(g.s)); break;
            case CH_NEG: x = E::neg(x); break;
            case CH_FIRST_ADD: if (first) x = E::add(x, E::from(g.s)); break;
            case CH_FIRST_SUB: if (first) x = E::sub(x, E::from(g.s)); break;
            case CH_FIRST_SUB_NEG_ALL:
                if (first) x = E::sub(x, E::from(g.s));
                x = E::neg(x);
                break;
            case CH_MUL_TAB: x = E::mul(x, E::ld(g.tab, g.tab_plane, k[g.axis] - (pad ? (unsigned)pad[g.axis] : 0u))); break;  // (the view's own coordinate)
            default: break;
        }
    }
    return x;
}

// IDX: the type of the per-element odometer.  `unsigned` whenever the output and the base tensors have fewer than 2^31 elements
// (always, in practice): 64-bit divisions by a run-time divisor are ~150 instructions each, and on the 10^4-10^5 element
// tensors of the NeurIPS programs — one element per thread, one wave per SIMD — this kernel's duration IS its instruction
// count (mixture: 12 000 launches of 5.7 us, hmm: 56 % of the kernel time).
// A workgroup's copy of the kernel-argument segment in LDS (see k_chain_nest)
template <class KA>
__device__ __forceinline__ const KA& kernargs_to_lds(unsigned char* lds) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef const v4u __attribute__((address_space(4))) * seg_t;
    const seg_t src = (seg_t)__builtin_amdgcn_kernarg_segment_ptr();
    for (unsigned i = threadIdx.x; i < (sizeof(KA) + 15) / 16; i += blockDim.x) {
        const v4u t = src[i];
        reinterpret_cast<v4u*>(lds)[i] = t;
    }
    __syncthreads();
    return *reinterpret_cast<const KA*>(lds);
}
template <class E, bool TWO, typename IDX>
__device__ __forceinline__ void chain_body(double* __restrict__ out, size_t out_plane, const Shape& sh, const ChainSrc& a, const ChainSrc& b, int subtract,
                                           size_t total) {
    typedef typename E::V V;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total; lin += (size_t)gridDim.x * blockDim.x) {
        IDX r = (IDX)lin, aoff = 0, boff = 0;
        bool ina = true, inb = true;
        // positional stages (FIRST_*, MUL_TAB) look at the coordinates of the chain's OWN view — what lies under a front
        // pad is the view shifted, its element 0 is the output's element `pad`
        bool firsta = true, firstb = true;
        unsigned k[MAXD];
#pragma unroll 1
        for (int ax = sh.nd - 1; ax >= 0; --ax) {
            const unsigned d = sh.d[ax];
            const unsigned kk = (unsigned)(r % d);
            r /= d;
            k[ax] = kk;
            const unsigned ka = kk - (unsigned)a.pad[ax];  // (wraps below the pad: fails the box test)
            if (ka >= a.box[ax]) ina = false;
            if (ka != 0) firsta = false;
            aoff += (IDX)ka * (IDX)a.stride[ax];
            if (TWO) {
                const unsigned kb = kk - (unsigned)b.pad[ax];
                if (kb >= b.box[ax]) inb = false;
                if (kb != 0) firstb = false;
                boff += (IDX)kb * (IDX)b.stride[ax];
            }
        }
        V v;
        if (!TWO) {
            v = ina ? chain_eval<E>(a, (size_t)aoff, k, firsta) : E::zero();
        } else {
            v = E::zero();
            if (ina) v = E::add(v, chain_eval<E>(a, (size_t)aoff, k, firsta));
            if (inb) {
                V w = chain_eval<E>(b, (size_t)boff, k, firstb);
                v = subtract ? E::sub(v, w) : E::add(v, w);
            }
        }
        E::st(out, out_plane, lin, v);
    }
}
// the arguments (two chains: 0.9 KB) are copied to LDS first (see k_chain_nest)
struct ChainKArgs {
    double* out;
    size_t out_plane;
    Shape sh;
    ChainSrc a, b;
    int subtract;
    size_t total;
};
template <class E, bool TWO, typename IDX>
__global__ void __launch_bounds__(256) k_chain_lds(ChainKArgs) {
    __shared__ __align__(16) unsigned char s_args[(sizeof(ChainKArgs) + 15) / 16 * 16];
    const ChainKArgs& A = kernargs_to_lds<ChainKArgs>(s_args);
    chain_body<E, TWO, IDX>(A.out, A.out_plane, A.sh, A.a, A.b, A.subtract, A.total);
}
template <class E, bool TWO, typename IDX>
static void launch_chain(hipStream_t st, double* out, size_t out_plane, const Shape& sh, const ChainSrc& a, const ChainSrc& b, int subtract, size_t total) {
    ChainKArgs ka;
    ka.out = out;
    ka.out_plane = out_plane;
    ka.sh = sh;
    ka.a = a;
    ka.b = b;
    ka.subtract = subtract;
    ka.total = total;
    GFT_LAUNCH((k_chain_lds<E, TWO, IDX>), dim3(grid_for(total)), dim3(256), 0, st, ka);
}
// (offsets of lanes outside an operand's box may wrap in 32 bits: they are never dereferenced)
static bool chain_fits_u32(const ChainSrc& c, const Shape& sh, size_t total) {
    if (total >= 0x7fffffffull) return false;
    unsigned long long span = 1;
    for (int ax = 0; ax < sh.nd; ++ax) span += (unsigned long long)(c.box[ax] ? c.box[ax] - 1 : 0) * c.stride[ax];
    return span < 0x7fffffffull;
}
// The same when every operand is its whole base tensor in the output's own layout (no sub-box, no pad, no table stage): the
// element index IS the offset — no per-element odometer (64-bit divisions per axis: the general kernel reached 33 % of the
// HBM roof at 384^3, tools/bench_streaming.py), two elements per thread and iteration in flight.
template <class E, bool TWO>
__device__ __forceinline__ void chain_flat_body(double* __restrict__ out, size_t out_plane, const ChainSrc& a, const ChainSrc& b, int subtract, size_t total) {
    typedef typename E::V V;
    const unsigned k0[MAXD] = {0};
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total; lin += 2 * step) {
        const size_t lin2 = lin + step;
        const bool two = lin2 < total;
        V v0, v1 = E::zero();
        if (!TWO) {
            v0 = chain_eval<E>(a, lin, k0, lin == 0);
            if (two) v1 = chain_eval<E>(a, lin2, k0, false);
        } else {
            const V a0 = chain_eval<E>(a, lin, k0, lin == 0), b0 = chain_eval<E>(b, lin, k0, lin == 0);
            V a1 = E::zero(), b1 = E::zero();
            if (two) {
                a1 = chain_eval<E>(a, lin2, k0, false);
                b1 = chain_eval<E>(b, lin2, k0, false);
            }
            v0 = E::add(E::zero(), a0);
            v0 = subtract ? E::sub(v0, b0) : E::add(v0, b0);
            v1 = E::add(E::zero(), a1);
            v1 = subtract ? E::sub(v1, b1) : E::add(v1, b1);
        }
        E::st(out, out_plane, lin, v0);
        if (two) E::st(out, out_plane, lin2, v1);
    }
}
template <class E, bool TWO>
__global__ void __launch_bounds__(256) k_chain_flat(double* __restrict__ out, size_t out_plane, ChainSrc a, ChainSrc b, int subtract, size_t total) {
    chain_flat_body<E, TWO>(out, out_plane, a, b, subtract, total);
}
// the operand is its whole base tensor, laid out like the output, and none of its stages looks at coordinates beyond "element 0"
static bool chain_is_flat(const ChainSrc& c, const Shape& sh) {
    size_t stride = 1;
    for (int ax = sh.nd - 1; ax >= 0; --ax) {
        if (c.pad[ax] != 0 || c.box[ax] != sh.d[ax] || c.stride[ax] != stride) return false;
        stride *= sh.d[ax];
    }
    for (int i = 0; i < c.nstages; ++i)
        if (c.st[i].kind == CH_MUL_TAB) return false;
    return true;
}

// Arrival of a scan's workgroup (thread 0, after its verdict is in state[0]): true for the LAST workgroup of the launch.
// Two levels — groups of 32 workgroups count on their own word (128 bytes apart, behind the state words: state + 248 +
// 32 g), a group's last arriver counts on state[1] — because same-address atomics serialise at ~26 ns each: one counter for
// the 329 workgroups of a 290^2 tensor was 8 us of a 6 us kernel's tail, for the 2048 of a 100^3 tensor 50 us.
// INVARIANT (what the last workgroup may read, and who fences it): after the last arrival the publishing workgroup reads back
// ONLY state[0] (atomics: coherent by themselves) and, in k_chain_scan / k_linear_scan, element 0 and the unit positions of the
// tensor — and exactly the WRITERS of those elements fence their stores (device scope) before their workgroup's barrier and
// arrival.  The __threadfence() below orders thread 0's own view (its workgroup's atomics before its arrival); it is NOT a
// licence to read other elements another XCD wrote: a read-back of anything else needs its writer to fence first
// (__syncthreads alone does not wait for vmcnt outside tgsplit mode).
__device__ __forceinline__ bool scan_arrive(unsigned* state) {
    constexpr unsigned G = 32;
    const unsigned group = blockIdx.x / G, ngroups = (gridDim.x + G - 1) / G;
    const unsigned gsize = group + 1 < ngroups ? G : gridDim.x - group * G;
    unsigned* gc = state + 248 + 32 * (group & 63u);  // (at most 2048 workgroups = 64 groups per launch)
    __threadfence();
    if (atomicAdd(gc, 1u) != gsize - 1) return false;
    atomicExch(gc, 0u);  // everyone of the group has arrived: ready for the next launch on this stream
    __threadfence();
    return atomicAdd(&state[1], 1u) == ngroups - 1;
}

// k_chain<E, false> and k_linear_scan in one launch: the accumulator of a Horner step is a deferred chain whose FIRST
// consumer is Mul's `extract_linear` (mt:1014-1072 asks `self` first) — materialise it and settle the question in the same
// pass (one launch and its gap less per subst_var of a `--bounds` program).  Verdict and mailbox as in k_linear_scan.
template <class E, typename IDX>
__global__ void __launch_bounds__(256) k_chain_scan(double* __restrict__ out, size_t out_plane, Shape sh, ChainSrc a, unsigned axes_mask,
                                                    unsigned* state, Mailbox mb, size_t total) {
    typedef typename E::V V;
    unsigned local = axes_mask;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total; lin += (size_t)gridDim.x * blockDim.x) {
        IDX r = (IDX)lin, aoff = 0;
        bool ina = true, firsta = true;
        unsigned k[MAXD];
        int nonzero_axes = 0, which = -1;
        bool unit = true;
#pragma unroll 1
        for (int ax = sh.nd - 1; ax >= 0; --ax) {
            const unsigned d = sh.d[ax];
            const unsigned kk = (unsigned)(r % d);
            r /= d;
            k[ax] = kk;
            if (kk != 0) {
                nonzero_axes++;
                which = ax;
                if (kk != 1) unit = false;
            }
            const unsigned ka = kk - (unsigned)a.pad[ax];
            if (ka >= a.box[ax]) ina = false;
            if (ka != 0) firsta = false;
            aoff += (IDX)ka * (IDX)a.stride[ax];
        }
        const V v = ina ? chain_eval<E>(a, (size_t)aoff, k, firsta) : E::zero();
        E::st(out, out_plane, lin, v);
        // the last workgroup reads back element 0 and the element at 1 along the linear axis: only THEIR writers make
        // their store visible device-wide (a fence is a write-back of the XCD's whole L2 — every thread paid one)
        if (nonzero_axes == 0 || (nonzero_axes == 1 && unit)) __threadfence();
        if (local != 0 && nonzero_axes != 0 && !E::is_zero(v)) {
            if (nonzero_axes == 1 && unit) local &= (1u << which);
            else local = 0;
        }
    }
    for (int off = 32; off > 0; off >>= 1) local &= __shfl_xor(local, off, 64);
    __shared__ unsigned s_and[4];
    __shared__ unsigned s_last;
    if ((threadIdx.x & 63) == 0) s_and[threadIdx.x >> 6] = local;
    __syncthreads();  // (also: the fenced stores of this workgroup's threads precede its arrival)
    if (threadIdx.x == 0) {
        unsigned blk = s_and[0] & s_and[1] & s_and[2] & s_and[3];
        unsigned cur = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur & ~blk) atomicAnd(&state[0], blk);
        s_last = scan_arrive(state) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    double* res = mb.payload;
    unsigned m = atomicAnd(&state[0], 0xffffffffu);  // device-scope read of the combined mask
    res[0] = (double)m;
    res[1] = res[2] = res[3] = res[4] = 0.0;
    if (m) {
        int ax = __ffs((int)m) - 1;
        size_t stride = 1;
        for (int i = sh.nd - 1; i > ax; --i) stride *= sh.d[i];
        res[1] = __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        res[3] = __hip_atomic_load(out + stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (E::W == 2) {
            res[2] = __hip_atomic_load(out + out_plane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            res[4] = __hip_atomic_load(out + out_plane + stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    atomicExch(&state[0], 0xffffffffu);  // restore for the next call on this stream
    atomicExch(&state[1], 0u);
    if (mb.dev_word) *mb.dev_word = m;  // (read by kernels queued behind this one: HornerLoopArgs::guard)
    mailbox_publish(mb);
}
template <class E>
void K<E>::chain_copy_scan(hipStream_t st, double* out, size_t out_plane, const Shape& sh, const ChainSrc& a, unsigned axes_mask,
                           unsigned* state, const Mailbox& mb) {
    size_t total = 1;
    for (int i = 0; i < sh.nd; ++i) total *= sh.d[i];
    if (total == 0) return;
    if (chain_fits_u32(a, sh, total)) GFT_LAUNCH((k_chain_scan<E, unsigned>), dim3(grid_for(total)), dim3(256), 0, st, out, out_plane, sh, a, axes_mask, state, mb, total);
    else GFT_LAUNCH((k_chain_scan<E, size_t>), dim3(grid_for(total)), dim3(256), 0, st, out, out_plane, sh, a, axes_mask, state, mb, total);
}
template <class E>
void K<E>::chain_copy(hipStream_t st, double* out, size_t out_plane, const Shape& sh, const ChainSrc& a) {
    size_t total = 1;
    for (int i = 0; i < sh.nd; ++i) total *= sh.d[i];
    if (total == 0) return;
    if (total >= 4096 && chain_is_flat(a, sh)) {
        GFT_LAUNCH((k_chain_flat<E, false>), dim3(grid_for((total + 1) / 2)), dim3(256), 0, st, out, out_plane, a, a, 0, total);
        return;
    }
    if (chain_fits_u32(a, sh, total)) launch_chain<E, false, unsigned>(st, out, out_plane, sh, a, a, 0, total);
    else launch_chain<E, false, size_t>(st, out, out_plane, sh, a, a, 0, total);
}
template <class E>
void K<E>::chain_addsub(hipStream_t st, double* out, size_t out_plane, const Shape& sh, const ChainSrc& a, const ChainSrc& b,
                        int subtract) {
    size_t total = 1;
    for (int i = 0; i < sh.nd; ++i) total *= sh.d[i];
    if (total == 0) return;
    if (total >= 4096 && chain_is_flat(a, sh) && chain_is_flat(b, sh)) {
        GFT_LAUNCH((k_chain_flat<E, true>), dim3(grid_for((total + 1) / 2)), dim3(256), 0, st, out, out_plane, a, b, subtract, total);
        return;
    }
    if (chain_fits_u32(a, sh, total) && chain_fits_u32(b, sh, total))
        launch_chain<E, true, unsigned>(st, out, out_plane, sh, a, b, subtract, total);
    else launch_chain<E, true, size_t>(st, out, out_plane, sh, a, b, subtract, total);
}

// ---- batches of chain launches (gft_kernels.hpp ChainItem): variant = 4 * two + {0: flat, 1: 32-bit odometer, 2: 64-bit} ----
template <class E, bool TWO, int MODE>
__global__ void __launch_bounds__(256) k_chain_batch(const ChainItem* __restrict__ items) {
    __shared__ __align__(16) unsigned char s_args[sizeof(ChainItem)];
    const ChainItem& A = item_to_lds<ChainItem>(items, s_args);
    if (MODE == 0) chain_flat_body<E, TWO>(A.out, A.out_plane, A.a, A.b, A.subtract, A.total);
    else if (MODE == 1) chain_body<E, TWO, unsigned>(A.out, A.out_plane, A.sh, A.a, A.b, A.subtract, A.total);
    else chain_body<E, TWO, size_t>(A.out, A.out_plane, A.sh, A.a, A.b, A.subtract, A.total);
}
template <class E>
typename K<E>::Geometry K<E>::chain_geometry(const ChainItem& it) {
    Geometry g;
    const bool two = it.two != 0;
    int mode;
    if (it.total >= 4096 && chain_is_flat(it.a, it.sh) && (!two || chain_is_flat(it.b, it.sh))) mode = 0;
    else if (chain_fits_u32(it.a, it.sh, it.total) && (!two || chain_fits_u32(it.b, it.sh, it.total))) mode = 1;
    else mode = 2;
    g.variant = (two ? 4 : 0) + mode;
    g.gx = grid_for(mode == 0 ? (it.total + 1) / 2 : it.total);
    g.threads = 256;
    g.lds = 0;
    g.ok = it.total != 0;
    return g;
}
template <class E>
void K<E>::chain_batch(hipStream_t st, const ChainItem* items, unsigned n, const Geometry& g) {
    const dim3 grid(g.gx, n), block(256);
    switch (g.variant) {
        case 0: GFT_LAUNCH((k_chain_batch<E, false, 0>), grid, block, 0, st, items); break;
        case 1: GFT_LAUNCH((k_chain_batch<E, false, 1>), grid, block, 0, st, items); break;
        case 2: GFT_LAUNCH((k_chain_batch<E, false, 2>), grid, block, 0, st, items); break;
        case 4: GFT_LAUNCH((k_chain_batch<E, true, 0>), grid, block, 0, st, items); break;
        case 5: GFT_LAUNCH((k_chain_batch<E, true, 1>), grid, block, 0, st, items); break;
        default: GFT_LAUNCH((k_chain_batch<E, true, 2>), grid, block, 0, st, items); break;
    }
}

// Nested chain add (NestSrc): out = (0 + A) (+|-) B where A / B are chains or recorded two-chain sums.
// Where a leaf is present at output index k, and where its element lies.
__device__ __forceinline__ bool nest_locate(const ChainSrc& c, const unsigned* k, int nd, unsigned& off, bool& first) {
    bool in = true;
    first = true;
    off = 0;
#pragma unroll 1
    for (int ax = 0; ax < nd; ++ax) {
        const unsigned ka = k[ax] - (unsigned)c.pad[ax];  // (wraps below the pad: fails the box test)
        if (ka >= c.box[ax]) in = false;
        if (ka != 0) first = false;
        off += ka * (unsigned)c.stride[ax];
    }
    return in;
}
// The arguments are 2.5 KB.  Read from the kernel-argument segment where they are used they are scalar loads inside
// run-time loops — one dependent miss of the (cold, per-CU) scalar cache after the other, for the first wave of every
// workgroup: most of this launch's 8 us.  So the workgroup copies the whole segment to LDS first — one 16-byte vector load
// per thread, all in flight together — and reads its arguments from there.
struct NestKArgs {
    double* out;
    size_t out_plane;
    Shape sh;
    NestSrc a, b;
    int subtract;
    unsigned total;
};
template <class E>
__device__ __forceinline__ void chain_nest_body(double* __restrict__ out, size_t out_plane, const Shape& sh, const NestSrc& a, const NestSrc& b, int subtract,
                                                unsigned total) {
    typedef typename E::V V;
    for (unsigned lin = blockIdx.x * 256u + threadIdx.x; lin < total; lin += gridDim.x * 256u) {
        unsigned k[MAXD], r = lin;
        bool all0 = true, in_abox = true, in_bbox = true;
#pragma unroll 1
        for (int ax = sh.nd - 1; ax >= 0; --ax) {
            const unsigned d = sh.d[ax];
            k[ax] = r % d;
            r /= d;
            if (k[ax]) all0 = false;
            if (k[ax] >= a.box[ax]) in_abox = false;
            if (k[ax] >= b.box[ax]) in_bbox = false;
        }
        // the (up to) four leaves: located first, then ALL their elements requested before any is used — a launch of this
        // kind is a few dependent memory latencies long, and four loads one behind the other were most of its 8 us
        unsigned o0, o1, o2, o3;
        bool f0, f1, f2, f3;
        const bool i0 = nest_locate(a.a, k, sh.nd, o0, f0) && (!a.nested || in_abox);
        const bool i1 = a.nested && in_abox && nest_locate(a.b, k, sh.nd, o1, f1);
        const bool i2 = nest_locate(b.a, k, sh.nd, o2, f2) && (!b.nested || in_bbox);
        const bool i3 = b.nested && in_bbox && nest_locate(b.b, k, sh.nd, o3, f3);
        const V r0 = E::ld(a.a.p, a.a.plane, i0 ? o0 : 0u), r1 = E::ld(a.nested ? a.b.p : a.a.p, a.nested ? a.b.plane : a.a.plane, i1 ? o1 : 0u);
        const V r2 = E::ld(b.a.p, b.a.plane, i2 ? o2 : 0u), r3 = E::ld(b.nested ? b.b.p : b.a.p, b.nested ? b.b.plane : b.a.plane, i3 ? o3 : 0u);
        V v = E::zero();
        {   // operand A
            V t = E::zero();
            bool present;
            if (!a.nested) {
                present = i0;
                if (i0) t = chain_apply<E>(a.a.st, a.a.nstages, a.a.pad, r0, k, f0);
            } else {
                present = in_abox;
                if (i0) t = E::add(t, chain_apply<E>(a.a.st, a.a.nstages, a.a.pad, r0, k, f0));
                if (i1) {
                    const V w = chain_apply<E>(a.b.st, a.b.nstages, a.b.pad, r1, k, f1);
                    t = a.sub_inner ? E::sub(t, w) : E::add(t, w);
                }
                if (present) t = chain_apply<E>(a.post, a.npost, nullptr, t, k, all0);
            }
            if (present) v = E::add(v, t);
        }
        {   // operand B
            V t = E::zero();
            bool present;
            if (!b.nested) {
                present = i2;
                if (i2) t = chain_apply<E>(b.a.st, b.a.nstages, b.a.pad, r2, k, f2);
            } else {
                present = in_bbox;
                if (i2) t = E::add(t, chain_apply<E>(b.a.st, b.a.nstages, b.a.pad, r2, k, f2));
                if (i3) {
                    const V w = chain_apply<E>(b.b.st, b.b.nstages, b.b.pad, r3, k, f3);
                    t = b.sub_inner ? E::sub(t, w) : E::add(t, w);
                }
                if (present) t = chain_apply<E>(b.post, b.npost, nullptr, t, k, all0);
            }
            if (present) v = subtract ? E::sub(v, t) : E::add(v, t);
        }
        E::st(out, out_plane, lin, v);
    }
}
template <class E>
__global__ void __launch_bounds__(256) k_chain_nest(NestKArgs) {
    __shared__ __align__(16) unsigned char s_args[(sizeof(NestKArgs) + 15) / 16 * 16];
    const NestKArgs& A = kernargs_to_lds<NestKArgs>(s_args);
    chain_nest_body<E>(A.out, A.out_plane, A.sh, A.a, A.b, A.subtract, A.total);
}
template <class E>
void K<E>::chain_nest(hipStream_t st, double* out, size_t out_plane, const Shape& sh, const NestSrc& a, const NestSrc& b, int subtract) {
    size_t total = 1;
    for (int i = 0; i < sh.nd; ++i) total *= sh.d[i];
    if (total == 0) return;
    NestKArgs ka;
    ka.out = out;
    ka.out_plane = out_plane;
    ka.sh = sh;
    ka.a = a;
    ka.b = b;
    ka.subtract = subtract;
    ka.total = (unsigned)total;
    GFT_LAUNCH((k_chain_nest<E>), dim3(grid_for(total)), dim3(256), 0, st, ka);
}
template <class E>
__global__ void __launch_bounds__(256) k_chain_nest_batch(const NestItem* __restrict__ items) {
    __shared__ __align__(16) unsigned char s_args[sizeof(NestItem)];
    const NestItem& A = item_to_lds<NestItem>(items, s_args);
    chain_nest_body<E>(A.out, A.out_plane, A.sh, A.a, A.b, A.subtract, A.total);
}
template <class E>
typename K<E>::Geometry K<E>::chain_nest_geometry(size_t total) {
    Geometry g;
    g.gx = grid_for(total);
    g.threads = 256;
    g.ok = total != 0;
    return g;
}
template <class E>
void K<E>::chain_nest_batch(hipStream_t st, const NestItem* items, unsigned n, const Geometry& g) {
    GFT_LAUNCH((k_chain_nest_batch<E>), dim3(g.gx, n), dim3(256), 0, st, items);
}

// ------------------------------------------------------------------------------------------
// tiny scalar kernels
// ------------------------------------------------------------------------------------------
template <class E>
__global__ void __launch_bounds__(256) k_copy_first(const double* __restrict__ src, size_t sp, double* __restrict__ dst,
                                                    size_t dp, size_t n, int op, const double* s, size_t s_plane,
                                                    Scalar2 sv) {
    typedef typename E::V V;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V x = E::ld(src, sp, i);
        if (i == 0) {
            V y = s ? E::ld(s, s_plane, 0) : E::from(sv);
            x = (op == FIRST_ADD) ? E::add(x, y) : E::sub(x, y);
        }
        if (op == FIRST_SUB_NEG_ALL) x = E::neg(x);
        E::st(dst, dp, i, x);
    }
}
template <class E>
void K<E>::copy_first(hipStream_t st, const double* src, size_t src_plane, double* dst, size_t dst_plane, size_t n, int op,
                      const double* s, size_t s_plane, Scalar2 sv) {
    if (n == 0) return;
    GFT_LAUNCH(k_copy_first<E>, dim3(grid_for(n)), dim3(256), 0, st, src, src_plane, dst, dst_plane, n, op, s,
                       s_plane, sv);
}

template <class E>
__global__ void k_set_small(double* p, size_t plane, unsigned n, Scalar2 v0, Scalar2 v1) {
    E::st(p, plane, 0, E::from(v0));
    if (n > 1) E::st(p, plane, 1, E::from(v1));
}
template <class E>
void K<E>::set_small(hipStream_t st, double* p, size_t plane, unsigned n, Scalar2 v0, Scalar2 v1) {
    GFT_LAUNCH(k_set_small<E>, dim3(1), dim3(1), 0, st, p, plane, n, v0, v1);
}

template <class E>
__global__ void __launch_bounds__(256) k_linear_scan(DView t, unsigned axes_mask, unsigned* state, Mailbox mb, size_t total) {
    double* out = mb.payload;
    unsigned local = axes_mask;
    // (four loads in flight per thread: the loop's exit condition depends on what was loaded, so one load per iteration is one
    // memory latency per element — a tensor that IS linear, the case in which every element is read, ran at 7 % of the HBM roof)
    auto look = [&](typename E::V v, size_t lin) {
        if (E::is_zero(v)) return;
        size_t r = lin;
        int nonzero_axes = 0, which = -1;
        bool unit = true;
#pragma unroll 1
        for (int ax = t.sh.nd - 1; ax >= 0; --ax) {
            unsigned d = t.sh.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            if (k != 0) {
                nonzero_axes++;
                which = ax;
                if (k != 1) unit = false;
            }
        }
        if (nonzero_axes == 0) return;
        if (nonzero_axes == 1 && unit) local &= (1u << which);
        else local = 0;  // nothing more to learn: the loop ends
    };
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total && local != 0; lin += 4 * step) {
        // (unconditional loads on clamped indices: a load under a condition is a basic block of its own, and hipcc waits for
        // vmcnt(0) at every block boundary — the four loads would go out one after the other)
        const size_t l1 = lin + step, l2 = lin + 2 * step, l3 = lin + 3 * step, last = total - 1;
        const typename E::V v0 = E::ld(t.p, t.plane, lin);
        const typename E::V v1 = E::ld(t.p, t.plane, l1 < total ? l1 : last);
        const typename E::V v2 = E::ld(t.p, t.plane, l2 < total ? l2 : last);
        const typename E::V v3 = E::ld(t.p, t.plane, l3 < total ? l3 : last);
        look(v0, lin);
        look(v1, l1 < total ? l1 : last);
        look(v2, l2 < total ? l2 : last);
        look(v3, l3 < total ? l3 : last);
    }
    // block-level AND, then at most one device atomic per block — and none if the global verdict already
    // implies ours (a dense tensor is settled by the first block; ~1400 same-address atomics cost 20 us)
    for (int off = 32; off > 0; off >>= 1) local &= __shfl_xor(local, off, 64);
    __shared__ unsigned s_and[4];
    __shared__ unsigned s_last;
    if ((threadIdx.x & 63) == 0) s_and[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned blk = s_and[0] & s_and[1] & s_and[2] & s_and[3];
        unsigned cur = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur & ~blk) atomicAnd(&state[0], blk);
        s_last = scan_arrive(state) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    unsigned m = atomicAnd(&state[0], 0xffffffffu);  // device-scope read of the combined mask
    out[0] = (double)m;
    out[1] = out[2] = out[3] = out[4] = 0.0;
    if (m) {
        int ax = __ffs((int)m) - 1;
        size_t stride = 1;
        for (int i = t.sh.nd - 1; i > ax; --i) stride *= t.sh.d[i];
        out[1] = t.p[0];
        out[3] = t.p[stride];
        if (E::W == 2) {
            out[2] = t.p[t.plane];
            out[4] = t.p[t.plane + stride];
        }
    }
    atomicExch(&state[0], 0xffffffffu);  // restore for the next call on this stream
    atomicExch(&state[1], 0u);
    if (mb.dev_word) *mb.dev_word = m;  // (read by kernels queued behind this one: HornerLoopArgs::guard)
    mailbox_publish(mb);
}
template <class E>
void K<E>::linear_scan(hipStream_t st, const DView& t, unsigned axes_mask, unsigned* state, const Mailbox& mb) {
    size_t total = 1;
    for (int i = 0; i < t.sh.nd; ++i) total *= t.sh.d[i];
    if (total == 0) return;
    // few tickets (same-address atomics serialise at ~26 ns each; dense tensors exit at once) — but a tensor of hundreds of MB
    // that IS linear is read in full, and 128 blocks keep only 1 MB of loads in flight
    unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, total > ((size_t)1 << 22) ? 1024 : 128);
    GFT_LAUNCH(k_linear_scan<E>, dim3(blocks), dim3(256), 0, st, t, axes_mask, state, mb, total);
}

// The observation step's interval arithmetic, operation for operation the reference's (derivative * factor, * 1 inside
// mul_linear's mul_var, 0 + A, x * D, + B, * c), with two of its guaranteed short-circuits resolved where they stand —
//   X * [1,1]  is X, except that an exact zero interval of either sign becomes [+0,+0] (iv:164-190 tests for zero first),
//   a derivative factor of exactly 1 (the first coefficient's) is that same identity,
// so that the remaining products and sums can take the wave-checked general formulas (E::mulw / E::addw: ~80 instead
// of ~135 instructions each) in every wave that holds no OTHER special operand.  F64: the literal operations.
template <class E>
__device__ __forceinline__ typename E::V obs_mul_one(typename E::V x) {
    if constexpr (E::W == 1) return E::mul(x, E::one());
    else return E::is_zero(x) ? E::zero() : x;
}
template <class E>
__device__ __forceinline__ typename E::V obs_mul_tab(typename E::V x, typename E::V t) {
    if constexpr (E::W == 1) {
        return E::mul(x, t);
    } else {
        const bool one = E::is_one(t);
        const typename E::V r = E::mulw(x, one ? E::from_u32(2u) : t), i = obs_mul_one<E>(x);
        return Iv{one ? i.lo : r.lo, one ? i.hi : r.hi};
    }
}

template <class E, typename IDX>  // IDX: the odometer's type (see k_chain)
__global__ void __launch_bounds__(256) k_observe_step(const double* __restrict__ a, size_t ap, double* __restrict__ out,
                                                      size_t op, ObserveArgs g, size_t total) {
    typedef typename E::V V;
    const V xv = E::from(g.x), cv = E::from(g.c);
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        IDX r = (IDX)lin;
        size_t off = 0;
        unsigned kv = 0;
        bool in_d = true;  // inside D' on every axis other than v
#pragma unroll 1
        for (int ax = g.out.nd - 1; ax >= 0; --ax) {
            unsigned d = g.out.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            if (ax == g.axis) kv = k;
            else {
                if (k >= g.d_len[ax]) in_d = false;
                off += (size_t)k * g.a_stride[ax];
            }
        }
        const size_t sv = g.a_stride[g.axis];
        const unsigned dl = g.d_len[g.axis];
        // D'_j = a[.., j+1, ..] * ff_j   (derivative, mt:471-479)
        V res;
        if (g.x_is_zero) {  // mul_var only (mt:619-621): zeros, then slab assignment
            res = E::zero();
            if (in_d && kv >= 1 && kv - 1 < dl) res = obs_mul_one<E>(obs_mul_tab<E>(E::ld(a, ap, off + (size_t)kv * sv), E::ld(g.tab, g.tab_plane, kv - 1)));
        } else {  // mul_var(...) + self * from(x): zeros += A; += B  (mt:622, 873-880)
            V A = E::zero();
            if (in_d && kv >= 1 && kv - 1 < dl) A = obs_mul_one<E>(obs_mul_tab<E>(E::ld(a, ap, off + (size_t)kv * sv), E::ld(g.tab, g.tab_plane, kv - 1)));
            res = E::add0(A);  // (0 + A)
            if (in_d && kv < dl) {
                const V di = obs_mul_tab<E>(E::ld(a, ap, off + (size_t)(kv + 1) * sv), E::ld(g.tab, g.tab_plane, kv));
                res = E::addw(res, g.x_is_one ? di : E::mulw(xv, di));
            }
        }
        if (!g.c_is_one) res = E::mulw(cv, res);
        E::st(out, op, lin, res);
    }
}
template <class E>
void K<E>::observe_step(hipStream_t st, const double* a, size_t a_plane, double* out, size_t out_plane,
                        const ObserveArgs& args) {
    size_t total = 1;
    for (int i = 0; i < args.out.nd; ++i) total *= args.out.d[i];
    if (total == 0) return;
    if (total < 0x7fffffffull) GFT_LAUNCH((k_observe_step<E, unsigned>), dim3(grid_for(total)), dim3(256), 0, st, a, a_plane, out, out_plane, args, total);
    else GFT_LAUNCH((k_observe_step<E, size_t>), dim3(grid_for(total)), dim3(256), 0, st, a, a_plane, out, out_plane, args, total);
}

template <class E, bool EPI>
__device__ __forceinline__ void observe_chain_line(const double* __restrict__ a, size_t ap, double* __restrict__ out, size_t op,
                                                   const ObserveChainArgs& g, const typename std::conditional<EPI, ObsEpi, int>::type& epi,
                                                   unsigned line, double* oc_lds) {
    typedef typename E::V V;
    size_t aoff = 0, ooff = 0;
    unsigned kk[MAXD];  // (EPI) the line's coordinates on the collapsed axes; kk[axis] is filled per element
    {
        size_t r = line;
#pragma unroll
        for (int ax = MAXD - 1; ax >= 0; --ax) {
            kk[ax] = 0;
            if (ax < g.nd && ax != g.axis) {
                const unsigned d = g.fs[ax];
                const unsigned k = (unsigned)(r % d);
                r /= d;
                kk[ax] = k;
                aoff += (size_t)k * g.a_stride[ax];
                ooff += (size_t)k * g.o_stride[ax];
            }
        }
    }
    const size_t sva = g.a_stride[g.axis], svo = g.o_stride[g.axis];
    const V xv = E::from(g.x);
    // the derivative factors of this thread's (first) position are the same in every step: read them ONCE — a global load
    // inside the step loop is a full memory latency on a chain of a handful of steps (the whole kernel), as is the
    // full fence of __syncthreads(); indices are < dl[t] <= len0 - 1
    const unsigned k0 = threadIdx.x, ntab = g.len0 ? g.len0 - 1 : 0;
    const V tab_lo = (k0 >= 1 && k0 - 1 < ntab) ? E::ld(g.tab, g.tab_plane, k0 - 1) : E::zero();
    const V tab_hi = k0 < ntab ? E::ld(g.tab, g.tab_plane, k0) : E::zero();
    // (EPI) the other operand's element at this thread's first output position: requested NOW, so that its memory latency runs
    // beside the chain's instead of behind it
    bool y_in0 = false, y_first0 = false, line_first = true;
    size_t y_off_line = 0;
    V y_raw0 = E::zero();
    if constexpr (EPI) {
        y_in0 = true;
        y_first0 = true;
#pragma unroll
        for (int ax = 0; ax < MAXD; ++ax)
            if (ax < g.nd && ax != g.axis) {
                const unsigned ka = kk[ax] - (unsigned)epi.y.pad[ax];  // (wraps below the pad: fails the box test)
                if (ka >= epi.y.box[ax]) y_in0 = false;
                if (ka != 0) y_first0 = false;
                if (kk[ax] != 0) line_first = false;
                y_off_line += (size_t)ka * epi.y.stride[ax];
            }
        const unsigned ka = k0 - (unsigned)epi.y.pad[g.axis];
        const bool in0 = y_in0 && ka < epi.y.box[g.axis];
        if (in0) y_raw0 = E::ld(epi.y.p, epi.y.plane, y_off_line + (size_t)ka * epi.y.stride[g.axis]);
    }
    for (unsigned t = 0; t < g.nsteps; ++t) {
        const unsigned dlt = g.dl[t], lout = g.lo[t];
        const bool first = t == 0, last = t + 1 == g.nsteps;
        const double* src_l = oc_lds + (size_t)((t + 1) & 1u) * E::W * g.lw_pad;  // written by step t - 1
        double* dst_l = oc_lds + (size_t)(t & 1u) * E::W * g.lw_pad;
        const V cv = E::from(g.c[t]);
        const bool c_one = (g.c_one >> t) & 1ull;
        for (unsigned kv = threadIdx.x; kv < lout; kv += blockDim.x) {
            auto src = [&](unsigned j) -> V { return first ? E::ld(a, ap, aoff + (size_t)j * sva) : E::ld(src_l, g.lw_pad, j); };
            // D'_j = src[j + 1] * ff_j  (derivative, mt:471-479), then exactly k_observe_step's sequence
            V res = E::zero();
            const bool mine = kv == k0;
            if (g.x_is_zero) {
                if (kv >= 1 && kv - 1 < dlt) res = obs_mul_one<E>(obs_mul_tab<E>(src(kv), mine ? tab_lo : E::ld(g.tab, g.tab_plane, kv - 1)));
            } else {
                V A = E::zero();
                if (kv >= 1 && kv - 1 < dlt) A = obs_mul_one<E>(obs_mul_tab<E>(src(kv), mine ? tab_lo : E::ld(g.tab, g.tab_plane, kv - 1)));
                res = E::add0(A);  // (0 + A)
                if (kv < dlt) {
                    const V di = obs_mul_tab<E>(src(kv + 1), mine ? tab_hi : E::ld(g.tab, g.tab_plane, kv));
                    res = E::addw(res, g.x_is_one ? di : E::mulw(xv, di));
                }
            }
            if (!c_one) res = E::mulw(cv, res);
            if (!last) E::st(dst_l, g.lw_pad, kv, res);
            else if constexpr (!EPI) E::st(out, op, ooff + (size_t)kv * svo, res);
            else {
                // the consumer's Add, element for element k_chain<E, true>'s operations
                kk[g.axis] = kv;
                const V X = chain_apply<E>(epi.post, epi.npost, nullptr, res, kk, line_first && kv == 0);
                const unsigned ka = kv - (unsigned)epi.y.pad[g.axis];
                const bool iny = y_in0 && ka < epi.y.box[g.axis];
                V Y = E::zero();
                if (iny) {
                    const V raw = mine ? y_raw0 : E::ld(epi.y.p, epi.y.plane, y_off_line + (size_t)ka * epi.y.stride[g.axis]);
                    Y = chain_apply<E>(epi.y.st, epi.y.nstages, epi.y.pad, raw, kk, y_first0 && ka == 0);
                }
                V v = E::zero();
                if (epi.mode == 1) {
                    if (iny) v = E::add(v, Y);
                    v = epi.subtract ? E::sub(v, X) : E::add(v, X);
                } else {
                    v = E::add(v, X);
                    if (iny) v = epi.subtract ? E::sub(v, Y) : E::add(v, Y);
                }
                E::st(out, op, ooff + (size_t)kv * svo, v);
            }
        }
        lds_barrier();  // the line passes through LDS only
    }
}
// The kernel with its arguments — up to 1.9 KB: the steps' lengths and constants, the epilogue's chain — copied to LDS by the
// workgroup first (see k_chain_nest: read in place they are dependent scalar-cache misses, step after step).
template <bool EPI>
struct ObsKArgs {
    const double* a;
    size_t ap;
    double* out;
    size_t op;
    ObserveChainArgs g;
    typename std::conditional<EPI, ObsEpi, int>::type epi;
};
template <class E, bool EPI>
__global__ void __launch_bounds__(1024) k_observe_chain_lds(ObsKArgs<EPI>) {
    extern __shared__ double oc_lds[];
    __shared__ __align__(16) unsigned char s_args[(sizeof(ObsKArgs<EPI>) + 15) / 16 * 16];
    const ObsKArgs<EPI>& A = kernargs_to_lds<ObsKArgs<EPI>>(s_args);
    observe_chain_line<E, EPI>(A.a, A.ap, A.out, A.op, A.g, A.epi, blockIdx.x, oc_lds);
}
template <class E>
void K<E>::observe_chain(hipStream_t st, const double* a, size_t a_plane, double* out, size_t out_plane,
                         const ObserveChainArgs& args, unsigned lines, unsigned longest) {
    observe_chain_epi(st, a, a_plane, out, out_plane, args, lines, longest, nullptr);
}
// optional epilogue (the consumer's Add: ObsEpi)
template <class E>
void K<E>::observe_chain_epi(hipStream_t st, const double* a, size_t a_plane, double* out, size_t out_plane,
                             const ObserveChainArgs& args, unsigned lines, unsigned longest, const ObsEpi* epi) {
    if (lines == 0 || args.nsteps == 0) return;
    const unsigned threads = std::min<unsigned>(1024, (longest + 63) / 64 * 64);
    const size_t lds = (size_t)2 * E::W * args.lw_pad * sizeof(double);
    if (epi) {
        ObsKArgs<true> ka;
        ka.a = a; ka.ap = a_plane; ka.out = out; ka.op = out_plane; ka.g = args; ka.epi = *epi;
        GFT_LAUNCH((k_observe_chain_lds<E, true>), dim3(lines), dim3(threads), lds, st, ka);
    } else {
        ObsKArgs<false> ka;
        ka.a = a; ka.ap = a_plane; ka.out = out; ka.op = out_plane; ka.g = args; ka.epi = 0;
        GFT_LAUNCH((k_observe_chain_lds<E, false>), dim3(lines), dim3(threads), lds, st, ka);
    }
}

template <class E, bool EPI>
__global__ void __launch_bounds__(1024) k_observe_chain_batch(const ObsItem* __restrict__ items) {
    extern __shared__ double oc_lds[];
    __shared__ __align__(16) unsigned char s_args[sizeof(ObsItem)];
    const ObsItem& A = item_to_lds<ObsItem>(items, s_args);
    if (blockIdx.x >= A.lines) return;
    if constexpr (EPI) observe_chain_line<E, true>(A.a, A.ap, A.out, A.op, A.g, A.epi, blockIdx.x, oc_lds);
    else observe_chain_line<E, false>(A.a, A.ap, A.out, A.op, A.g, 0, blockIdx.x, oc_lds);
}
template <class E>
typename K<E>::Geometry K<E>::observe_chain_geometry(unsigned lines, unsigned longest, unsigned lw_pad, unsigned nsteps, bool epi) {
    Geometry g;
    g.gx = lines;
    g.threads = std::min<unsigned>(1024, (longest + 63) / 64 * 64);
    g.lds = (size_t)2 * E::W * lw_pad * sizeof(double);
    g.variant = epi ? 1 : 0;
    g.ok = lines != 0 && nsteps != 0;
    return g;
}
template <class E>
void K<E>::observe_chain_batch(hipStream_t st, const ObsItem* items, unsigned n, const Geometry& g) {
    if (g.variant) GFT_LAUNCH((k_observe_chain_batch<E, true>), dim3(g.gx, n), dim3(g.threads), g.lds, st, items);
    else GFT_LAUNCH((k_observe_chain_batch<E, false>), dim3(g.gx, n), dim3(g.threads), g.lds, st, items);
}

template <class E>
__global__ void k_scalar_op(int op, const double* a, size_t ap, const double* b, size_t bp, double* out,
                            size_t op_plane) {
    (void)op;  // SC_DIV
    E::st(out, op_plane, 0, E::div(E::ld(a, ap, 0), E::ld(b, bp, 0)));
}
template <class E>
void K<E>::scalar_op(hipStream_t st, int op, const double* a, size_t a_plane, const double* b, size_t b_plane,
                     double* out, size_t out_plane) {
    GFT_LAUNCH(k_scalar_op<E>, dim3(1), dim3(1), 0, st, op, a, a_plane, b, b_plane, out, out_plane);
}

template <class E>
__device__ inline typename E::V apply_map(typename E::V x, int op, unsigned u, typename E::V s) {
    switch (op) {
        case MAP_NEG: return E::neg(x);
        case MAP_DIV_U32: return E::div(x, E::from_u32(u));
        case MAP_MUL_U32: return E::mul(x, E::from_u32(u));
        case MAP_MUL_S: return E::mul(x, s);
        case MAP_DIV_S: return E::div(x, s);
        case MAP_LMUL_S: return E::mul(s, x);
        default: return x;
    }
}

template <class E>
__global__ void __launch_bounds__(256) k_map_inplace(double* p, size_t plane, size_t n, int op, unsigned u,
                                                     Scalar2 s, const double* s_ptr, size_t s_plane) {
    typename E::V sv = s_ptr ? E::ld(s_ptr, s_plane, 0) : E::from(s);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        E::st(p, plane, i, apply_map<E>(E::ld(p, plane, i), op, u, sv));
}
template <class E>
void K<E>::map_inplace(hipStream_t st, double* p, size_t plane, size_t n, int op, unsigned u, Scalar2 s) {
    if (n == 0) return;
    GFT_LAUNCH(k_map_inplace<E>, dim3(grid_for(n)), dim3(256), 0, st, p, plane, n, op, u, s,
                       (const double*)nullptr, (size_t)0);
}
template <class E>
void K<E>::map_inplace_dev(hipStream_t st, double* p, size_t plane, size_t n, int op, const double* s_ptr,
                           size_t s_plane) {
    if (n == 0) return;
    GFT_LAUNCH(k_map_inplace<E>, dim3(grid_for(n)), dim3(256), 0, st, p, plane, n, op, 0u, Scalar2{0, 0},
                       s_ptr, s_plane);
}

// ------------------------------------------------------------------------------------------
// block_op: dst[leading block of src's shape] (op)= src
// ------------------------------------------------------------------------------------------
template <class E>
__global__ void __launch_bounds__(256) k_block_op(DView dst, DView src, int op, unsigned u, size_t total) {
    typedef typename E::V V;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        size_t r = lin, doff = 0, dstr = 1;
#pragma unroll 1
        for (int ax = src.sh.nd - 1; ax >= 0; --ax) {
            unsigned d = src.sh.d[ax];
            unsigned k = (unsigned)(r % d);
            r /= d;
            doff += k * dstr;
            dstr *= dst.sh.d[ax];
        }
        V x = E::ld(src.p, src.plane, lin);
        V rv;
        if (op == BLK_ASSIGN) rv = x;
        else {
            V cur = E::ld(dst.p, dst.plane, doff);
            rv = (op == BLK_ADD) ? E::add(cur, x) : E::add(cur, E::mul(E::from_u32(u), x));
        }
        E::st(dst.p, dst.plane, doff, rv);
    }
}
template <class E>
void K<E>::block_op(hipStream_t st, const DView& dst, const DView& src, int op, unsigned u) {
    size_t total = 1;
    for (int i = 0; i < src.sh.nd; ++i) total *= src.sh.d[i];
    if (total == 0) return;
    GFT_LAUNCH(k_block_op<E>, dim3(grid_for(total)), dim3(256), 0, st, dst, src, op, u, total);
}

// ------------------------------------------------------------------------------------------
// sequential 1-D recurrences (one lane; n is a few hundred at most in the reference's workloads)
// ------------------------------------------------------------------------------------------
template <class E>
__global__ void k_exp_1d(const double* xs, size_t xp, unsigned nx, double* res, size_t rp, unsigned n, Scalar2 seed) {
    typedef typename E::V V;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    E::st(res, rp, 0, E::from(seed));
    for (unsigned k = 1; k < n; ++k) {
        V sum = E::zero();
        unsigned hi = nx < k + 1 ? nx : k + 1;
        for (unsigned j = 1; j < hi; ++j)
            sum = E::add(sum, E::mul(E::mul(E::ld(xs, xp, j), E::from_u32(j)), E::ld(res, rp, k - j)));
        E::st(res, rp, k, E::div(sum, E::from_u32(k)));
    }
}
template <class E>
void K<E>::exp_1d(hipStream_t st, const double* xs, size_t x_plane, unsigned nx, double* res, size_t r_plane,
                  unsigned n, Scalar2 seed) {
    GFT_LAUNCH(k_exp_1d<E>, dim3(1), dim3(64), 0, st, xs, x_plane, nx, res, r_plane, n, seed);
}

template <class E>
__global__ void k_log_1d(const double* xs, size_t xp, unsigned nx, double* res, size_t rp, unsigned n, Scalar2 seed) {
    typedef typename E::V V;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    V x0 = E::ld(xs, xp, 0);
    E::st(res, rp, 0, E::from(seed));
    for (unsigned k = 1; k < n; ++k) {
        V sum = E::zero();
        unsigned lo = (k + 1 > nx) ? (k + 1 - nx) : 0;
        if (lo < 1) lo = 1;
        for (unsigned j = lo; j < k; ++j)
            sum = E::add(sum, E::mul(E::mul(E::ld(xs, xp, k - j), E::ld(res, rp, j)), E::from_u32(j)));
        V xk = k < nx ? E::ld(xs, xp, k) : E::zero();
        V num = E::sub(E::mul(xk, E::from_u32(k)), sum);
        E::st(res, rp, k, E::div(E::div(num, x0), E::from_u32(k)));
    }
}
template <class E>
void K<E>::log_1d(hipStream_t st, const double* xs, size_t x_plane, unsigned nx, double* res, size_t r_plane,
                  unsigned n, Scalar2 seed) {
    GFT_LAUNCH(k_log_1d<E>, dim3(1), dim3(64), 0, st, xs, x_plane, nx, res, r_plane, n, seed);
}

template <class E>
__global__ void k_factor_table(int op, unsigned n, unsigned len, const double* m, size_t mp, double* tab,
                               size_t tp) {
    typedef typename E::V V;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (op == TAB_DERIV) {  // mt:472-478
        V ff = E::one();
        for (unsigned i = 1; i <= n; ++i) ff = E::mul(ff, E::from_u32(i));
        for (unsigned k = 0; k < len; ++k) {
            E::st(tab, tp, k, ff);
            ff = E::mul(ff, E::div(E::from_u32(n + k + 1), E::from_u32(k + 1)));
        }
    } else if (op == TAB_COEFF) {  // mt:499-506 (slab 0 is left untouched: factor one)
        V f = E::one();
        E::st(tab, tp, 0, f);
        for (unsigned k = 1; k < len; ++k) {
            f = E::mul(f, E::div(E::from_u32(n + k), E::from_u32(k)));
            E::st(tab, tp, k, f);
        }
    } else if (op == TAB_POW) {  // mt:557-565
        V f = E::one();
        V mv = E::ld(m, mp, 0);
        for (unsigned k = 0; k < len; ++k) {
            E::st(tab, tp, k, f);
            f = E::mul(f, mv);
        }
    } else {  // TAB_INDEX: tab[k] = T::from(k)  (exp/log pre-scaling, mt:1309, 1364)
        for (unsigned k = 0; k < len; ++k) E::st(tab, tp, k, E::from_u32(k));
    }
}
template <class E>
void K<E>::factor_table(hipStream_t st, int op, unsigned n, unsigned len, const double* m, size_t m_plane,
                        double* tab, size_t tab_plane) {
    GFT_LAUNCH(k_factor_table<E>, dim3(1), dim3(64), 0, st, op, n, len, m, m_plane, tab, tab_plane);
}

// ------------------------------------------------------------------------------------------
// axis sums (shift_down, mt:514-536)
// ------------------------------------------------------------------------------------------
// One thread per output (o, i): coalesced across i; sequential ascending k == ndarray's
// slab-by-slab `res = res + view` order.  SUM_UNROLL8 reproduces ndarray's 8-accumulator fold.
template <class E>
__global__ void __launch_bounds__(256) k_sum_axis_seq(const double* __restrict__ in, size_t ip, unsigned outer,
                                                      unsigned len, unsigned inner, size_t outer_stride,
                                                      double* __restrict__ out, size_t op, int mode) {
    typedef typename E::V V;
    size_t total = (size_t)outer * inner;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        unsigned i = (unsigned)(lin % inner);
        size_t o = lin / inner;
        const size_t base = o * outer_stride + i;
        V acc = E::zero();
        if (mode == SUM_UNROLL8) {
            V p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] = E::zero();
            unsigned k = 0;
            for (; k + 8 <= len; k += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) p[u] = E::add(p[u], E::ld(in, ip, base + (size_t)(k + u) * inner));
            }
            acc = E::add(acc, E::add(p[0], p[4]));
            acc = E::add(acc, E::add(p[1], p[5]));
            acc = E::add(acc, E::add(p[2], p[6]));
            acc = E::add(acc, E::add(p[3], p[7]));
            for (; k < len; ++k) acc = E::add(acc, E::ld(in, ip, base + (size_t)k * inner));
        } else {
#pragma unroll 8
            for (unsigned k = 0; k < len; ++k) acc = E::add(acc, E::ld(in, ip, base + (size_t)k * inner));
        }
        E::st(out, op, lin, acc);
    }
}

// Innermost-axis sum of f64 rows: one 64-lane wave per row, coalesced loads, butterfly reduction
// with wavefront shuffles.  (Different summation order than the reference: 1e-10 parity row A13.)
__global__ void __launch_bounds__(256) k_sum_last_axis_wave(const double* __restrict__ in, unsigned rows,
                                                            unsigned len, size_t row_stride,
                                                            double* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave_in_block = threadIdx.x >> 6;
    const unsigned waves_per_block = blockDim.x >> 6;
    for (size_t row = (size_t)blockIdx.x * waves_per_block + wave_in_block; row < rows;
         row += (size_t)gridDim.x * waves_per_block) {
        const double* p = in + row * row_stride;
        double acc = 0.0;
        for (unsigned k = lane; k < len; k += 64) acc += p[k];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) out[row] = acc;
    }
}

// f64, even inner extent: two adjacent outputs per thread (16-byte loads), 8 loads in flight per thread;
// per-output summation order unchanged (sequential ascending k).
__global__ void __launch_bounds__(256) k_sum_axis_seq_f64x2(const double* __restrict__ in, unsigned outer, unsigned len,
                                                            unsigned inner, size_t outer_stride, double* __restrict__ out) {
    const unsigned half = inner >> 1;
    size_t total = (size_t)outer * half;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        unsigned ip = (unsigned)(lin % half);
        size_t o = lin / half;
        const double* p = in + o * outer_stride + 2 * (size_t)ip;
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll 8
        for (unsigned k = 0; k < len; ++k) {
            double2 v = *reinterpret_cast<const double2*>(p + (size_t)k * inner);
            acc.x = acc.x + v.x;
            acc.y = acc.y + v.y;
        }
        *reinterpret_cast<double2*>(out + o * inner + 2 * (size_t)ip) = acc;
    }
}

template <class E>
void K<E>::sum_axis(hipStream_t st, const double* in, size_t in_plane, unsigned outer, unsigned len,
                    unsigned inner, size_t axis_stride_outer, double* out, size_t out_plane, int mode) {
    size_t total = (size_t)outer * inner;
    if (total == 0) return;
    if (mode == SUM_WAVE && E::W == 1 && inner == 1) {
        size_t blocks = (outer + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        GFT_LAUNCH(k_sum_last_axis_wave, dim3((unsigned)blocks), dim3(256), 0, st, in, outer, len,
                           axis_stride_outer, out);
        return;
    }
    if (E::W == 1 && mode != SUM_UNROLL8 && (inner & 1u) == 0 && (axis_stride_outer & 1) == 0 &&
        (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
        GFT_LAUNCH(k_sum_axis_seq_f64x2, dim3(grid_for(total / 2)), dim3(256), 0, st, in, outer, len, inner,
                           axis_stride_outer, out);
        return;
    }
    GFT_LAUNCH(k_sum_axis_seq<E>, dim3(grid_for(total)), dim3(256), 0, st, in, in_plane, outer, len, inner,
                       axis_stride_outer, out, out_plane, mode == SUM_WAVE ? SUM_SEQ : mode);
}

// ------------------------------------------------------------------------------------------
// equality
// ------------------------------------------------------------------------------------------
template <class E>
__global__ void __launch_bounds__(256) k_count_neq(const double* a, size_t ap, const double* b, size_t bp, size_t n,
                                                   unsigned* count) {
    unsigned local = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (!E::eq(E::ld(a, ap, i), E::ld(b, bp, i))) local++;
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}
template <class E>
void K<E>::count_neq(hipStream_t st, const double* a, size_t a_plane, const double* b, size_t b_plane, size_t n,
                     unsigned* count) {
    if (n == 0) return;
    GFT_LAUNCH(k_count_neq<E>, dim3(grid_for(n)), dim3(256), 0, st, a, a_plane, b, b_plane, n, count);
}


// Does the tensor hold a coefficient that is exactly zero ([0,0] for intervals)?  One launch, the answer through the mailbox
// (payload[0] = 1 if so): the premise of the "no exact zero anywhere" proofs of gft_api.hip (Ops::nz_of).  state[0] collects
// the blocks' findings, state[1] counts their arrivals; the last block publishes and leaves both words zero.
template <class E>
__global__ void __launch_bounds__(256) k_any_zero(const double* __restrict__ p, size_t plane, size_t n, unsigned* state, Mailbox mb) {
    int z = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n && !z; i += (size_t)gridDim.x * blockDim.x)
        z = E::is_zero(E::ld(p, plane, i)) ? 1 : 0;
    z = __syncthreads_or(z);
    if (threadIdx.x != 0) return;
    if (z) atomicOr(&state[0], 1u);
    __threadfence();
    if (atomicAdd(&state[1], 1u) != gridDim.x - 1) return;
    __threadfence();
    const unsigned v = atomicExch(&state[0], 0u);
    atomicExch(&state[1], 0u);
    mb.payload[0] = v ? 1.0 : 0.0;
    mailbox_publish(mb);
}
// The same question with the answer "WHERE": payload[0] = number of coefficients that are exactly zero, payload[1 + u] = how
// many of them lie in slab 0 of (collapsed) axis u < 6 — enough for the host to recognise "the zeros are exactly the slabs 0
// of some axes" (gft_api.hip Support: what an observation at X = 0 leaves behind).  state[0 .. 6] count, state[7] arrivals;
// the last block publishes and leaves all eight words zero.
template <class E>
__global__ void __launch_bounds__(256) k_zero_pattern(const double* __restrict__ p, size_t plane, Shape sh, unsigned n, unsigned* state, Mailbox mb) {
    __shared__ unsigned s_cnt[7];
    if (threadIdx.x < 7) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned mine[7] = {0, 0, 0, 0, 0, 0, 0};
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!E::is_zero(E::ld(p, plane, i))) continue;
        mine[0]++;
        unsigned r = i;
        for (int ax = sh.nd - 1; ax >= 0; --ax) {
            const unsigned d = sh.d[ax];
            if (r % d == 0 && ax < 6) mine[1 + ax]++;
            r /= d;
        }
    }
    for (int k = 0; k < 7; ++k)
        if (mine[k]) atomicAdd(&s_cnt[k], mine[k]);
    __syncthreads();
    if (threadIdx.x < 7 && s_cnt[threadIdx.x]) atomicAdd(&state[threadIdx.x], s_cnt[threadIdx.x]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (atomicAdd(&state[7], 1u) != gridDim.x - 1) return;
    __threadfence();
    for (int k = 0; k < 7; ++k) mb.payload[k] = (double)atomicExch(&state[k], 0u);
    atomicExch(&state[7], 0u);
    mailbox_publish(mb);
}
template <class E>
void K<E>::zero_pattern(hipStream_t st, const double* p, size_t plane, const Shape& sh, size_t n, unsigned* state, const Mailbox& mb) {
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 64));
    GFT_LAUNCH(k_zero_pattern<E>, dim3(blocks), dim3(256), 0, st, p, plane, sh, (unsigned)n, state, mb);
}
template <class E>
void K<E>::any_zero(hipStream_t st, const double* p, size_t plane, size_t n, unsigned* state, const Mailbox& mb) {
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 64));
    GFT_LAUNCH(k_any_zero<E>, dim3(blocks), dim3(256), 0, st, p, plane, n, state, mb);
}

// ------------------------------------------------------------------------------------------
// Reference-order convolution: one thread per output element, loops in exactly the reference's
// order (outer axes lexicographic ascending, innermost partial sum from zero), separate
// multiply and add => bit-identical to the CPU algorithm for f64.  Used for small tensors, for
// interval tensors, for rank > 4 and as the on-device cross-check of the tiled kernel.
// ------------------------------------------------------------------------------------------
// OFF: the type of the operand offsets — `unsigned` where both operands have fewer than 2^31 elements (k_conv_shallow: the
// 64-bit multiply-adds of the address arithmetic were most of its ~800 VALU instructions per output, and the kernel is
// instruction-bound: profiles/r04/pmc_k_conv_shallow.txt)
template <class E, int AX, int ND, bool INNER0, typename OFF = size_t>
struct ConvLoop {
    __device__ static inline void run(const ConvArgs& a, const unsigned* k, const double* x, size_t xp,
                                      const double* y, size_t yp, OFF xoff, OFF yoff,
                                      typename E::V& acc) {
        typedef typename E::V V;
        const unsigned kk = k[AX];
        unsigned lo = (kk + 1 > a.ys[AX]) ? (kk + 1 - a.ys[AX]) : 0;
        unsigned hi = (kk + 1 < a.xs[AX]) ? (kk + 1) : a.xs[AX];
        bool desc = false;
        if (AX == 0) {
            if (lo < (unsigned)a.j0_min) lo = (unsigned)a.j0_min;
            if (a.j0_excl && hi > kk) hi = kk;
            desc = a.j0_desc != 0;
        }
        if (hi <= lo) return;
        if constexpr (AX == ND - 1) {
            if (INNER0) {
                V inner = E::zero();
                for (unsigned j = lo; j < hi; ++j)
                    inner = E::add(inner, E::mul(E::ld(x, xp, xoff + (OFF)j * (OFF)a.xstr[AX]),
                                                 E::ld(y, yp, yoff + (OFF)(kk - j) * (OFF)a.ystr[AX])));
                acc = E::add(acc, inner);
            } else {
                const unsigned cnt = hi - lo;
                for (unsigned t = 0; t < cnt; ++t) {
                    unsigned j = desc ? (hi - 1 - t) : (lo + t);
                    acc = E::add(acc, E::mul(E::ld(x, xp, xoff + (OFF)j * (OFF)a.xstr[AX]),
                                             E::ld(y, yp, yoff + (OFF)(kk - j) * (OFF)a.ystr[AX])));
                }
            }
        } else {
            const unsigned cnt = hi - lo;
            for (unsigned t = 0; t < cnt; ++t) {
                unsigned j = desc ? (hi - 1 - t) : (lo + t);
                ConvLoop<E, (AX + 1 < ND ? AX + 1 : AX), ND, INNER0, OFF>::run(
                    a, k, x, xp, y, yp, xoff + (OFF)j * (OFF)a.xstr[AX], yoff + (OFF)(kk - j) * (OFF)a.ystr[AX], acc);
            }
        }
    }
};

template <class E, int ND, bool INNER0>
__global__ void __launch_bounds__(256) k_conv_naive(const double* __restrict__ x, size_t xp,
                                                    const double* __restrict__ y, size_t yp,
                                                    double* __restrict__ z, size_t zp, ConvArgs a, size_t total,
                                                    size_t slab_elems) {
    typedef typename E::V V;
    if (a.guard && *a.guard != a.guard_epoch) return;
    for (size_t lin = blockIdx.x * (size_t)blockDim.x + threadIdx.x; lin < total;
         lin += (size_t)gridDim.x * blockDim.x) {
        size_t zlin = lin + (size_t)a.slab_lo * slab_elems;
        unsigned k[ND > 0 ? ND : 1];
        size_t r = zlin;
#pragma unroll
        for (int ax = ND - 1; ax >= 0; --ax) {
            unsigned d = a.zs[ax];
            k[ax] = (unsigned)(r % d);
            r /= d;
        }
        V acc = a.accumulate ? E::ld(z, zp, zlin) : E::zero();
        if (ND == 0) {
            acc = E::add(acc, E::mul(E::ld(x, xp, 0), E::ld(y, yp, 0)));
        } else {
            ConvLoop<E, 0, (ND > 0 ? ND : 1), INNER0>::run(a, k, x, xp, y, yp, 0, 0, acc);
        }
        E::st(z, zp, zlin, acc);
    }
}

template <class E, bool INNER0>
static void launch_conv_naive(hipStream_t st, const double* x, size_t xp, const double* y, size_t yp, double* z,
                              size_t zp, const ConvArgs& a) {
    size_t slab = 1;
    for (int i = 1; i < a.nd; ++i) slab *= a.zs[i];
    size_t total = (a.nd == 0) ? 1 : (size_t)(a.slab_hi - a.slab_lo) * slab;
    if (total == 0) return;
    dim3 g(grid_for(total)), b(256);
#define GFT_CASE(N)                                                                                         \
    case N:                                                                                                 \
        GFT_LAUNCH((k_conv_naive<E, N, INNER0>), g, b, 0, st, x, xp, y, yp, z, zp, a, total, slab); \
        break;
    switch (a.nd) {
        GFT_CASE(0) GFT_CASE(1) GFT_CASE(2) GFT_CASE(3) GFT_CASE(4) GFT_CASE(5) GFT_CASE(6) GFT_CASE(7)
        GFT_CASE(8) GFT_CASE(9) GFT_CASE(10) GFT_CASE(11) GFT_CASE(12)
        default: break;
    }
#undef GFT_CASE
}

template <class E>
void K<E>::conv_naive(hipStream_t st, const double* x, size_t x_plane, const double* y, size_t y_plane, double* z,
                      size_t z_plane, const ConvArgs& a) {
    if (a.inner_from_zero) launch_conv_naive<E, true>(st, x, x_plane, y, y_plane, z, z_plane, a);
    else launch_conv_naive<E, false>(st, x, x_plane, y, y_plane, z, z_plane, a);
}


// ------------------------------------------------------------------------------------------
// Shallow products with a fused Add (round 4): the GENERAL Horner step  res * subst + slab_i  (mt:569-579) in one launch.
// The substitutions Genfer's programs produce (`b +~ Binomial(a, p)`: a -> a (1 - p + p b)) are 3-6 coefficient
// tensors, so a step's product is a small STENCIL over the accumulator: a handful of terms per output, HBM-bound
// streaming work for which the compute-bound tiled kernel (8x8x8 chunk products over zero padding, plus its prep /
// reduce / guard launches) was 10x too slow.  Same loop nest as k_conv_naive (the reference's order: bit-exact), 32-bit
// index arithmetic, and the consumer's operations applied to the finished sum where it stands:
//   mode 1   out[k] = ((0 + prod[k])? + slab[k]?)   on the leading boxes of out — Add of two tensors (mt:873-880)
//   mode 2   out = prod, element 0 = prod[0] + slab[0]                         — Add of a 1-element slab (mt:862-869)
//   mode 0   out = prod
// and, optionally, the speculative Horner loop's witness of non-linearity on the result (k_witness).
// ------------------------------------------------------------------------------------------
template <class E, int ND, bool INNER0, typename OFF>
__device__ __forceinline__ void conv_shallow_body(const double* __restrict__ x, size_t xp, const double* __restrict__ y, size_t yp, double* __restrict__ out,
                                                  size_t op, const ConvArgs& a, const ConvEpi& e, unsigned total) {
    typedef typename E::V V;
    int found = 0;
    for (unsigned lin = blockIdx.x * 256u + threadIdx.x; lin < total; lin += gridDim.x * 256u) {
        unsigned k[ND];
        unsigned r = lin;
        bool inz = true, ina = e.mode == 1, big = false;
        int nz = 0;
        OFF aoff = 0;  // (the slab's source tensor has fewer than 2^31 elements whenever OFF is 32 bits: the host checks)
#pragma unroll
        for (int ax = ND - 1; ax >= 0; --ax) {
            const unsigned d = e.os[ax];
            const unsigned kk = r % d;
            r /= d;
            k[ax] = kk;
            if (kk >= a.zs[ax]) inz = false;
            if (kk >= e.abox[ax]) ina = false;
            aoff += (OFF)kk * (OFF)e.astr[ax];
            if (kk) nz++;
            if (kk >= 2) big = true;
        }
        V v = E::zero();
        if (inz) {
            V acc = E::zero();
            ConvLoop<E, 0, ND, INNER0, OFF>::run(a, k, x, xp, y, yp, (OFF)0, (OFF)0, acc);
            v = e.mode == 1 ? E::add(v, acc) : acc;
        }
        if (e.mode == 1) {
            if (ina) v = E::add(v, E::ld(e.ap, e.aplane, (size_t)aoff));
        } else if (e.mode == 2 && lin == 0) {
            v = E::add(v, E::ld(e.ap, e.aplane, 0));
        }
        E::st(out, op, lin, v);
        if (e.wit && (big || nz >= 2) && !E::is_zero(v)) found = 1;
    }
    if (e.wit) {  // (kernel argument: uniform)
        // (every workgroup of a dense result raises the word: see wit_raise for why that must not be a write-through store)
        if (__syncthreads_or(found) && threadIdx.x == 0) wit_raise_once(e.wit);
    }
}
template <class E, int ND, bool INNER0, typename OFF>
__global__ void __launch_bounds__(256) k_conv_shallow(const double* __restrict__ x, size_t xp, const double* __restrict__ y,
                                                      size_t yp, double* __restrict__ out, size_t op, ConvArgs a, ConvEpi e,
                                                      unsigned total) {
    conv_shallow_body<E, ND, INNER0, OFF>(x, xp, y, yp, out, op, a, e, total);
}
// ... with the arguments (shapes, strides, the fused Add's box: 0.8 KB) copied to LDS first (see k_chain_nest)
struct ShallowKArgs {
    const double* x;
    size_t xp;
    const double* y;
    size_t yp;
    double* out;
    size_t op;
    ConvArgs a;
    ConvEpi e;
    unsigned total;
};
template <class E, int ND, bool INNER0, typename OFF>
__global__ void __launch_bounds__(256) k_conv_shallow_lds(ShallowKArgs) {
    __shared__ __align__(16) unsigned char s_args[(sizeof(ShallowKArgs) + 15) / 16 * 16];
    const ShallowKArgs& A = kernargs_to_lds<ShallowKArgs>(s_args);
    conv_shallow_body<E, ND, INNER0, OFF>(A.x, A.xp, A.y, A.yp, A.out, A.op, A.a, A.e, A.total);
}

// smallest result (elements) that takes the pair form of k_conv_shallow; negative: never ("shallow_pair_min", GFT_SHALLOW_PAIR_MIN)
static double g_shallow_pair_min = 4096.0;
void shallow_set_pair_min(double v) { g_shallow_pair_min = v; }

// The same with TWO outputs per thread, neighbours along the last axis, for stencils that are FLAT along it (ys[last] == 1:
// the binomial-like substitutions, 2 x 2 x 1): both outputs share every outer index, so the loop nest, its bounds and the y
// coefficients are worked out once for the pair, and the x elements of a term are one 16-byte load (the host checks the
// parities that make it aligned).  Per output the operations and their order are k_conv_shallow's (mt:984-1012) => same bits.
template <class E>
__device__ __forceinline__ void ld_pair(const double* p, size_t plane, size_t i, typename E::V& a, typename E::V& b) {
    if constexpr (E::W == 2) {
        const double2 l = *reinterpret_cast<const double2*>(p + i), h = *reinterpret_cast<const double2*>(p + plane + i);
        a = Iv{l.x, h.x};
        b = Iv{l.y, h.y};
    } else {
        const double2 t = *reinterpret_cast<const double2*>(p + i);
        a = t.x;
        b = t.y;
    }
}
template <class E, int AX, int ND, bool INNER0, typename OFF>
struct ConvLoopPair {
    __device__ static inline void run(const ConvArgs& a, const unsigned* k, const double* x, size_t xp, const double* y, size_t yp, OFF xoff,
                                      OFF yoff, typename E::V& acc0, typename E::V& acc1, bool in1) {
        typedef typename E::V V;
        const unsigned kk = k[AX];
        if constexpr (AX == ND - 1) {
            // flat stencil: the only term of the last axis is j = k (y index 0), if x reaches that far
            if (kk >= a.xs[AX]) return;
            const V yv = E::ld(y, yp, yoff);
            V x0, x1;
            if (kk + 1 < a.xs[AX]) ld_pair<E>(x, xp, (size_t)(xoff + (OFF)kk), x0, x1);
            else x0 = x1 = E::ld(x, xp, (size_t)(xoff + (OFF)kk));
            if (INNER0) {
                acc0 = E::add(acc0, E::add(E::zero(), E::mul(x0, yv)));
                if (in1 && kk + 1 < a.xs[AX]) acc1 = E::add(acc1, E::add(E::zero(), E::mul(x1, yv)));
            } else {
                acc0 = E::add(acc0, E::mul(x0, yv));
                if (in1 && kk + 1 < a.xs[AX]) acc1 = E::add(acc1, E::mul(x1, yv));
            }
        } else {
            const unsigned lo = (kk + 1 > a.ys[AX]) ? (kk + 1 - a.ys[AX]) : 0;
            const unsigned hi = (kk + 1 < a.xs[AX]) ? (kk + 1) : a.xs[AX];
            for (unsigned j = lo; j < hi; ++j)
                ConvLoopPair<E, (AX + 1 < ND ? AX + 1 : AX), ND, INNER0, OFF>::run(a, k, x, xp, y, yp, xoff + (OFF)j * (OFF)a.xstr[AX],
                                                                                  yoff + (OFF)(kk - j) * (OFF)a.ystr[AX], acc0, acc1, in1);
        }
    }
};

template <class E, int ND, bool INNER0, typename OFF>
__global__ void __launch_bounds__(256) k_conv_shallow_pair(const double* __restrict__ x, size_t xp, const double* __restrict__ y, size_t yp,
                                                           double* __restrict__ out, size_t op, ConvArgs a, ConvEpi e, unsigned pairs) {
    typedef typename E::V V;
    int found = 0;
    for (unsigned pl = blockIdx.x * 256u + threadIdx.x; pl < pairs; pl += gridDim.x * 256u) {
        const unsigned lin = 2u * pl;  // (os[last] is even: a pair never straddles two rows)
        unsigned k[ND];
        unsigned r = lin;
        bool inz = true, ina = e.mode == 1, big = false;
        int nz = 0;
        OFF aoff = 0;
#pragma unroll
        for (int ax = ND - 1; ax >= 0; --ax) {
            const unsigned d = e.os[ax];
            const unsigned kk = r % d;
            r /= d;
            k[ax] = kk;
            if (ax < ND - 1) {  // (the last axis is tested per output below)
                if (kk >= a.zs[ax]) inz = false;
                if (kk >= e.abox[ax]) ina = false;
                if (kk) nz++;
                if (kk >= 2) big = true;
            }
            aoff += (OFF)kk * (OFF)e.astr[ax];
        }
        const unsigned kl = k[ND - 1];
        const bool inz0 = inz && kl < a.zs[ND - 1], inz1 = inz && kl + 1 < a.zs[ND - 1];
        V v0 = E::zero(), v1 = E::zero();
        if (inz0) {
            V acc0 = E::zero(), acc1 = E::zero();
            ConvLoopPair<E, 0, ND, INNER0, OFF>::run(a, k, x, xp, y, yp, (OFF)0, (OFF)0, acc0, acc1, inz1);
            v0 = e.mode == 1 ? E::add(v0, acc0) : acc0;
            if (inz1) v1 = e.mode == 1 ? E::add(v1, acc1) : acc1;
        }
        if (e.mode == 1) {
            if (ina && kl < e.abox[ND - 1]) v0 = E::add(v0, E::ld(e.ap, e.aplane, (size_t)aoff));
            if (ina && kl + 1 < e.abox[ND - 1]) v1 = E::add(v1, E::ld(e.ap, e.aplane, (size_t)(aoff + (OFF)e.astr[ND - 1])));
        } else if (e.mode == 2 && lin == 0) {
            v0 = E::add(v0, E::ld(e.ap, e.aplane, 0));
        }
        if constexpr (E::W == 2) {
            *reinterpret_cast<double2*>(out + lin) = double2{v0.lo, v1.lo};
            *reinterpret_cast<double2*>(out + op + lin) = double2{v0.hi, v1.hi};
        } else {
            *reinterpret_cast<double2*>(out + lin) = double2{v0, v1};
        }
        if (e.wit) {
            const int nz0 = nz + (kl ? 1 : 0), nz1 = nz + 1;
            if ((big || kl >= 2 || nz0 >= 2) && !E::is_zero(v0)) found = 1;
            if ((big || kl + 1 >= 2 || nz1 >= 2) && !E::is_zero(v1)) found = 1;
        }
    }
    if (e.wit) {
        if (__syncthreads_or(found) && threadIdx.x == 0) wit_raise_once(e.wit);
    }
}

template <class E>
bool K<E>::conv_shallow(hipStream_t st, const double* x, size_t x_plane, const double* y, size_t y_plane, double* out,
                        size_t out_plane, const ConvArgs& a, const ConvEpi& e) {
    if (a.nd < 1 || a.nd > 6 || a.accumulate || a.j0_min || a.j0_excl || a.j0_desc || a.guard) return false;
    if (a.slab_lo != 0 || a.slab_hi != a.zs[0]) return false;
    unsigned long long total = 1;
    for (int i = 0; i < a.nd; ++i) total *= e.os[i];
    if (total == 0 || total > 0x7fffffffull) return false;
    {   // the pair form: a stencil flat along the last axis, rows of even length, 16-byte alignment of every pair
        const int L = a.nd - 1;
        bool ok = g_shallow_pair_min >= 0.0 && a.nd >= 2 && a.ys[L] == 1 && a.xstr[L] == 1 && e.os[L] % 2 == 0 && (double)total >= g_shallow_pair_min && !((uintptr_t)x & 15) && !((uintptr_t)out & 15) &&
                  x_plane % 2 == 0 && out_plane % 2 == 0;
        for (int i = 0; i < L && ok; ++i) ok = a.xstr[i] % 2 == 0;
        unsigned long long nx = 1, ny = 1, aspan = 1;
        for (int i = 0; i < a.nd; ++i) {
            nx *= a.xs[i];
            ny *= a.ys[i];
            aspan += (unsigned long long)(e.abox[i] ? e.abox[i] - 1 : 0) * e.astr[i];
        }
        if (ok && nx < 0x7fffffffull && ny < 0x7fffffffull && aspan + e.astr[L] < 0x7fffffffull) {
            const unsigned pairs = (unsigned)(total / 2);
            dim3 g(grid_for((size_t)pairs)), b(256);
#define GFT_CASE(N)                                                                                                                                      \
    case N:                                                                                                                                              \
        if (a.inner_from_zero) GFT_LAUNCH((k_conv_shallow_pair<E, N, true, unsigned>), g, b, 0, st, x, x_plane, y, y_plane, out, out_plane, a, e, pairs); \
        else GFT_LAUNCH((k_conv_shallow_pair<E, N, false, unsigned>), g, b, 0, st, x, x_plane, y, y_plane, out, out_plane, a, e, pairs);                 \
        return true;
            switch (a.nd) {
                GFT_CASE(2) GFT_CASE(3) GFT_CASE(4) GFT_CASE(5) GFT_CASE(6)
                default: break;
            }
#undef GFT_CASE
        }
    }
    dim3 g(grid_for((size_t)total)), b(256);
    unsigned long long nx = 1, ny = 1;
    for (int i = 0; i < a.nd; ++i) {
        nx *= a.xs[i];
        ny *= a.ys[i];
    }
    unsigned long long aspan = 1;  // largest slab offset + 1
    for (int i = 0; i < a.nd; ++i) aspan += (unsigned long long)(e.abox[i] ? e.abox[i] - 1 : 0) * e.astr[i];
    const bool u32 = nx < 0x7fffffffull && ny < 0x7fffffffull && aspan < 0x7fffffffull;
    ShallowKArgs ka;
    ka.x = x; ka.xp = x_plane; ka.y = y; ka.yp = y_plane; ka.out = out; ka.op = out_plane; ka.a = a; ka.e = e; ka.total = (unsigned)total;
#define GFT_CASE(N)                                                                                                        \
    case N:                                                                                                                \
        if (u32) {                                                                                                         \
            if (a.inner_from_zero) GFT_LAUNCH((k_conv_shallow_lds<E, N, true, unsigned>), g, b, 0, st, ka);                \
            else GFT_LAUNCH((k_conv_shallow_lds<E, N, false, unsigned>), g, b, 0, st, ka);                                 \
        } else {                                                                                                           \
            if (a.inner_from_zero) GFT_LAUNCH((k_conv_shallow<E, N, true, size_t>), g, b, 0, st, x, x_plane, y, y_plane, out, out_plane, a, e, (unsigned)total); \
            else GFT_LAUNCH((k_conv_shallow<E, N, false, size_t>), g, b, 0, st, x, x_plane, y, y_plane, out, out_plane, a, e, (unsigned)total);                 \
        }                                                                                                                  \
        break;
    switch (a.nd) {
        GFT_CASE(1) GFT_CASE(2) GFT_CASE(3) GFT_CASE(4) GFT_CASE(5) GFT_CASE(6)
        default: return false;
    }
#undef GFT_CASE
    return true;
}

// ------------------------------------------------------------------------------------------
// Line products (round 5): z = x (*) y where one operand is a LINE — extent 1 on every axis but A, which is not the last
// axis (three_populations: [100,1,1] x [100,100,100]; mixture: 290 x 290 by [209,1]).  Per output a 1-d sum of up to ~100
// terms along A: on the one-thread-per-output kernel every term is a global load of the full operand F at stride
// (product of the later axes) — 192 us for 10^6 outputs.  Here a workgroup owns a tile of TK values of k_A x TC columns
// (the other axes, flattened: consecutive threads are consecutive along the last axis): it stages the rows of F the tile
// can see into LDS once (coalesced), then every thread walks its column's TK outputs with F from LDS and the line's
// coefficient from a uniform load.  The reference's order per output (mt:984-1012 with the line's axis an OUTER axis):
// acc = acc + (0 + x_j * y_{k-j}) for ascending j of x — ascending line index if the line is x, descending if it is y —
// separate multiply and add: bit-exact, f64 and interval.
// ------------------------------------------------------------------------------------------
struct LineArgs {
    unsigned zA, fA, LL;      // extents along A: result, full operand, line
    unsigned Cn;              // product of the extents after A (contiguous run of a column index)
    unsigned cols;            // columns = (product of the extents before A) * Cn
    unsigned TC, TK, rows_cap;  // columns and k_A values per workgroup; LDS rows (>= TK + LL - 1)
    int line_is_x;
};
// 256 threads = TC columns x (256 / TC) groups; group g takes the tile's k_A values ka0 + g, ka0 + g + NG, ...  (narrow
// column tiles keep the LDS tile at <= 32 KB, i.e. >= 16 waves per CU: with one 64-column wave per 64 KB tile the kernel was
// latency-bound at 2 waves per CU — 170 us for the 100-term case, no better than one thread per output)
template <class E>
__global__ void __launch_bounds__(256) k_conv_line(const double* __restrict__ F, size_t fp, const double* __restrict__ L, size_t lp,
                                                   double* __restrict__ z, size_t zp, LineArgs g) {
    typedef typename E::V V;
    extern __shared__ double ln_lds[];  // F tile [plane][row][TC], then the line [plane][LL]
    const unsigned TC = g.TC, NG = 256u / TC, tid = threadIdx.x;
    const unsigned col = tid % TC, grp = tid / TC;
    const unsigned q = blockIdx.x * TC + col;       // this thread's column
    const unsigned ka0 = blockIdx.y * g.TK;
    const unsigned ka1 = ka0 + g.TK < g.zA ? ka0 + g.TK : g.zA;
    // rows of F some output of the tile reads: k_A - j for j < LL, clipped to F
    const unsigned r0 = ka0 + 1 > g.LL ? ka0 + 1 - g.LL : 0u, r1 = ka1 < g.fA ? ka1 : g.fA;
    const bool have = q < g.cols;
    const unsigned b = have ? q / g.Cn : 0u, c = have ? q - b * g.Cn : 0u;
    const size_t fplane = (size_t)g.rows_cap * TC;
    double* line = ln_lds + (size_t)E::W * fplane;
    if (have)
        for (unsigned r = r0 + grp; r < r1; r += NG) E::st(ln_lds, fplane, (size_t)(r - r0) * TC + col, E::ld(F, fp, ((size_t)b * g.fA + r) * g.Cn + c));
    for (unsigned i = tid; i < g.LL; i += 256u) E::st(line, g.LL, i, E::ld(L, lp, i));
    __syncthreads();
    if (!have) return;
    for (unsigned ka = ka0 + grp; ka < ka1; ka += NG) {
        // line indices with a partner in F: ka - jl in [0, fA)
        const unsigned lo = ka + 1 > g.fA ? ka + 1 - g.fA : 0u, hi = ka + 1 < g.LL ? ka + 1 : g.LL;
        V acc = E::zero();
        if (g.line_is_x) {
            for (unsigned jl = lo; jl < hi; ++jl)
                acc = E::add(acc, E::add(E::zero(), E::mul(E::ld(line, g.LL, jl), E::ld(ln_lds, fplane, (size_t)(ka - jl - r0) * TC + col))));
        } else {  // ascending index of x = F: descending line index
            for (unsigned jl = hi; jl-- > lo;)
                acc = E::add(acc, E::add(E::zero(), E::mul(E::ld(ln_lds, fplane, (size_t)(ka - jl - r0) * TC + col), E::ld(line, g.LL, jl))));
        }
        E::st(z, zp, ((size_t)b * g.zA + ka) * g.Cn + c, acc);
    }
}
template <class E>
bool K<E>::conv_line(hipStream_t st, const double* x, size_t x_plane, const double* y, size_t y_plane, double* z, size_t z_plane,
                     const ConvArgs& a) {
    static const bool on = true;
    if (!on || a.nd < 2 || a.nd > MAXD || a.accumulate || a.j0_min || a.j0_excl || a.j0_desc || a.guard || !a.inner_from_zero) return false;
    if (a.slab_lo != 0 || a.slab_hi != a.zs[0]) return false;
    // which operand is the line, and along which axis
    auto line_axis = [&](const unsigned* s) -> int {
        int ax = -1;
        for (int i = 0; i < a.nd; ++i)
            if (s[i] != 1) {
                if (ax >= 0) return -1;
                ax = i;
            }
        return ax;
    };
    const int ax_x = line_axis(a.xs), ax_y = line_axis(a.ys);
    int A;
    bool line_is_x;
    if (ax_x >= 0 && ax_y < 0) {
        A = ax_x;
        line_is_x = true;
    } else if (ax_y >= 0 && ax_x < 0) {
        A = ax_y;
        line_is_x = false;
    } else
        return false;
    if (A >= a.nd - 1) return false;  // (a line along the last axis: the reference's inner sum from zero — another order, and strided columns)
    const unsigned* fs = line_is_x ? a.ys : a.xs;
    const size_t* fstr = line_is_x ? a.ystr : a.xstr;
    const unsigned LL = line_is_x ? a.xs[A] : a.ys[A];
    if (LL < 16) return false;  // (a handful of terms: the stencil kernel)
    // contiguous operands and result; the full operand spans the result on every other axis
    size_t zs_ = 1, fs_ = 1;
    for (int i = a.nd - 1; i >= 0; --i) {
        if (a.zstr[i] != zs_ || fstr[i] != fs_) return false;
        if (i != A && fs[i] != a.zs[i]) return false;
        zs_ *= a.zs[i];
        fs_ *= fs[i];
    }
    if ((line_is_x ? a.xstr[A] : a.ystr[A]) != 1) return false;
    if (zs_ >= 0x7fffffffull || fs_ >= 0x7fffffffull) return false;
    LineArgs g;
    g.zA = a.zs[A];
    g.fA = fs[A];
    g.LL = LL;
    g.Cn = 1;
    for (int i = A + 1; i < a.nd; ++i) g.Cn *= a.zs[i];
    g.cols = (unsigned)(zs_ / g.zA);
    g.line_is_x = line_is_x ? 1 : 0;
    // the widest column tile whose LDS tile (one k_A per group) stays within 32 KB
    const size_t budget = (size_t)32 * 1024 - (size_t)LL * 8 * E::W;
    unsigned TC = 0;
    for (unsigned tc = 64; tc >= 8; tc /= 2) {
        const unsigned ng = 256 / tc;
        if ((size_t)(LL + ng - 1) * tc * 8 * E::W <= budget) {
            TC = tc;
            break;
        }
    }
    if (!TC || (size_t)LL * 8 * E::W >= (size_t)16 * 1024) return false;  // (a line too long for the tile)
    const unsigned NG = 256 / TC, ctiles = (g.cols + TC - 1) / TC;
    // k_A values per workgroup: multiples of the groups while the tile fits and ~1000 workgroups remain
    unsigned TK = NG;
    while ((size_t)(LL + 2 * TK - 1) * TC * 8 * E::W <= budget && (unsigned long long)ctiles * ((g.zA + 2 * TK - 1) / (2 * TK)) >= 1024ull) TK *= 2;
    g.TC = TC;
    g.TK = TK;
    g.rows_cap = TK + LL - 1;
    const size_t lds = ((size_t)g.rows_cap * TC + LL) * 8 * E::W;
    const double* F = line_is_x ? y : x;
    const double* Lp = line_is_x ? x : y;
    GFT_LAUNCH(k_conv_line<E>, dim3(ctiles, (g.zA + TK - 1) / TK), dim3(256), lds, st, F, line_is_x ? y_plane : x_plane, Lp, line_is_x ? x_plane : y_plane, z,
               z_plane, g);
    return true;
}

struct UploadChunk {
    double v[480];
};
__global__ void __launch_bounds__(256) k_upload_small(double* __restrict__ dst, UploadChunk c, unsigned n) {
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) dst[i] = c.v[i];
}
void upload_small(hipStream_t st, double* dst, const double* host_src, size_t n) {
    for (size_t off = 0; off < n; off += 480) {
        UploadChunk c;
        const unsigned cnt = (unsigned)std::min<size_t>(480, n - off);
        std::memcpy(c.v, host_src + off, sizeof(double) * cnt);
        GFT_LAUNCH(k_upload_small, dim3(1), dim3(256), 0, st, dst + off, c, cnt);
    }
}

__global__ void k_peek(const double* __restrict__ src, size_t stride, unsigned n, Mailbox mb) {
    if (threadIdx.x < n) mb.payload[threadIdx.x] = src[threadIdx.x * stride];
    __syncthreads();
    if (threadIdx.x == 0) mailbox_publish(mb);
}
void peek_to_mailbox(hipStream_t st, const double* src, size_t stride, unsigned n, const Mailbox& mb) {
    GFT_LAUNCH(k_peek, dim3(1), dim3(64), 0, st, src, stride, n, mb);
}

template struct K<EF64>;
template struct K<EIv>;

}  // namespace gft
