// The launch thread (round 3).  A Genfer program is 10^4-10^5 dependent kernels of 2-6 us; after the deferred chains
// (gft_kernels.hpp ChainSrc) the wall clock of such a run was the HOST: ~3 us inside hipLaunchKernel per launch on the
// one thread that also interprets the program.  So the API thread does not launch: GFT_LAUNCH copies the kernel's
// arguments into a slot of a single-producer / single-consumer ring and a worker thread issues the launches to their
// streams in that order.  The interpreter runs ahead of the launches the way the stream runs ahead of the GPU.
//
// Ordering rules (all enforced here, none left to callers):
//   * launches are issued in program order, so everything that was legal on one in-order stream still is — in
//     particular the pool's "a freed block may be reused by a later launch" rule;
//   * every other operation on a stream (copies, memsets, event records / waits, synchronisation, hipFree, RCCL) first
//     waits until the worker has issued everything queued so far (launch_drain): the hip* names used by this library are
//     wrapped below; the frequent device-to-device copies / memsets / event operations of the recurrences are queued as
//     tasks instead (enqueue), so they do not stall the API thread;
//   * the raw entry points that exist to interoperate with the CALLER's stream (gft_conv_raw*, gft_dist_*) drain before
//     they return, so a caller that records an event or launches its own work on that stream afterwards sees the
//     library's launches in the stream.  gft_synchronize() and every value inspection drain as well.
// GFT_ASYNC_LAUNCH=0 (or gft_set_option("async_launch", 0)) launches on the calling thread as before (A/B, debugging).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <new>
#include <tuple>
#include <type_traits>
#include <utility>

namespace gft {

extern unsigned long long g_launches;  // launches requested so far (gft_op_stats_ex)
// every operation handed to a stream so far — launches, queued tasks, and the wrapped hip* calls the API thread makes itself
// (the side-stream scopes of gft_api.hip tell "nothing was issued on the main stream since" by it)
extern unsigned long long g_stream_ops;

constexpr size_t LQ_SLOT_BYTES = 4096;  // (k_upload_small's 480 doubles by value are the largest block: 3.9 KB)
extern unsigned long long g_launches_in_place;  // closures too large for a slot: launched on the calling thread after a full drain
  // largest argument block that travels through the ring (bigger: launched in place)
struct LaunchSlot {
    void (*run)(void*);
    alignas(16) unsigned char payload[LQ_SLOT_BYTES];
};
bool lq_enabled();
int lq_debug();  // GFT_ASYNC_DEBUG bits: 1 = wait for the worker after every queued item, 2 = non-launch tasks run on the calling thread
void lq_configure(int device, bool enabled);  // gft_init / options
void lq_shutdown();                           // drains and stops the worker
void lq_report();                             // GFT_TRACE_LQ=1: prints the two threads' time split since the last report (stderr)
LaunchSlot* lq_begin();                       // next free slot (waits while the ring is full; starts the worker on first use)
void lq_commit();                             // publishes the slot written since lq_begin
void launch_drain();                          // returns when the worker has issued everything queued before this call;
                                              // THROWS std::runtime_error if a queued launch / stream operation failed
// Failures of queued work.  HIP's last-error state is per thread, so the API thread cannot see what happened to a launch
// the worker issued: the worker (and the in-place path) hands every status to lq_note, which latches the FIRST failure
// with the kernel's address; launch_drain() — i.e. every value inspection, gft_synchronize and every wrapped hip* call —
// raises it (once) as an exception that reaches the caller through gft_last_error().
void launch_drain_nothrow();                  // the same wait without raising (destructors, shutdown)
void lq_note(hipError_t e, const void* kernel, const char* what);
// test knob (gft_set_option("debug_fail_next_launch", 1)): the next kernel launch requests 1 MB of LDS and fails
extern std::atomic<int> g_fail_next_launch;

template <class F>
inline void enqueue(F&& f) {
    typedef typename std::decay<F>::type Fn;
    ++g_stream_ops;
    if (!lq_enabled() || sizeof(Fn) > LQ_SLOT_BYTES) {
        if (lq_enabled()) ++g_launches_in_place;
        launch_drain();
        f();
        return;
    }
    LaunchSlot* s = lq_begin();
    new (s->payload) Fn(std::forward<F>(f));
    s->run = [](void* p) {
        Fn* fn = static_cast<Fn*>(p);
        (*fn)();
        fn->~Fn();
    };
    lq_commit();
    if (lq_debug() & 1) launch_drain();
}
// a stream operation that is not a kernel launch (copy, memset, event): queued like a launch
template <class F>
inline void enqueue_task(F&& f) {
    if (lq_debug() & 2) {
        ++g_stream_ops;
        launch_drain();
        f();
        return;
    }
    enqueue(std::forward<F>(f));
}

// One queued kernel launch.  The argument tuple is built ONCE, in the ring slot (the fused observation / nested-Add launches
// carry 2.5-3.5 KB of arguments: built on the stack, captured and moved into the slot they were copied three times, and the
// calling thread is what the launch-bound programs wait for — profiles/r05/host_profile.txt).
template <class... KArgs>
struct LaunchFn {
    void (*kernel)(KArgs...);
    dim3 grid, block;
    unsigned lds;
    hipStream_t st;
    std::tuple<typename std::decay<KArgs>::type...> t;
    template <class... Args>
    LaunchFn(void (*k)(KArgs...), dim3 g, dim3 b, unsigned l, hipStream_t s, Args&&... args)
        : kernel(k), grid(g), block(b), lds(l), st(s), t(std::forward<Args>(args)...) {}
    void operator()() {
        std::apply([&](auto&... a) { hipLaunchKernelGGL(kernel, grid, block, lds, st, a...); }, t);
        const hipError_t e = hipGetLastError();  // this thread's state: the launch just made
        if (e != hipSuccess) lq_note(e, reinterpret_cast<const void*>(kernel), nullptr);
    }
};
template <class... KArgs, class... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args&&... args) {
    typedef LaunchFn<KArgs...> Fn;
    ++g_launches;
    if (g_fail_next_launch.load(std::memory_order_relaxed) && g_fail_next_launch.exchange(0)) lds = (size_t)1 << 20;
    if (!lq_enabled() || sizeof(Fn) > LQ_SLOT_BYTES) {  // (the arguments are evaluated by now: nothing below launches)
        enqueue(Fn(kernel, grid, block, (unsigned)lds, st, std::forward<Args>(args)...));
        return;
    }
    ++g_stream_ops;
    LaunchSlot* s = lq_begin();
    new (s->payload) Fn(kernel, grid, block, (unsigned)lds, st, std::forward<Args>(args)...);
    s->run = [](void* p) {
        Fn* fn = static_cast<Fn*>(p);
        (*fn)();
        fn->~Fn();
    };
    lq_commit();
    if (lq_debug() & 1) launch_drain();
}

}  // namespace gft

#define GFT_LAUNCH(...) ::gft::launch(__VA_ARGS__)

// Stream operations issued directly by the API thread: first let the worker catch up.  The ARGUMENTS are evaluated
// before the drain (they may launch: dp() materialises deferred chains and lazy handles), hence a call through a generic
// lambda and not a comma expression.  (A macro may name itself in its own replacement list without recursing, so the
// wrapped call is the real HIP function.)
#ifndef GFT_LAUNCH_NO_WRAP
#define GFT_DRAINED(fn, ...)                              \
    ([&](auto&&... a_) {                                  \
        ++::gft::g_stream_ops;                            \
        ::gft::launch_drain();                            \
        return (fn)(static_cast<decltype(a_)&&>(a_)...);  \
    }(__VA_ARGS__))
#define hipMemcpyAsync(...) GFT_DRAINED(hipMemcpyAsync, __VA_ARGS__)
#define hipMemsetAsync(...) GFT_DRAINED(hipMemsetAsync, __VA_ARGS__)
#define hipMemsetD32Async(...) GFT_DRAINED(hipMemsetD32Async, __VA_ARGS__)
#define hipMemcpy(...) GFT_DRAINED(hipMemcpy, __VA_ARGS__)
#define hipStreamSynchronize(...) GFT_DRAINED(hipStreamSynchronize, __VA_ARGS__)
#define hipStreamQuery(...) GFT_DRAINED(hipStreamQuery, __VA_ARGS__)
#define hipEventRecord(...) GFT_DRAINED(hipEventRecord, __VA_ARGS__)
#define hipStreamWaitEvent(...) GFT_DRAINED(hipStreamWaitEvent, __VA_ARGS__)
#define hipFree(...) GFT_DRAINED(hipFree, __VA_ARGS__)
#define hipStreamDestroy(...) GFT_DRAINED(hipStreamDestroy, __VA_ARGS__)
#endif
