// Cost of ordering work ACROSS streams on one device (round 5: side streams for the independent arms of an `if`).
//   a) N dependent 3 us kernels on one stream
//   b) fork / join per iteration with HIP events: main kernel; record(main) -> side waits; side kernel; record(side);
//      main kernel; main waits side's event; main kernel
//   c) the same with stream memory operations (hipStreamWriteValue64 / hipStreamWaitValue64) instead of events
//   d) K independent chains on K streams, no cross-stream ordering at all (what overlap can buy)
// hipcc --offload-arch=gfx950 -O2 tools/microbench_streams.hip -o /tmp/microbench_streams
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_work(double* p, int n) {
    double x = p[threadIdx.x];
    for (int i = 0; i < n; ++i) x = x * 1.0000001 + 1e-9;
    p[threadIdx.x] = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int N = 5000, W = 300;  // W: ~3 us of dependent FMAs
    hipStream_t ms, ss[8];
    hipStreamCreateWithFlags(&ms, hipStreamNonBlocking);
    for (auto& s : ss) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double* d;
    hipMalloc(&d, 64 * 8 * 16);
    hipMemset(d, 0, 64 * 8 * 16);
    std::vector<hipEvent_t> ev(2 * N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    auto sync_all = [&] {
        hipStreamSynchronize(ms);
        for (auto& s : ss) hipStreamSynchronize(s);
    };
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < 4 * N; ++i) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
        double t1 = now();
        sync_all();
        double t2 = now();
        printf("a) one stream: %.2f us per kernel (host issue %.2f us)\n", (t2 - t0) / (4 * N) * 1e6, (t1 - t0) / (4 * N) * 1e6);
        for (int ns = 1; ns <= 4; ns *= 2) {
            t0 = now();
            for (int i = 0; i < N; ++i) {
                hipStream_t s = ss[i % ns];
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
                hipEventRecord(ev[2 * i], ms);
                hipStreamWaitEvent(s, ev[2 * i], 0);
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, d + 64 * (1 + i % ns), W);
                hipEventRecord(ev[2 * i + 1], s);
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
                hipStreamWaitEvent(ms, ev[2 * i + 1], 0);
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
            }
            t1 = now();
            sync_all();
            t2 = now();
            printf("b) fork/join with events, %d side stream(s): %.2f us per iteration of 4 kernels (host issue %.2f us)\n", ns, (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
        }
        // deferred join: the side result is consumed LAG iterations later (the memoised-arm pattern)
        for (int lag : {1, 8}) {
            t0 = now();
            for (int i = 0; i < N; ++i) {
                hipStream_t s = ss[i % 4];
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
                hipEventRecord(ev[2 * i], ms);
                hipStreamWaitEvent(s, ev[2 * i], 0);
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, d + 64 * (1 + i % 4), W);
                hipEventRecord(ev[2 * i + 1], s);
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
                if (i >= lag) hipStreamWaitEvent(ms, ev[2 * (i - lag) + 1], 0);
                hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
            }
            t1 = now();
            sync_all();
            t2 = now();
            printf("b') join %d iterations late, 4 side streams: %.2f us per iteration of 4 kernels (host issue %.2f us)\n", lag, (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
        }
        {
            uint64_t* sig = nullptr;
            hipError_t e = hipExtMallocWithFlags((void**)&sig, 4096, hipMallocSignalMemory);
            if (e != hipSuccess) {
                printf("c) hipMallocSignalMemory: %s\n", hipGetErrorString(e));
                (void)hipGetLastError();
            } else {
                hipMemset(sig, 0, 4096);
                hipDeviceSynchronize();
                bool ok = true;
                t0 = now();
                for (int i = 0; i < N && ok; ++i) {
                    hipStream_t s = ss[0];
                    const uint64_t v = (uint64_t)rep * 4 * N + 2 * i + 1;
                    hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
                    ok = ok && hipStreamWriteValue64(ms, sig, v, 0) == hipSuccess;
                    ok = ok && hipStreamWaitValue64(s, sig, v, hipStreamWaitValueGte) == hipSuccess;
                    hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, d + 64, W);
                    ok = ok && hipStreamWriteValue64(s, sig + 8, v + 1, 0) == hipSuccess;
                    hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
                    ok = ok && hipStreamWaitValue64(ms, sig + 8, v + 1, hipStreamWaitValueGte) == hipSuccess;
                    hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ms, d, W);
                }
                t1 = now();
                sync_all();
                t2 = now();
                printf("c) fork/join with stream memory ops (%s): %.2f us per iteration (host issue %.2f us)\n", ok ? "ok" : "FAILED", (t2 - t0) / N * 1e6, (t1 - t0) / N * 1e6);
                (void)hipGetLastError();
                hipFree(sig);
            }
        }
        for (int K : {2, 4, 8}) {
            t0 = now();
            for (int i = 0; i < 4 * N; ++i) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, ss[i % K], d + 64 * (i % K), W);
            t1 = now();
            sync_all();
            t2 = now();
            printf("d) %d independent streams: %.2f us per kernel (host issue %.2f us)\n", K, (t2 - t0) / (4 * N) * 1e6, (t1 - t0) / (4 * N) * 1e6);
        }
        // event query cost
        t0 = now();
        int done = 0;
        for (int i = 0; i < 2 * N; ++i) done += hipEventQuery(ev[i]) == hipSuccess;
        t1 = now();
        printf("e) hipEventQuery: %.2f us each (%d complete)\n", (t1 - t0) / (2 * N) * 1e6, done);
    }
    return 0;
}
