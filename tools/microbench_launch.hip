// Floor of the launch-bound regime: host time per hipLaunchKernelGGL of an empty kernel on one stream (no sync inside
// the loop).  hipcc --offload-arch=gfx950 -O2 tools/microbench_launch.hip -o tools/microbench_launch
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
__global__ void k_empty(double* p, double v) {
    if (p) p[0] = v;
}
int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double* d;
    hipMalloc(&d, 64);
    const int N = 200000;
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d, 1.0);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, d, (double)i);
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(st);
    auto t2 = std::chrono::steady_clock::now();
    std::printf("host time per launch: %.2f us; including drain: %.2f us per kernel\n",
                std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
                std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    return 0;
}
