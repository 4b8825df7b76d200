// Microbenchmark: FP64 issue rates on gfx950 — v_fma_f64 (VGPR and SGPR operand forms) and
// v_mfma_f64_16x16x4_f64 — to calibrate the roof the tiled convolution is priced against.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_fp64.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k_fma_vgpr(double* out, int iters, double a, double b) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x + i;
    double y = b + threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], y, a);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// acc[i] = fma(s_x (SGPR), y_i (VGPR), acc[i]) — the shape of the convolution's inner loop
template <int NACC>
__global__ void __launch_bounds__(256) k_fma_sgpr(double* out, int iters, const double* __restrict__ xs) {
    double acc[NACC], y[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc[i] = 0; y[i] = threadIdx.x * 1e-3 + i; }
    for (int it = 0; it < iters; ++it) {
        double x = xs[it & 255];  // uniform address -> scalar load
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(x, y[i], acc[i]);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    d4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    double *out, *xs;
    CK(hipMalloc(&out, 256 * 2048 * 8 * sizeof(double)));
    CK(hipMalloc(&xs, 256 * sizeof(double)));
    std::vector<double> h(256, 1.000001);
    CK(hipMemcpy(xs, h.data(), 256 * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    auto run = [&](const char* name, auto launch, double flop_per_thread_iter, int blocks) -> int {
        launch(blocks);  // warm
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            launch(blocks);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        double flops = flop_per_thread_iter * iters * 256.0 * blocks;
        printf("%-44s blocks/CU %2d  %8.3f ms  %7.2f TFLOP/s\n", name, blocks / 256, best, flops / best / 1e9);
        return 0;
    };
    for (int bpc : {1, 2, 4, 8}) {
        int blocks = 256 * bpc;
        run("v_fma_f64 vgpr 16 acc", [&](int b) { hipLaunchKernelGGL(k_fma_vgpr<16>, dim3(b), dim3(256), 0, 0, out, iters, 1.0, 1.0); }, 2.0 * 16, blocks);
        run("v_fma_f64 sgpr-x 16 acc", [&](int b) { hipLaunchKernelGGL(k_fma_sgpr<16>, dim3(b), dim3(256), 0, 0, out, iters, xs); }, 2.0 * 16, blocks);
        run("v_fma_f64 sgpr-x 8 acc", [&](int b) { hipLaunchKernelGGL(k_fma_sgpr<8>, dim3(b), dim3(256), 0, 0, out, iters, xs); }, 2.0 * 8, blocks);
        run("mfma_f64_16x16x4 4 acc", [&](int b) { hipLaunchKernelGGL(k_mfma<4>, dim3(b), dim3(256), 0, 0, out, iters); }, 2.0 * 16 * 16 * 4 * 4 / 64.0, blocks);
        run("mfma_f64_16x16x4 1 acc", [&](int b) { hipLaunchKernelGGL(k_mfma<1>, dim3(b), dim3(256), 0, 0, out, iters); }, 2.0 * 16 * 16 * 4 * 1 / 64.0, blocks);
    }
    return 0;
}
