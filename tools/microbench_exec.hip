// Does a wave64 whose upper (or 3 of 4) lane groups are masked off issue 64-bit VALU operations faster?  One wave per SIMD
// (4 blocks of one wave per CU), 8 independent chains per lane, lanes >= ACTIVE leave before the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_exec.hip -o /tmp/mbexec && /tmp/mbexec
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(unsigned long long* out, int iters, unsigned long long seed, int active) {
    unsigned long long a[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 8 + i; d[i] = (double)(seed + i) * 1e-3; }
    if ((int)(threadIdx.x & 63) < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_lshl_add_u64 %0, %0, 0, 1" : "+v"(a[i]));
                if (MODE == 2) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i]));
                if (MODE == 3) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));
                if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
            }
        }
    }
    unsigned long long s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (unsigned long long)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, unsigned long long* out) {
    for (int wpc : {4, 8}) {
        for (int active : {64, 32, 16}) {
            const int iters = 20000, blocks = 256 * 4, threads = wpc * 64 / 4;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1ull, active);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1ull, active);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double ops_per_simd = (double)iters * 8 * (wpc / 4.0);
            printf("%-16s waves/CU %2d active lanes %2d: %7.2f cycles per wave-op per SIMD @2.4 GHz\n", name, wpc, active, ms * 1e-3 * 2.4e9 / ops_per_simd);
        }
    }
}
int main() {
    unsigned long long* out;
    hipMalloc(&out, sizeof(unsigned long long) * 256 * 4 * 1024);
    run<0>("v_lshl_add_u64", out);
    run<2>("v_add_f64", out);
    run<3>("v_mul_f64", out);
    run<4>("v_fma_f64", out);
    return 0;
}
