// Issue rate of the 64-bit integer step that interval widening uses (bits +- 1) on gfx950: v_lshl_add_u64 (what hipcc
// emits for a 64-bit add) against a v_add_co_u32 / v_addc_co_u32 pair and against v_add_f64, 8 independent chains per
// lane, 1..8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/microbench_int64.hip -o /tmp/mb64 && /tmp/mb64
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(unsigned long long* out, int iters, unsigned long long seed) {
    unsigned long long a[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 8 + i; d[i] = (double)(seed + i) * 1e-3; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_lshl_add_u64 %0, %0, 0, 1" : "+v"(a[i]));
            if (MODE == 1) {
                unsigned lo = (unsigned)a[i], hi = (unsigned)(a[i] >> 32);
                asm volatile("v_add_co_u32 %0, vcc, 1, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : : "vcc");
                a[i] = ((unsigned long long)hi << 32) | lo;
            }
            if (MODE == 2) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[i]));
            if (MODE == 3) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[i]));
        }
    }
    unsigned long long s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (unsigned long long)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, unsigned long long* out) {
    for (int wpc : {4, 8, 16, 32}) {
        const int iters = 20000, blocks = 256 * 4, threads = wpc * 64 / 4;  // 4 blocks per CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1ull);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1ull);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops_per_simd = (double)iters * 8 * (wpc / 4.0);  // wave-instructions (pairs count once) per SIMD
        printf("%-34s waves/CU %2d: %7.2f cycles per wave-op per SIMD @2.4 GHz\n", name, wpc, ms * 1e-3 * 2.4e9 / ops_per_simd);
    }
}
int main() {
    unsigned long long* out;
    hipMalloc(&out, sizeof(unsigned long long) * 256 * 4 * 1024);
    run<0>("v_lshl_add_u64", out);
    run<1>("v_add_co_u32 + v_addc_co_u32", out);
    run<2>("v_add_f64", out);
    run<3>("v_mul_f64", out);
    return 0;
}
