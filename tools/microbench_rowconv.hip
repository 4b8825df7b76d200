// Cost of one step of the in-register row product of gft_div2d.hip (k_div_2d_rows64): which of its ingredients —
// v_readlane broadcast, DPP wave shift, the f64 mul/add chain — sets the cycles per step, for 1..16 waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench_rowconv.hip -o /tmp/mbrc && /tmp/mbrc
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ inline double bcast_f64(double x, unsigned j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), (int)j), hi = __builtin_amdgcn_readlane(__double2hiint(x), (int)j);
    return __hiloint2double(hi, lo);
}
__device__ inline double wave_shr1_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ inline double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// G independent chains sharing one readlane broadcast (the shape of k_div_2d_rows64's update loop); CTRL = DPP control:
// 0x138 wave_shr:1 (crosses the 16-lane rows), 0x111 row_shr:1 (stays inside them)
template <int G, int CTRL>
__global__ void kg(double* out, long long* cycles, int reps, int n2) {
    const unsigned c = threadIdx.x & 63;
    double xr = 1.0 + c * 1e-3, acc = 0.0;
    long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        double ys[G], inner[G];
#pragma unroll
        for (int q = 0; q < G; ++q) ys[q] = 0.5 + c * 1e-3 + r + q, inner[q] = 0.0;
#pragma unroll 2
        for (int j2 = 0; j2 < n2; ++j2) {
            const double xs = bcast_f64(xr, j2);
#pragma unroll
            for (int q = 0; q < G; ++q) {
                inner[q] = inner[q] + xs * ys[q];
                ys[q] = dpp_f64<CTRL>(ys[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < G; ++q) acc += inner[q];
        xr += 1e-9;
    }
    long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
template <int G, int CTRL>
void rung(const char* name, double* out, long long* cyc) {
    for (int waves : {1, 4, 8, 16}) {
        const int reps = 200, n2 = 64;
        hipLaunchKernelGGL((kg<G, CTRL>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, reps, n2);
        hipDeviceSynchronize();
        long long c;
        hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
        printf("%-34s waves/CU %2d: %8.1f ns per 64-step group of %d row products, %6.2f ns per step (cycles @2.4 GHz: %6.1f, per chain %5.1f)\n", name,
               waves, c * 10.0 / reps, G, c * 10.0 / reps / n2, c * 10.0 / reps / n2 * 2.4, c * 10.0 / reps / n2 * 2.4 / G);
    }
}

// mode bit 0: readlane broadcast (else a per-lane constant); bit 1: DPP shift (else none); bit 2: LDS operands instead
template <int MODE>
__global__ void k(double* out, long long* cycles, int reps, int n2) {
    __shared__ double lds[2][64 * 17];
    const unsigned c = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds[0][wave * 64 + c] = 1.0 + c * 1e-3;
    lds[1][wave * 64 + c] = 0.5 + c * 1e-3;
    __syncthreads();
    double xr = 1.0 + c * 1e-3, acc = 0.0;
    long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        double ys = 0.5 + c * 1e-3 + r, inner = 0.0;
        if (MODE & 4) {
#pragma unroll 4
            for (int j2 = 0; j2 < n2; ++j2) inner = inner + lds[0][wave * 64 + j2] * lds[1][wave * 64 + ((c - j2) & 63)];
        } else {
#pragma unroll 4
            for (int j2 = 0; j2 < n2; ++j2) {
                const double xs = (MODE & 1) ? bcast_f64(xr, j2) : xr;
                inner = inner + xs * ys;
                if (MODE & 2) ys = wave_shr1_f64(ys);
            }
        }
        acc += inner;
        xr += 1e-9;
    }
    long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, double* out, long long* cyc) {
    for (int waves : {1, 4, 8, 16}) {
        const int reps = 200, n2 = 64;
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, reps, n2);
        hipDeviceSynchronize();
        long long c;
        hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
        // wall_clock64 ticks at 100 MHz; the shader clock is ~2.4 GHz
        printf("%-34s waves/CU %2d: %8.1f ns per 64-step row product, %6.2f ns per step (x2.4 = cycles @2.4 GHz: %6.1f)\n", name, waves,
               c * 10.0 / reps, c * 10.0 / reps / n2, c * 10.0 / reps / n2 * 2.4);
    }
}

int main() {
    double* out;
    long long* cyc;
    hipMalloc(&out, 1024 * sizeof(double));
    hipMalloc(&cyc, 64);
    run<0>("mul+add only", out, cyc);
    run<1>("readlane + mul+add", out, cyc);
    run<2>("dpp shift + mul+add", out, cyc);
    run<3>("readlane + dpp + mul+add (kernel)", out, cyc);
    run<4>("two LDS reads + mul+add", out, cyc);
    rung<5, 0x138>("5 chains, wave_shr:1", out, cyc);
    rung<5, 0x111>("5 chains, row_shr:1", out, cyc);
    rung<2, 0x138>("2 chains, wave_shr:1", out, cyc);
    rung<8, 0x111>("8 chains, row_shr:1", out, cyc);
    return 0;
}
