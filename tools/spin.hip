// One wave that stays resident for N seconds (s_sleep between looks at the constant-rate clock): does a resident wave keep
// the shader clock up for a launch-bound program running beside it?   hipcc --offload-arch=gfx950 -O3 tools/spin.hip -o /tmp/spin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void spin(unsigned long long ticks, unsigned long long* out, int busy) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long n = 0;
    double x = 1.0;
    while (wall_clock64() - t0 < ticks) {
        if (busy) { for (int i = 0; i < 64; ++i) x = x * 1.0000001 + 1e-9; } else __builtin_amdgcn_s_sleep(64);
        ++n;
    }
    out[0] = n + (unsigned long long)x;
}
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 10.0;
    const int busy = argc > 2 ? atoi(argv[2]) : 0, blocks = argc > 3 ? atoi(argv[3]) : 1;
    unsigned long long* out;
    hipMalloc(&out, 8);
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, 0, (unsigned long long)(secs * 1e8), out, busy);
    hipDeviceSynchronize();
    printf("spin done\n");
    return 0;
}
