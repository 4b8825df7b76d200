// Host-side cost of one asynchronous C-ABI operation (launch-bound regime): time per call of a few representative
// entry points on tiny tensors, no synchronisation inside the timed loop.
//   g++ -O2 -std=c++17 tools/microbench_api.cpp -Iinclude -Lgenfer_amd/csrc -lgftaylor -Wl,-rpath,$PWD/genfer_amd/csrc -o /tmp/mb_api
#include <chrono>
#include <cstdio>
#include <vector>

#include "gftaylor.h"

template <class F>
static double per_call_us(int n, F&& f) {
    f();
    gft_synchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    auto t1 = std::chrono::steady_clock::now();
    gft_synchronize();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}

int main() {
    if (gft_init(0) != 0) {
        std::printf("init failed: %s\n", gft_last_error());
        return 1;
    }
    size_t sh[2] = {8, 8}, dg[2] = {8, 8};
    std::vector<double> h(64, 0.5);
    gft_poly* a = gft_from_host(h.data(), sh, dg, 2);
    gft_poly* b = gft_from_host(h.data(), sh, dg, 2);
    double two = 2.0, three = 3.0;
    gft_poly* s2 = gft_scalar(&two);
    gft_poly* s3 = gft_scalar(&three);
    const int N = 200000;
    std::printf("tensor + tensor (8x8)      : %6.2f us/call\n", per_call_us(N, [&] { gft_free(gft_add(a, b)); }));
    std::printf("tensor + lazy scalar       : %6.2f us/call\n", per_call_us(N, [&] { gft_free(gft_add(a, s2)); }));
    std::printf("tensor * lazy scalar       : %6.2f us/call\n", per_call_us(N, [&] { gft_free(gft_mul(a, s2)); }));
    std::printf("lazy scalar * lazy scalar  : %6.2f us/call\n", per_call_us(N, [&] { gft_free(gft_mul(s2, s3)); }));
    std::printf("derivative(a, 1, 1)        : %6.2f us/call\n", per_call_us(N, [&] { gft_free(gft_derivative(a, 1, 1)); }));
    std::printf("truncate_to_degree_p1(a,4) : %6.2f us/call\n", per_call_us(N, [&] { gft_free(gft_truncate_to_degree_p1(a, 4)); }));
    std::printf("gft_scalar (lazy, no GPU)  : %6.2f us/call\n", per_call_us(N, [&] { gft_free(gft_scalar(&two)); }));
    std::printf("clone                      : %6.2f us/call\n", per_call_us(N, [&] { gft_free(gft_clone(a)); }));
    return 0;
}
