// Helpers shared by the translation units of the structural kernels (gft_kernels.hip, gft_horner.hip): launch grids, the sticky
// witness word, a workgroup's copy of its batch item.  Internal to genfer_amd/csrc.
#pragma once
#include "gft_kernels.hpp"

namespace gft {

static inline unsigned grid_for(size_t n, unsigned block = 256) {
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;  // 256 CUs x 8 blocks; the rest is grid-strided
    return (unsigned)g;
}

// Raising a sticky witness word (read by a LATER kernel on the same stream, k_witness_verdict).  Many waves / workgroups
// raise the same word, and write-through stores to one address — agent-scope atomics, and `volatile` stores, which hipcc
// also emits with sc0 sc1 — serialise in the memory system at ~26 ns each: 2048 workgroups of k_conv_shallow made a 13 us
// kernel take 60 us, one store per wave 217 us.  So: an ORDINARY (L2 write-back) store — stores of the one value ever
// written merge in each XCD's L2 and reach memory at the end of the kernel.  wit_raise_once first looks (an ordinary load:
// within an XCD the first store makes every later load hit in L2); the Horner pipelines store without looking — a load
// would have to be waited for, and with it their whole prefetch ring.
__device__ inline void wit_raise(unsigned* w) { *w = 1u; }
__device__ inline void wit_raise_once(unsigned* w) {
    if (*w == 0u) *w = 1u;
}

// A workgroup's copy of ITS item of a batch (gft_kernels.hpp ObsItem): blockIdx.y = item
template <class IT>
__device__ __forceinline__ const IT& item_to_lds(const IT* __restrict__ items, unsigned char* lds) {
    static_assert(sizeof(IT) % 16 == 0, "batch items are copied in 16-byte pieces");
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u* src = reinterpret_cast<const v4u*>(items + blockIdx.y);
    for (unsigned i = threadIdx.x; i < sizeof(IT) / 16; i += blockDim.x) {
        const v4u t = src[i];
        reinterpret_cast<v4u*>(lds)[i] = t;
    }
    __syncthreads();
    return *reinterpret_cast<const IT*>(lds);
}

}  // namespace gft
