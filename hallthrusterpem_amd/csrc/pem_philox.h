// pem_philox.h -- Philox4x32-10 counter-based generator and the prior transforms, shared by the stand-alone
// sampler (pem_sampler.hip) and the fused Monte-Carlo mode of the coupled kernel (pem_kernels.hip).
// Value (i, d) of a design = transform_d( u53( philox(counter = (i_lo, i_hi, d / 2, stream), key = seed) ) ).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_hip.h"

namespace pem {

struct Philox4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// 53-bit uniform in [0, 1) from two 32-bit words
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    return (double)((((uint64_t)(hi >> 5)) << 26) | (uint64_t)(lo >> 6)) * 0x1.0p-53;
}

// products and sums rounded separately (no FMA) so that a host restatement reproduces the design bit for bit
__device__ __forceinline__ double transform(int kind, double a, double b, double u) {
#pragma clang fp contract(off)
    switch (kind) {
        case PEM_DIST_LOGUNIFORM: return exp(2.302585092994045684 * (a + (b - a) * u));   // 10^(a + (b-a) u)
        case PEM_DIST_NORMAL: return a + b * normcdfinv(u);                              // mean a, std b
        default: return a + (b - a) * u;                                                  // uniform on [a, b)
    }
}

}  // namespace pem
