// Is a slot allocation `slot = atomicAdd(&s_n, 1u)` on an LDS counter, taken by a few lanes of a wave inside nested divergent
// branches of an unrolled streaming loop, ever lost?  (A version of csrc/pem_quantile.hip's compact_bracket_kernel that parked
// its hits in LDS this way lost about one hit in a thousand, more with hipcc's atomic optimizer on: profiles/quantile_pilot_r03.txt.
// Answer on one MI355X, ROCm 7.2.0, with and without the optimizer: no -- 1.2e6 hits, every counter and every parked key
// accounted for.  The pattern by itself is sound; what that kernel lost it lost for a reason of its own, which was not found
// before the parking was dropped for measuring no faster.)
// Every block streams `per_block` pseudo-random keys, parks those below `thresh` and afterwards checks (a) that the counter
// equals the number of hits the lanes counted in registers and (b) that the parked keys sum up to the hits' sum.
// hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_slot_alloc.hip -o /tmp/lsa && /tmp/lsa
// (add -mllvm -amdgpu-atomic-optimizer-strategy=None for the other build)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
constexpr int CAP = 4096;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int NQ>
__global__ __launch_bounds__(512) void park(const u64* __restrict__ data, long long per_block, u64 thresh, u64* __restrict__ bad, u64* __restrict__ totals) {
    __shared__ u64 s_key[CAP];
    __shared__ unsigned s_tg[CAP];
    __shared__ unsigned s_n;
    __shared__ u64 s_hits, s_sum;
    if (threadIdx.x == 0) { s_n = 0; s_hits = 0; s_sum = 0; }
    __syncthreads();
    u64 my_hits = 0, my_sum = 0;
    const u64* src = data + (long long)blockIdx.x * per_block;
    for (long long i0 = threadIdx.x; i0 < per_block; i0 += 512 * 8) {
        u64 x[8];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            ok[u] = i0 + u * 512 < per_block;
            x[u] = ok[u] ? src[i0 + u * 512] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (ok[u]) {
                const unsigned kh = (unsigned)(x[u] >> 32);
                bool near = false;
#pragma unroll
                for (int q = 0; q < NQ; ++q) near |= (kh ^ (0x9e3779b9u * q)) < (unsigned)(thresh >> 32) + 1u;
                if (near) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const u64 k = x[u] ^ ((u64)(0x9e3779b9u * q) << 32);
                        if (k < thresh) {
                            ++my_hits;
                            my_sum += k;
                            const unsigned slot = atomicAdd(&s_n, 1u);
                            if (slot < (unsigned)CAP) {
                                s_key[slot] = k;
                                s_tg[slot] = (unsigned)q;
                            }
                        }
                    }
                }
            }
        }
    }
    atomicAdd(&s_hits, my_hits);
    atomicAdd(&s_sum, my_sum);
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 parked_sum = 0;
        const unsigned n = s_n < (unsigned)CAP ? s_n : (unsigned)CAP;
        for (unsigned i = 0; i < n; ++i) parked_sum += s_key[i];
        if (s_n != s_hits) atomicAdd(&bad[0], 1ull);                       // counter != hits counted in registers
        if (s_n <= (unsigned)CAP && parked_sum != s_sum) atomicAdd(&bad[1], 1ull);   // a parked key missing or overwritten
        atomicAdd(&totals[0], s_hits);
        atomicAdd(&totals[1], (u64)s_n);
    }
}

__global__ void fill(u64* d, long long n, u64 seed) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) d[i] = mix(i + seed);
}

int main() {
    const int blocks = 512;
    const long long per_block = 1 << 18;
    u64 *data, *bad, *totals;
    hipMalloc(&data, sizeof(u64) * blocks * per_block);
    hipMalloc(&bad, 16);
    hipMalloc(&totals, 16);
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, data, blocks * per_block, 12345ull);
    for (double frac : {1e-4, 1e-3, 3e-3}) {
        const u64 thresh = (u64)(frac * 18446744073709551615.0);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(bad, 0, 16);
            hipMemset(totals, 0, 16);
            hipLaunchKernelGGL(park<3>, dim3(blocks), dim3(512), 0, 0, data, per_block, thresh, bad, totals);
            u64 hb[2], ht[2];
            hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
            hipMemcpy(ht, totals, 16, hipMemcpyDeviceToHost);
            printf("hit fraction %.0e per value and bracket: %llu hits counted, counter total %llu; blocks whose counter differs: %llu, whose parked keys differ: %llu\n",
                   frac, ht[0], ht[1], hb[0], hb[1]);
        }
    }
    return 0;
}
