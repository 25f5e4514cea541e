// pem_sampler.hip -- counter-based input samplers for the Monte-Carlo / Latin-hypercube / Saltelli loops.
//
// What it stands in for: `system.sample_inputs(N, normalize=True, use_pdf=[...])` of scripts/gen_data.py:238
// and the `(Ns, Nx)` sampling of scripts/pem_v0/monte_carlo.py:63-300 / sobol.py:46-66.  Those live in amisc /
// uqtils (third-party, absent from the reference tree), so parity is UNPINNED here: the formulas below are this
// library's own and are checked against a numpy restatement (oracle/sampler_np.py), not against the reference.
//
// Every value is a pure function of (seed, stream, global sample index, dimension): Philox4x32-10 with
//   counter = (index_lo, index_hi, dimension pair, stream), key = (seed_lo, seed_hi),
// so a batch can be generated in any sharding over GPUs and any batch size and is always the same batch.
// One Philox call yields 4 x 32 bits = two 53-bit uniforms = dimensions 2b and 2b+1.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"
#include "pem_philox.h"

namespace {

constexpr int MAXDIM = PEM_SAMPLE_MAX_DIM;

struct DimTable {
    int32_t kind[MAXDIM];
    double a[MAXDIM], b[MAXDIM];
};

using pem::Philox4;
using pem::philox4x32_10;
using pem::transform;
using pem::u53;

// out of line: inlined, the library exp / normcdfinv bodies cost the kernel 438 registers (one wave per SIMD)
__device__ __attribute__((noinline)) double transform_call(int kind, double a, double b, double u) { return transform(kind, a, b, u); }

// keyed bijection of [0, n): 4-round Feistel network on 2*h bits with cycle walking (Latin-hypercube strata)
__device__ __forceinline__ uint64_t feistel_permute(uint64_t i, uint64_t n, int half_bits, uint32_t k0, uint32_t k1,
                                                    uint32_t dim) {
    const uint64_t mask = (1ull << half_bits) - 1;
    do {
        uint64_t l = i >> half_bits, r = i & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const Philox4 f = philox4x32_10((uint32_t)r, (uint32_t)(r >> 32), dim, 0x4C485300u + round, k0, k1);
            const uint64_t t = l ^ ((((uint64_t)f.y << 32) | f.x) & mask);
            l = r;
            r = t;
        }
        i = (l << half_bits) | r;
    } while (i >= n);
    return i;
}

// mode: 0 Monte-Carlo, 1 Latin hypercube over n_total strata.  swap_dim: Saltelli blocks -- dimension `swap_dim`
// (or every dimension if swap_dim == -2) is drawn from stream+1 instead of stream; -1 = plain.
__global__ __launch_bounds__(256) void sample_kernel(long long n, uint64_t first, uint64_t seed, uint32_t stream,
                                                     int ndim, DimTable tab, int mode, uint64_t n_total, int half_bits,
                                                     int swap_dim, double* __restrict__ out, size_t ld, int tiled) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t g = first + (uint64_t)i;
        // SoA: out[d][i] with leading dimension ld.  Tile-interleaved (pem_sample_tiled_f64_dev): [i / 64][d][i % 64], the
        // layout pem_coupled_tiled_f64_dev reads -- one contiguous ndim x 512-byte block per 64-sample tile.
        double* row = tiled ? out + (size_t)(i >> 6) * ((size_t)ndim * 64) + (size_t)(i & 63) : out + i;
        const size_t dstride = tiled ? 64 : ld;
        for (int d0 = 0; d0 < ndim; d0 += 2) {
            // one Philox call serves both dimensions of the pair unless a Saltelli block draws them from different
            // streams (a wave-uniform choice)
            const uint32_t st0 = stream + ((swap_dim == -2 || swap_dim == d0) ? 1u : 0u);
            const uint32_t st1 = stream + ((swap_dim == -2 || swap_dim == d0 + 1) ? 1u : 0u);
            const Philox4 r0 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(d0 >> 1), st0, k0, k1);
            Philox4 r1 = r0;
            if (st1 != st0) r1 = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(d0 >> 1), st1, k0, k1);
            double u[2] = {u53(r0.x, r0.y), u53(r1.z, r1.w)};
            if (mode == 1) {   // stratum pi_d(g) of n_total, jittered by u
                u[0] = ((double)feistel_permute(g, n_total, half_bits, k0, k1 ^ st0, (uint32_t)d0) + u[0]) / (double)n_total;
                if (d0 + 1 < ndim)
                    u[1] = ((double)feistel_permute(g, n_total, half_bits, k0, k1 ^ st1, (uint32_t)(d0 + 1)) + u[1]) / (double)n_total;
            }
            row[(size_t)d0 * dstride] = transform_call(tab.kind[d0], tab.a[d0], tab.b[d0], u[0]);
            if (d0 + 1 < ndim) row[(size_t)(d0 + 1) * dstride] = transform_call(tab.kind[d0 + 1], tab.a[d0 + 1], tab.b[d0 + 1], u[1]);
        }
    }
}

int launch(size_t n, uint64_t first, uint64_t seed, uint32_t stream, int ndim, const int32_t* kind, const double* a,
           const double* b, int mode, uint64_t n_total, int swap_dim, double* out, size_t ld, hipStream_t st, int tiled = 0) {
    if (ndim < 1 || ndim > MAXDIM) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sample: ndim must be in [1, %d]", MAXDIM);
    if (!kind || !a || !b || !out) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sample: NULL array");
    if (!tiled && ld < n) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sample: leading dimension smaller than n");
    if (swap_dim < -2 || swap_dim >= ndim) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sample: swap_dim out of range");
    if (n == 0) return PEM_OK;
    if (int rc = pem::check_device()) return rc;
    DimTable tab;
    for (int d = 0; d < MAXDIM; ++d) {
        tab.kind[d] = d < ndim ? kind[d] : 0;
        tab.a[d] = d < ndim ? a[d] : 0.0;
        tab.b[d] = d < ndim ? b[d] : 0.0;
        if (d < ndim && (kind[d] < 0 || kind[d] > PEM_DIST_NORMAL))
            return pem::fail(PEM_ERR_INVALID_ARG, "pem_sample: unknown distribution kind %d for dimension %d", kind[d], d);
    }
    int half_bits = 1;
    if (mode == 1) {
        if (n_total == 0 || first + n > n_total) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sample_lhs: indices exceed n_total");
        while ((1ull << (2 * half_bits)) < n_total) ++half_bits;
    }
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (long long)n, first, seed, stream, ndim, tab,
                       mode, n_total, half_bits, swap_dim, out, ld, tiled);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // namespace

extern "C" {

int pem_sample_tiled_f64_dev(size_t n, uint64_t first_index, uint64_t seed, uint32_t stream_id, int ndim, const int32_t* kind,
                             const double* a, const double* b, int swap_dim, double* out, pem_stream_t stream) {
    return launch(n, first_index, seed, stream_id, ndim, kind, a, b, 0, 0, swap_dim, out, 0, static_cast<hipStream_t>(stream), 1);
}

int pem_sample_f64_dev(size_t n, uint64_t first_index, uint64_t seed, uint32_t stream_id, int ndim, const int32_t* kind,
                       const double* a, const double* b, int swap_dim, double* out, size_t ld, pem_stream_t stream) {
    return launch(n, first_index, seed, stream_id, ndim, kind, a, b, 0, 0, swap_dim, out, ld, static_cast<hipStream_t>(stream));
}

int pem_sample_lhs_f64_dev(size_t n, uint64_t first_index, uint64_t n_total, uint64_t seed, uint32_t stream_id, int ndim,
                           const int32_t* kind, const double* a, const double* b, double* out, size_t ld,
                           pem_stream_t stream) {
    return launch(n, first_index, seed, stream_id, ndim, kind, a, b, 1, n_total, -1, out, ld, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Saltelli accumulation: the per-batch sums behind the Sobol' estimators of hallthrusterpem_amd/drivers.py, one pass.
//   fAB == NULL : out[b][q][0] += sum f_A + f_B,   out[b][q][1] += sum f_A^2 + f_B^2          (mean / variance)
//   fAB != NULL : out[b][q][0] += sum f_B (f_AB - f_A),  out[b][q][1] += sum (f_A - f_AB)^2    (S1_d / ST_d)
// Inputs are [nq][ld] row-major; every workgroup writes its own partial (deterministic), summed by the caller.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int SOBOL_BLOCK = 256;
constexpr int SOBOL_MAXQ = 8;

__global__ __launch_bounds__(SOBOL_BLOCK) void sobol_partial_kernel(long long m, int nq, size_t ld,
                                                                    const double* __restrict__ fA,
                                                                    const double* __restrict__ fB,
                                                                    const double* __restrict__ fAB,
                                                                    double* __restrict__ partial) {
    __shared__ double red[2 * SOBOL_MAXQ][SOBOL_BLOCK / 64];
    double s0[SOBOL_MAXQ], s1[SOBOL_MAXQ];
#pragma unroll
    for (int q = 0; q < SOBOL_MAXQ; ++q) s0[q] = s1[q] = 0.0;
    const long long stride = (long long)gridDim.x * SOBOL_BLOCK;
    for (long long i = (long long)blockIdx.x * SOBOL_BLOCK + threadIdx.x; i < m; i += stride) {
#pragma unroll
        for (int q = 0; q < SOBOL_MAXQ; ++q) {
            if (q < nq) {
                const double a = fA[(size_t)q * ld + i], b = fB[(size_t)q * ld + i];
                if (fAB) {
                    const double ab = fAB[(size_t)q * ld + i];
                    s0[q] = fma(b, ab - a, s0[q]);
                    s1[q] = fma(a - ab, a - ab, s1[q]);
                } else {
                    s0[q] += a + b;
                    s1[q] = fma(a, a, fma(b, b, s1[q]));
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < SOBOL_MAXQ; ++q) {
#pragma unroll
        for (int msk = 32; msk >= 1; msk >>= 1) {
            s0[q] += __shfl_xor(s0[q], msk);
            s1[q] += __shfl_xor(s1[q], msk);
        }
        if (lane == 0) {
            red[2 * q][wave] = s0[q];
            red[2 * q + 1][wave] = s1[q];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * nq) {
        double t = 0.0;
        for (int w = 0; w < SOBOL_BLOCK / 64; ++w) t += red[threadIdx.x][w];
        partial[(size_t)blockIdx.x * 2 * nq + threadIdx.x] = t;      // [block][q][2]
    }
}

}  // namespace

extern "C" int pem_sobol_partial_f64_dev(size_t m, int nq, size_t ld, const double* fA, const double* fB, const double* fAB,
                                         double* partial, int n_blocks, pem_stream_t stream) {
    if (nq < 1 || nq > SOBOL_MAXQ) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sobol_partial: 1 <= nq <= %d", SOBOL_MAXQ);
    if (n_blocks < 1 || !fA || !fB || !partial || ld < m) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sobol_partial: bad arguments");
    if (int rc = pem::check_device()) return rc;
    hipLaunchKernelGGL(sobol_partial_kernel, dim3((unsigned)n_blocks), dim3(SOBOL_BLOCK), 0, static_cast<hipStream_t>(stream),
                       (long long)m, nq, ld, fA, fB, fAB, partial);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}
