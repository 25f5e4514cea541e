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
                                                     int swap_dim, double* __restrict__ out, size_t ld) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t g = first + (uint64_t)i;
        for (int d0 = 0; d0 < ndim; d0 += 2) {
            double u[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = d0 + h;
                const uint32_t st = stream + ((swap_dim == -2 || swap_dim == d) ? 1u : 0u);
                const Philox4 r = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(d0 >> 1), st, k0, k1);
                u[h] = h == 0 ? u53(r.x, r.y) : u53(r.z, r.w);
                if (mode == 1) {   // stratum pi_d(g) of n_total, jittered by u
                    const uint64_t cell = feistel_permute(g, n_total, half_bits, k0, k1 ^ st, (uint32_t)d);
                    u[h] = ((double)cell + u[h]) / (double)n_total;
                }
            }
            out[(size_t)d0 * ld + i] = transform(tab.kind[d0], tab.a[d0], tab.b[d0], u[0]);
            if (d0 + 1 < ndim) out[(size_t)(d0 + 1) * ld + i] = transform(tab.kind[d0 + 1], tab.a[d0 + 1], tab.b[d0 + 1], u[1]);
        }
    }
}

int launch(size_t n, uint64_t first, uint64_t seed, uint32_t stream, int ndim, const int32_t* kind, const double* a,
           const double* b, int mode, uint64_t n_total, int swap_dim, double* out, size_t ld, hipStream_t st) {
    if (ndim < 1 || ndim > MAXDIM) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sample: ndim must be in [1, %d]", MAXDIM);
    if (!kind || !a || !b || !out) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sample: NULL array");
    if (ld < n) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sample: leading dimension smaller than n");
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
                       mode, n_total, half_bits, swap_dim, out, ld);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // namespace

extern "C" {

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
