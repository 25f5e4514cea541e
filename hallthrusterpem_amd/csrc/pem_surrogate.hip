// pem_surrogate.hip -- batched prediction of a sparse-grid (Smolyak / combination-technique) Lagrange surrogate.
//
// What it stands in for: the surrogate evaluations inside `System.fit(num_refine=1000, ...)` / `System.predict`
// that scripts/fit_surr.py:101-116 and scripts/pem_v0/{monte_carlo,sobol,mcmc}.py drive (BASELINE.json configs[3]:
// "batched tensor-interpolant predict").  The interpolant lives in amisc (third-party, absent): parity UNPINNED.
// Stated formula (hallthrusterpem_amd/surrogate.py builds the tables):
//     f(t) = sum_beta c_beta * sum_{j} Y_beta[j] * prod_{d active in beta} l^{(level_d)}_{j_d}(t_d),   t in [-1, 1]^D
// level 0 has the single node 0 (basis 1); level l >= 1 has m = 2^l + 1 Chebyshev-Lobatto nodes t_j = -cos(pi j/(m-1))
// with barycentric weights (-1)^j (halved at both ends).  Each multi-index beta has at most 3 active dimensions of
// level <= 3 (every index of a Smolyak set of level <= 3).
//
// Shape of the work: per point, sum over a few hundred tiny tensor grids -- K = prod(m) <= 729 products against
// n_out <= 16 columns.  It is not a dense GEMM worth MFMA: the "A matrix" (basis products) is generated on the fly
// per point, n_out is skinny, and fp64 MFMA issues slower than fp64 FMA on this chip (pem_svd.hip header).  One
// lane per point; the index table and the node values are wave-uniform (scalar loads), the basis vectors sit in LDS.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAXA = PEM_SURR_MAX_ACTIVE;    // active dimensions per multi-index
constexpr int MAXM = 9;                       // nodes at level 3
constexpr int IDX_STRIDE = 2 + 2 * MAXA;      // per beta: n_active, value offset, dims[MAXA], levels[MAXA]

__device__ __forceinline__ int nodes_of(int level) { return level == 0 ? 1 : (1 << level) + 1; }

// Chebyshev-Lobatto nodes -cos(pi j / (m - 1)) of levels 1..3, concatenated (offsets 0, 3, 8)
__device__ const double LOBATTO_NODES[17] = {
    -1.0, 0.0, 1.0,
    -1.0, -0.70710678118654752440, 0.0, 0.70710678118654752440, 1.0,
    -1.0, -0.92387953251128675613, -0.70710678118654752440, -0.38268343236508977173, 0.0,
    0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613, 1.0};

// barycentric Lagrange basis of the m Chebyshev-Lobatto nodes at t, written to b[0..m) with stride `bs`
__device__ __forceinline__ void lobatto_basis(double t, int m, double* b, int bs) {
    if (m == 1) {
        b[0] = 1.0;
        return;
    }
    double sum = 0.0;
    int hit = -1;
    const double* node = LOBATTO_NODES + (m == 3 ? 0 : (m == 5 ? 3 : 8));
    for (int j = 0; j < m; ++j) {
        const double tj = node[j];
        const double w = ((j & 1) ? -1.0 : 1.0) * ((j == 0 || j == m - 1) ? 0.5 : 1.0);
        const double diff = t - tj;
        if (diff == 0.0) hit = j;
        const double q = w / diff;
        b[j * bs] = q;
        sum += q;
    }
    const double inv = 1.0 / sum;
    for (int j = 0; j < m; ++j) b[j * bs] = hit < 0 ? b[j * bs] * inv : (j == hit ? 1.0 : 0.0);
}

template <int NOUT>
__global__ __launch_bounds__(BLOCK) void sparse_predict_kernel(long long n, int n_beta, const int32_t* __restrict__ index,
                                                               const double* __restrict__ coef,
                                                               const double* __restrict__ values, int n_out,
                                                               const double* __restrict__ t, size_t ld,
                                                               double* __restrict__ out, size_t ld_out) {
    __shared__ double basis[MAXA * MAXM * BLOCK];          // [active dim][node][thread]
    const int tid = threadIdx.x;
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + tid; i < n; i += stride) {
        double acc[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) acc[o] = 0.0;
        for (int bi = 0; bi < n_beta; ++bi) {
            const int32_t* e = index + (size_t)bi * IDX_STRIDE;
            const int na = e[0];
            const double* val = values + (size_t)e[1] * n_out;
            int m[MAXA];
#pragma unroll
            for (int a = 0; a < MAXA; ++a) {
                m[a] = a < na ? nodes_of(e[2 + MAXA + a]) : 1;
                if (a < na) lobatto_basis(t[(size_t)e[2 + a] * ld + i], m[a], basis + a * MAXM * BLOCK + tid, BLOCK);
                else basis[a * MAXM * BLOCK + tid] = 1.0;
            }
            double part[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) part[o] = 0.0;
            int node = 0;
            for (int j0 = 0; j0 < m[0]; ++j0) {
                const double w0 = basis[(0 * MAXM + j0) * BLOCK + tid];
                for (int j1 = 0; j1 < m[1]; ++j1) {
                    const double w1 = w0 * basis[(1 * MAXM + j1) * BLOCK + tid];
                    for (int j2 = 0; j2 < m[2]; ++j2, ++node) {
                        const double w = w1 * basis[(2 * MAXM + j2) * BLOCK + tid];
                        const double* row = val + (size_t)node * n_out;
#pragma unroll
                        for (int o = 0; o < NOUT; ++o)
                            if (o < n_out) part[o] = fma(w, row[o], part[o]);
                    }
                }
            }
            const double c = coef[bi];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) acc[o] = fma(c, part[o], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
            if (o < n_out) out[(size_t)o * ld_out + i] = acc[o];
    }
}

}  // namespace

extern "C" int pem_sparse_predict_f64_dev(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef,
                                          const double* values, int n_out, const double* t, size_t ld, double* out,
                                          size_t ld_out, pem_stream_t stream) {
    if (n_dim < 1 || n_beta < 1 || n_out < 1 || n_out > 16)
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_sparse_predict: need n_dim, n_beta >= 1 and 1 <= n_out <= 16");
    if (n == 0) return PEM_OK;
    if (!index || !coef || !values || !t || !out) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sparse_predict: NULL array");
    if (ld < n || ld_out < n) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sparse_predict: leading dimension smaller than n");
    if (int rc = pem::check_device()) return rc;
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_out <= 4)
        hipLaunchKernelGGL(sparse_predict_kernel<4>, dim3((unsigned)blocks), dim3(BLOCK), 0, st, (long long)n, n_beta, index, coef, values, n_out, t, ld, out, ld_out);
    else if (n_out <= 8)
        hipLaunchKernelGGL(sparse_predict_kernel<8>, dim3((unsigned)blocks), dim3(BLOCK), 0, st, (long long)n, n_beta, index, coef, values, n_out, t, ld, out, ld_out);
    else
        hipLaunchKernelGGL(sparse_predict_kernel<16>, dim3((unsigned)blocks), dim3(BLOCK), 0, st, (long long)n, n_beta, index, coef, values, n_out, t, ld, out, ld_out);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}
