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
// Shape of the work: per point, sum over the grids of the combination -- K = prod(m) <= 729 products against n_out <= 16
// columns.  It is not a dense GEMM worth MFMA: the "A matrix" (basis products) is generated on the fly per point, n_out
// is skinny (3 pads to 16 columns on the 16x16x4 tile) and the fp64 matrix pipe shares its units with the VALU
// (DESIGN.md section 4.5).  One lane per point; everything about the grid -- its index entry, the node values -- is
// wave-uniform and comes through the scalar cache.
//
// Round 2 rewrite (357 -> see DESIGN.md section 4.7): the first version was latency-bound, not arithmetic-bound -- 22 cycles
// per instruction at two waves per SIMD: one scalar load + one LDS read + a wait per NODE, the point's coordinate re-read
// from global memory per grid and dimension, and m + 1 IEEE divisions per barycentric basis.  Now
//   * the Lagrange bases are taken in product form, l_j(t) = c_j prod_{i != j} (t - t_i) by prefix/suffix products:
//     4 m multiplies, no division, no special case at the nodes;
//   * the point's coordinates are staged in LDS once;
//   * the innermost active dimension of a grid stays in registers and its loop is unrolled (m is 1, 3, 5 or 9): one batch
//     of scalar loads and one wait per ROW of the grid instead of per node, and the row is contracted first
//     (sum-factorisation: n_out FMAs per node, the outer basis product is applied once per row);
//   * only the two outer dimensions' bases go through LDS: 36 KB + coordinates per workgroup instead of 55 KB.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAXA = PEM_SURR_MAX_ACTIVE;    // active dimensions per multi-index
constexpr int MAXM = 9;                       // nodes at level 3
constexpr int IDX_STRIDE = 2 + 2 * MAXA;      // per beta: n_active, value offset, dims[MAXA], levels[MAXA]
static_assert(MAXA == 3, "the kernel nests exactly three dimensions");

__device__ __forceinline__ int nodes_of(int level) { return level == 0 ? 1 : (1 << level) + 1; }

// Chebyshev-Lobatto nodes -cos(pi j / (m - 1)) of levels 1..3, concatenated (offsets 0, 3, 8), and the Lagrange
// denominators c_j = 1 / prod_{i != j} (t_j - t_i) OF THESE DOUBLES (60-digit arithmetic, then rounded)
__device__ const double LOBATTO_NODES[17] = {
    -1.0, 0.0, 1.0,
    -1.0, -0.70710678118654752440, 0.0, 0.70710678118654752440, 1.0,
    -1.0, -0.92387953251128675613, -0.70710678118654752440, -0.38268343236508977173, 0.0,
    0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613, 1.0};
__device__ const double LOBATTO_INVDEN[17] = {
    0x1.0000000000000p-1, -0x1.0000000000000p+0, 0x1.0000000000000p-1,
    0x1.0000000000001p+0, -0x1.0000000000000p+1, 0x1.fffffffffffffp+0, -0x1.0000000000000p+1, 0x1.0000000000001p+0,
    0x1.fffffffffffffp+2, -0x1.0000000000001p+4, 0x1.0000000000001p+4, -0x1.fffffffffffffp+3, 0x1.fffffffffffffp+3,
    -0x1.fffffffffffffp+3, 0x1.0000000000001p+4, -0x1.0000000000001p+4, 0x1.fffffffffffffp+2};

// Lagrange basis of the M Chebyshev-Lobatto nodes at t, product form
template <int M>
__device__ __forceinline__ void lobatto_basis(double t, double (&b)[M]) {
    if constexpr (M == 1) {
        b[0] = 1.0;
    } else {
        constexpr int off = M == 3 ? 0 : (M == 5 ? 3 : 8);
        double d[M];
#pragma unroll
        for (int j = 0; j < M; ++j) d[j] = t - LOBATTO_NODES[off + j];
        double pre = 1.0;            // prod_{i < j} d_i
#pragma unroll
        for (int j = 0; j < M; ++j) {
            b[j] = pre * LOBATTO_INVDEN[off + j];
            pre *= d[j];
        }
        double suf = 1.0;            // prod_{i > j} d_i
#pragma unroll
        for (int j = M - 1; j >= 0; --j) {
            b[j] *= suf;
            suf *= d[j];
        }
    }
}

// the same into LDS (stride BLOCK), m in {1, 3, 5, 9} wave-uniform
__device__ __forceinline__ void stage_basis(double t, int m, double* dst) {
    if (m == 3) {
        double b[3];
        lobatto_basis<3>(t, b);
#pragma unroll
        for (int j = 0; j < 3; ++j) dst[j * BLOCK] = b[j];
    } else if (m == 5) {
        double b[5];
        lobatto_basis<5>(t, b);
#pragma unroll
        for (int j = 0; j < 5; ++j) dst[j * BLOCK] = b[j];
    } else if (m == 9) {
        double b[9];
        lobatto_basis<9>(t, b);
#pragma unroll
        for (int j = 0; j < 9; ++j) dst[j * BLOCK] = b[j];
    } else {
        dst[0] = 1.0;
    }
}

// One grid: part[o] = sum_{j0, j1} b0[j0] b1[j1] (sum_{j2} b2[j2] Y[(j0 m1 + j1) M2 + j2][o]); acc += c part.
// NOUT: columns kept in registers; EXACT: n_out == NOUT (no guards, the row loads merge into wide scalar loads)
template <int NOUT, bool EXACT, int M2>
__device__ __forceinline__ void contract_grid(int m0, int m1, const double* b0, const double* b1, double t2,
                                              const double* __restrict__ val, int n_out, double c, double (&acc)[NOUT]) {
    double b2[M2];
    lobatto_basis<M2>(t2, b2);
    double part[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) part[o] = 0.0;
    const double* row = val;
    for (int j0 = 0; j0 < m0; ++j0) {
        const double w0 = b0[j0 * BLOCK];
        for (int j1 = 0; j1 < m1; ++j1) {
            const double w1 = w0 * b1[j1 * BLOCK];
            double tmp[NOUT];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) tmp[o] = 0.0;
#pragma unroll
            for (int j2 = 0; j2 < M2; ++j2) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
                    if (EXACT || o < n_out) tmp[o] = fma(b2[j2], row[j2 * n_out + o], tmp[o]);
            }
            row += M2 * n_out;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) part[o] = fma(w1, tmp[o], part[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = fma(c, part[o], acc[o]);
}

template <int NOUT, bool EXACT>
__global__ __launch_bounds__(BLOCK) void sparse_predict_kernel(long long n, int n_dim, int n_beta, const int32_t* __restrict__ index,
                                                               const double* __restrict__ coef,
                                                               const double* __restrict__ values, int n_out_arg,
                                                               const double* __restrict__ t, size_t ld,
                                                               double* __restrict__ out, size_t ld_out, int per_grid) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* basis = lds;                                    // [2 outer dims][MAXM][BLOCK]
    double* coord = lds + 2 * MAXM * BLOCK;                 // [n_dim][BLOCK]
    const int n_out = EXACT ? NOUT : n_out_arg;
    const int tid = threadIdx.x;
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long i0 = (long long)blockIdx.x * BLOCK; i0 < n; i0 += stride) {
        const long long i = i0 + tid < n ? i0 + tid : n - 1;        // a dead lane recomputes the last point and stores nothing
        for (int d = 0; d < n_dim; ++d) coord[d * BLOCK + tid] = t[(size_t)d * ld + i];
        double acc[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) acc[o] = 0.0;
        for (int bi = 0; bi < n_beta; ++bi) {
            const int32_t* e = index + (size_t)bi * IDX_STRIDE;
            const int na = e[0];
            const double* val = values + (size_t)e[1] * n_out;
            // right-align the active dimensions: (m0, m1, m2) = (1, 1, m) / (1, m, m') / (m, m', m'') gives the same node order
            // (a dimension with one node contributes no stride), and the innermost -- unrolled -- one is always an active one
            const int sh = MAXA - na;
            int m[MAXA], dim[MAXA];
#pragma unroll
            for (int a = 0; a < MAXA; ++a) {
                const bool on = a >= sh;
                dim[a] = on ? e[2 + (on ? a - sh : 0)] : 0;
                m[a] = on ? nodes_of(e[2 + MAXA + (on ? a - sh : 0)]) : 1;
            }
            // (this thread's slots only: no barrier, a wave's LDS traffic is executed in order)
            stage_basis(coord[dim[0] * BLOCK + tid], m[0], basis + tid);
            stage_basis(coord[dim[1] * BLOCK + tid], m[1], basis + MAXM * BLOCK + tid);
            const double t2 = coord[dim[2] * BLOCK + tid];
            const double c = coef[bi];
            const double* b0 = basis + tid;
            const double* b1 = basis + MAXM * BLOCK + tid;
            if (m[2] == 3) contract_grid<NOUT, EXACT, 3>(m[0], m[1], b0, b1, t2, val, n_out, c, acc);
            else if (m[2] == 5) contract_grid<NOUT, EXACT, 5>(m[0], m[1], b0, b1, t2, val, n_out, c, acc);
            else if (m[2] == 9) contract_grid<NOUT, EXACT, 9>(m[0], m[1], b0, b1, t2, val, n_out, c, acc);
            else contract_grid<NOUT, EXACT, 1>(m[0], m[1], b0, b1, t2, val, n_out, c, acc);     // the constant grid (beta = 0)
            if (per_grid) {      // every grid's own (coefficient-weighted) interpolant: out[bi][o][i]
                if (i0 + tid < n) {
#pragma unroll
                    for (int o = 0; o < NOUT; ++o)
                        if (EXACT || o < n_out) out[((size_t)bi * n_out + o) * ld_out + i] = acc[o];
                }
#pragma unroll
                for (int o = 0; o < NOUT; ++o) acc[o] = 0.0;
            }
        }
        if (per_grid) continue;
        if (i0 + tid < n) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
                if (EXACT || o < n_out) out[(size_t)o * ld_out + i] = acc[o];
        }
    }
}

template <int NOUT, bool EXACT>
void launch_predict(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef, const double* values, int n_out,
                    const double* t, size_t ld, double* out, size_t ld_out, int per_grid, hipStream_t st) {
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const size_t lds = (size_t)(2 * MAXM + n_dim) * BLOCK * sizeof(double);    // <= 100 KB at PEM_SURR_MAX_DIM
    if (lds > 64 * 1024) {
        static pem::LdsAttrOnce attr;
        (void)attr.ensure(reinterpret_cast<const void*>(sparse_predict_kernel<NOUT, EXACT>));      // a refusal shows as a launch error below
    }
    hipLaunchKernelGGL((sparse_predict_kernel<NOUT, EXACT>), dim3((unsigned)blocks), dim3(BLOCK), lds, st, (long long)n, n_dim, n_beta,
                       index, coef, values, n_out, t, ld, out, ld_out, per_grid);
}

}  // namespace

namespace {

int sparse_predict(const char* who, size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef, const double* values,
                   int n_out, const double* t, size_t ld, double* out, size_t ld_out, int per_grid, pem_stream_t stream) {
    if (n_dim < 1 || n_beta < 1 || n_out < 1 || n_out > 16)
        return pem::fail(PEM_ERR_INVALID_ARG, "%s: need n_dim, n_beta >= 1 and 1 <= n_out <= 16", who);
    if (n == 0) return PEM_OK;
    if (!index || !coef || !values || !t || !out) return pem::fail(PEM_ERR_INVALID_ARG, "%s: NULL array", who);
    if (ld < n || ld_out < n) return pem::fail(PEM_ERR_INVALID_ARG, "%s: leading dimension smaller than n", who);
    if (n_dim > PEM_SURR_MAX_DIM) return pem::fail(PEM_ERR_INVALID_ARG, "%s: n_dim <= %d", who, PEM_SURR_MAX_DIM);
    if (int rc = pem::check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define PEM_PREDICT(NOUT_, EXACT_) launch_predict<NOUT_, EXACT_>(n, n_dim, n_beta, index, coef, values, n_out, t, ld, out, ld_out, per_grid, st)
    switch (n_out) {
        case 1: PEM_PREDICT(1, true); break;
        case 2: PEM_PREDICT(2, true); break;
        case 3: PEM_PREDICT(3, true); break;
        case 4: PEM_PREDICT(4, true); break;
        default:
            if (n_out <= 8) PEM_PREDICT(8, false);
            else PEM_PREDICT(16, false);
    }
#undef PEM_PREDICT
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // namespace

extern "C" {

int pem_sparse_predict_f64_dev(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef, const double* values,
                               int n_out, const double* t, size_t ld, double* out, size_t ld_out, pem_stream_t stream) {
    return sparse_predict("pem_sparse_predict", n, n_dim, n_beta, index, coef, values, n_out, t, ld, out, ld_out, 0, stream);
}

int pem_sparse_grid_values_f64_dev(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef, const double* values,
                                   int n_out, const double* t, size_t ld, double* out, size_t ld_out, pem_stream_t stream) {
    return sparse_predict("pem_sparse_grid_values", n, n_dim, n_beta, index, coef, values, n_out, t, ld, out, ld_out, 1, stream);
}

}  // extern "C"
