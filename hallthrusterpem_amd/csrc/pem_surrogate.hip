// pem_surrogate.hip -- batched prediction of a sparse-grid (Smolyak / combination-technique) Lagrange surrogate.
//
// What it stands in for: the surrogate evaluations inside `System.fit(num_refine=1000, ...)` / `System.predict`
// that scripts/fit_surr.py:101-116 and scripts/pem_v0/{monte_carlo,sobol,mcmc}.py drive (BASELINE.json configs[3]:
// "batched tensor-interpolant predict").  The interpolant lives in amisc (third-party, absent): parity UNPINNED.
// Stated formula (hallthrusterpem_amd/surrogate.py builds the tables):
//     f(t) = sum_beta c_beta * sum_{j} Y_beta[j] * prod_{d active in beta} l^{(level_d)}_{j_d}(t_d),   t in [-1, 1]^D
// level 0 has the single node 0 (basis 1); level l >= 1 has m = 2^l + 1 Chebyshev-Lobatto nodes t_j = -cos(pi j/(m-1))
// with barycentric weights (-1)^j (halved at both ends).  Each multi-index beta has at most PEM_SURR_MAX_ACTIVE = 5 active
// dimensions of level <= PEM_SURR_MAX_LEVEL = 4 (round 4; rounds 1-3: 3 and 3).
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
constexpr int MAXO = MAXA - 1;                // of which all but the innermost go through LDS
constexpr int MAXM = 17;                      // nodes at level 4
constexpr int IDX_STRIDE = 2 + 2 * MAXA;      // per beta: n_active, value offset, dims[MAXA], levels[MAXA]
static_assert(MAXA == 5 && PEM_SURR_MAX_LEVEL == 4, "the kernel nests four outer dimensions around the unrolled innermost one; tables up to level 4");

__device__ __forceinline__ int nodes_of(int level) { return level == 0 ? 1 : (1 << level) + 1; }

// Chebyshev-Lobatto nodes -cos(pi j / (m - 1)) of levels 1..4, concatenated (offsets 0, 3, 8, 17), and the Lagrange
// denominators c_j = 1 / prod_{i != j} (t_j - t_i) OF THESE DOUBLES (exact rational arithmetic, then rounded)
__device__ const double LOBATTO_NODES[34] = {
    -1.0, 0.0, 1.0,
    -1.0, -0.70710678118654752440, 0.0, 0.70710678118654752440, 1.0,
    -1.0, -0.92387953251128675613, -0.70710678118654752440, -0.38268343236508977173, 0.0,
    0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613, 1.0,
    -1.0, -0.9807852804032304, -0.9238795325112867, -0.8314696123025452, -0.7071067811865476, -0.5555702330196023, -0.38268343236508984,
    -0.19509032201612833, 0.0, 0.19509032201612833, 0.38268343236508984, 0.5555702330196023, 0.7071067811865476, 0.8314696123025452,
    0.9238795325112867, 0.9807852804032304, 1.0};
__device__ const double LOBATTO_INVDEN[34] = {
    0x1.0000000000000p-1, -0x1.0000000000000p+0, 0x1.0000000000000p-1,
    0x1.0000000000001p+0, -0x1.0000000000000p+1, 0x1.fffffffffffffp+0, -0x1.0000000000000p+1, 0x1.0000000000001p+0,
    0x1.fffffffffffffp+2, -0x1.0000000000001p+4, 0x1.0000000000001p+4, -0x1.fffffffffffffp+3, 0x1.fffffffffffffp+3,
    -0x1.fffffffffffffp+3, 0x1.0000000000001p+4, -0x1.0000000000001p+4, 0x1.fffffffffffffp+2,
    0x1.ffffffffffff8p+9, -0x1.ffffffffffffep+10, 0x1.0000000000003p+11, -0x1.0000000000005p+11, 0x1.0000000000003p+11, -0x1.0000000000001p+11,
    0x1.ffffffffffffdp+10, -0x1.ffffffffffff7p+10, 0x1.ffffffffffff4p+10, -0x1.ffffffffffff7p+10, 0x1.ffffffffffffdp+10, -0x1.0000000000001p+11,
    0x1.0000000000003p+11, -0x1.0000000000005p+11, 0x1.0000000000003p+11, -0x1.ffffffffffffep+10, 0x1.ffffffffffff8p+9};

// Lagrange basis of the M Chebyshev-Lobatto nodes at t, product form
template <int M>
__device__ __forceinline__ void lobatto_basis(double t, double (&b)[M]) {
    if constexpr (M == 1) {
        b[0] = 1.0;
    } else {
        constexpr int off = M == 3 ? 0 : (M == 5 ? 3 : (M == 9 ? 8 : 17));
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

// the same into LDS (stride BLOCK), m in {1, 3, 5, 9, 17} wave-uniform
__device__ __forceinline__ void stage_basis(double t, int m, double* dst) {
    if (m == 17) {
        double b[17];
        lobatto_basis<17>(t, b);
#pragma unroll
        for (int j = 0; j < 17; ++j) dst[j * BLOCK] = b[j];
        return;
    }
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

__device__ __forceinline__ void out_field_store(double* p, double v) { __builtin_nontemporal_store(v, p); }

// One grid: part[o] = sum over the outer dimensions' nodes of (prod_a b_a[j_a]) (sum_{j} bI[j] Y[row M + j][o]); acc += c part.
// Four outer loops, the unused ones (a grid with fewer active dimensions is right-aligned) with one node and weight 1: their
// trip count of one costs a compare per level.  NOUT: columns kept in registers; EXACT: n_out == NOUT (no guards, the row
// loads merge into wide scalar loads).  `outer`: this thread's LDS slots, dimension a at a * ostride, node j at j * BLOCK.
template <int NOUT, bool EXACT, int MI>
__device__ __forceinline__ void contract_grid(const int (&m)[MAXO], const double* outer, int ostride, double ti,
                                              const double* __restrict__ val, int n_out, double c, double (&acc)[NOUT]) {
    double bi[MI];
    lobatto_basis<MI>(ti, bi);
    double part[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) part[o] = 0.0;
    const double* row = val;
    for (int j0 = 0; j0 < m[0]; ++j0) {
        const double w0 = m[0] > 1 ? outer[j0 * BLOCK] : 1.0;
        for (int j1 = 0; j1 < m[1]; ++j1) {
            const double w1 = m[1] > 1 ? w0 * outer[ostride + j1 * BLOCK] : w0;
            for (int j2 = 0; j2 < m[2]; ++j2) {
                const double w2 = m[2] > 1 ? w1 * outer[2 * ostride + j2 * BLOCK] : w1;
                for (int j3 = 0; j3 < m[3]; ++j3) {
                    const double w3 = m[3] > 1 ? w2 * outer[3 * ostride + j3 * BLOCK] : w2;
                    double tmp[NOUT];
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) tmp[o] = 0.0;
#pragma unroll
                    for (int j = 0; j < MI; ++j) {
#pragma unroll
                        for (int o = 0; o < NOUT; ++o)
                            if (EXACT || o < n_out) tmp[o] = fma(bi[j], row[j * n_out + o], tmp[o]);
                    }
                    row += MI * n_out;
#pragma unroll
                    for (int o = 0; o < NOUT; ++o) part[o] = fma(w3, tmp[o], part[o]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = fma(c, part[o], acc[o]);
}

// the fused reconstruction (round 4): outputs lat0 .. lat0 + rank - 1 of the prediction are the SVD latent coefficients of a field
// (scripts/pem_v0/pem_v0_SPT-100.yml:273-280: `j_ion`, log10 norm), and the field itself -- denorm(latent @ basis^T), dof values per
// point, row-major -- leaves with the prediction instead of through a second pass (pem_svd_reconstruct_f64_dev)
struct Recon {
    const double* basis;     // [dof][rank]
    double* field;           // [n][dof]
    int dof, rank, lat0, norm;
    double scale;
};

template <int NOUT, bool EXACT>
__global__ __launch_bounds__(BLOCK) void sparse_predict_kernel(long long n, int n_dim, int n_beta, const int32_t* __restrict__ index,
                                                               const double* __restrict__ coef,
                                                               const double* __restrict__ values, int n_out_arg,
                                                               const double* __restrict__ t, size_t ld,
                                                               double* __restrict__ out, size_t ld_out, int per_grid, int max_outer,
                                                               int max_m, int basis_words, Recon rc) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ostride = max_m * BLOCK;
    double* basis = lds;                                    // [max_outer][max_m][BLOCK]; later: the latents [BLOCK][rank]
    double* coord = lds + (size_t)basis_words * BLOCK;      // [n_dim][BLOCK]  (basis_words = max(max_outer max_m, rank))
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
            // right-align the active dimensions: (1, .., 1, m) / (1, .., m, m') / ... gives the same node order (a dimension with one
            // node contributes no stride), and the innermost -- unrolled -- one is always an active one.  The outer dimensions' LDS
            // slots are right-aligned inside the max_outer slots the launch has room for.
            const int sh = MAXA - na;
            int m[MAXO];
            const double* slots = basis + tid - (MAXO - max_outer) * ostride;     // slot a of this thread (only active ones are touched)
#pragma unroll
            for (int a = 0; a < MAXO; ++a) {
                const bool on = a >= sh;
                m[a] = on ? nodes_of(e[2 + MAXA + (on ? a - sh : 0)]) : 1;
                // (this thread's slots only: no barrier, a wave's LDS traffic is executed in order)
                if (on) stage_basis(coord[e[2 + a - sh] * BLOCK + tid], m[a], const_cast<double*>(slots) + a * ostride);
            }
            const int mi = na > 0 ? nodes_of(e[2 + MAXA + na - 1]) : 1;
            const double ti = na > 0 ? coord[e[2 + na - 1] * BLOCK + tid] : 0.0;
            const double c = coef[bi];
            if (mi == 3) contract_grid<NOUT, EXACT, 3>(m, slots, ostride, ti, val, n_out, c, acc);
            else if (mi == 5) contract_grid<NOUT, EXACT, 5>(m, slots, ostride, ti, val, n_out, c, acc);
            else if (mi == 9) contract_grid<NOUT, EXACT, 9>(m, slots, ostride, ti, val, n_out, c, acc);
            else if (mi == 17) contract_grid<NOUT, EXACT, 17>(m, slots, ostride, ti, val, n_out, c, acc);
            else contract_grid<NOUT, EXACT, 1>(m, slots, ostride, ti, val, n_out, c, acc);     // the constant grid (beta = 0)
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
        if (rc.field) {
            // the latents of this workgroup's 256 points through LDS (the basis slots are free again), then every WAVE rebuilds the
            // dof values of its 64 points row by row: lane = field index, 512 contiguous bytes per store
            __syncthreads();                                 // (another wave may still read its basis slots)
            double* lat = lds;                               // [BLOCK][rank]
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
                if (o >= rc.lat0 && o < rc.lat0 + rc.rank) lat[tid * rc.rank + (o - rc.lat0)] = acc[o];
            __syncthreads();
            const int wave = tid >> 6, lane = tid & 63;
            for (int k0 = 0; k0 < rc.dof; k0 += 64) {
                const int k = k0 + lane;
                double bk[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) bk[q] = (k < rc.dof && q < rc.rank) ? rc.basis[(size_t)k * rc.rank + q] : 0.0;
                for (int r = 0; r < 64; ++r) {
                    const long long p = i0 + wave * 64 + r;
                    if (p >= n) break;
                    const double* lr = lat + (wave * 64 + r) * rc.rank;
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (q < rc.rank) v = fma(lr[q], bk[q], v);
                    if (rc.norm == PEM_NORM_LOG10) v = exp10(v);
                    else if (rc.norm == PEM_NORM_LINEAR) v = v / rc.scale;
                    if (k < rc.dof) out_field_store(rc.field + (size_t)p * rc.dof + k, v);
                }
            }
            __syncthreads();                                 // the next batch of points stages its bases over the latents
        }
    }
}

template <int NOUT, bool EXACT>
void launch_predict(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef, const double* values, int n_out,
                    const double* t, size_t ld, double* out, size_t ld_out, int per_grid, int max_outer, int max_m, const Recon& rc,
                    hipStream_t st) {
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 8) blocks = 256 * 8;
    int basis_words = max_outer * max_m;                                           // doubles per thread: bases (or latents) | coordinates
    if (rc.field && rc.rank > basis_words) basis_words = rc.rank;
    const size_t lds = (size_t)(basis_words + n_dim) * BLOCK * sizeof(double);
    if (lds > 64 * 1024) {
        static pem::LdsAttrOnce attr;
        (void)attr.ensure(reinterpret_cast<const void*>(sparse_predict_kernel<NOUT, EXACT>));      // a refusal shows as a launch error below
    }
    hipLaunchKernelGGL((sparse_predict_kernel<NOUT, EXACT>), dim3((unsigned)blocks), dim3(BLOCK), lds, st, (long long)n, n_dim, n_beta,
                       index, coef, values, n_out, t, ld, out, ld_out, per_grid, max_outer, max_m, basis_words, rc);
}

}  // namespace

namespace {

int sparse_predict(const char* who, size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef, const double* values,
                   int n_out, const double* t, size_t ld, double* out, size_t ld_out, int per_grid, int max_active, int max_level,
                   const Recon& rc, pem_stream_t stream) {
    if (n_dim < 1 || n_beta < 1 || n_out < 1 || n_out > 16)
        return pem::fail(PEM_ERR_INVALID_ARG, "%s: need n_dim, n_beta >= 1 and 1 <= n_out <= 16", who);
    if (max_active < 0 || max_active > PEM_SURR_MAX_ACTIVE || max_level < 0 || max_level > PEM_SURR_MAX_LEVEL)
        return pem::fail(PEM_ERR_INVALID_ARG, "%s: at most %d active dimensions of level <= %d per multi-index", who, PEM_SURR_MAX_ACTIVE,
                         PEM_SURR_MAX_LEVEL);
    if (n == 0) return PEM_OK;
    if (!index || !coef || !values || !t || !out) return pem::fail(PEM_ERR_INVALID_ARG, "%s: NULL array", who);
    if (ld < n || ld_out < n) return pem::fail(PEM_ERR_INVALID_ARG, "%s: leading dimension smaller than n", who);
    if (n_dim > PEM_SURR_MAX_DIM) return pem::fail(PEM_ERR_INVALID_ARG, "%s: n_dim <= %d", who, PEM_SURR_MAX_DIM);
    if (rc.field && (rc.rank < 1 || rc.rank > 16 || rc.lat0 < 0 || rc.lat0 + rc.rank > n_out || rc.dof < 1 || !rc.basis || per_grid))
        return pem::fail(PEM_ERR_INVALID_ARG, "%s: the reconstructed field takes 1 <= rank <= 16 latent outputs lat0 .. lat0 + rank - 1 of the n_out", who);
    // the LDS of the outer dimensions' bases is sized by what the table really holds (the caller says): a level-3, three-dimension
    // table keeps rounds 1-3's 36 KB, four outer dimensions of 17 nodes take 139 KB
    const int max_outer = max_active > 1 ? max_active - 1 : 0, max_m = max_level == 0 ? 1 : (1 << max_level) + 1;
    if ((size_t)(max_outer * max_m + n_dim) * BLOCK * sizeof(double) > 160 * 1024)
        return pem::fail(PEM_ERR_INVALID_ARG, "%s: %d outer dimensions of %d nodes and %d coordinates do not fit the LDS", who, max_outer, max_m, n_dim);
    if (int rc0 = pem::check_device()) return rc0;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define PEM_PREDICT(NOUT_, EXACT_) \
    launch_predict<NOUT_, EXACT_>(n, n_dim, n_beta, index, coef, values, n_out, t, ld, out, ld_out, per_grid, max_outer, max_m, rc, st)
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
                               int n_out, const double* t, size_t ld, double* out, size_t ld_out, int max_active, int max_level,
                               pem_stream_t stream) {
    return sparse_predict("pem_sparse_predict", n, n_dim, n_beta, index, coef, values, n_out, t, ld, out, ld_out, 0, max_active, max_level,
                          Recon{}, stream);
}

int pem_sparse_grid_values_f64_dev(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef, const double* values,
                                   int n_out, const double* t, size_t ld, double* out, size_t ld_out, int max_active, int max_level,
                                   pem_stream_t stream) {
    return sparse_predict("pem_sparse_grid_values", n, n_dim, n_beta, index, coef, values, n_out, t, ld, out, ld_out, 1, max_active,
                          max_level, Recon{}, stream);
}

int pem_sparse_predict_field_f64_dev(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef, const double* values,
                                     int n_out, const double* t, size_t ld, double* out, size_t ld_out, int max_active, int max_level,
                                     int lat0, int rank, int dof, int norm, double norm_scale, const double* basis, double* field,
                                     pem_stream_t stream) {
    if (!field) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sparse_predict_field: NULL field");
    if (norm != PEM_NORM_NONE && norm != PEM_NORM_LOG10 && norm != PEM_NORM_LINEAR) return pem::fail(PEM_ERR_INVALID_ARG, "pem_sparse_predict_field: unknown norm %d", norm);
    Recon rc{};
    rc.basis = basis;
    rc.field = field;
    rc.dof = dof;
    rc.rank = rank;
    rc.lat0 = lat0;
    rc.norm = norm;
    rc.scale = norm_scale;
    return sparse_predict("pem_sparse_predict_field", n, n_dim, n_beta, index, coef, values, n_out, t, ld, out, ld_out, 0, max_active, max_level,
                          rc, stream);
}

}  // extern "C"
