// pem_likelihood.hip -- Gaussian log-likelihood of measured ion current density given model profiles.
//
// What it stands in for: the `jion` branch of `spt100_log_likelihood` (scripts/pem_v0/mcmc.py:57-106): model
// profiles on the 91-point grid are interpolated linearly to the measurement angles on the mirrored grid
// (monte_carlo.py:265-270; the commented prototype in src/hallmd/models/plume.py:142-149) and compared with the
// data, `sum(-0.5 * ((ye - y) / std)**2)`.  Those scripts import modules that no longer exist (SURVEY.md section 2
// row 12) and nothing in the reference tests pins them: parity UNPINNED, the formula is stated here:
//     ll[i] = sum_a -0.5 * ((y[e][a] - J_i(|alpha[e][a]|)) * inv_std[e][a])^2 ,   e = i mod n_cond,
//     J_i(x) = (1 - w) j_ion[i][k] + w j_ion[i][k+1],  k = floor(x / h), w = x / h - k,  h = (pi/2) / 90.
// The host precomputes (k, w) per measurement; the kernel is one pass over j_ion (HBM-bound, 728 B per sample).
// The (k, w, y, inv_std) tables (<= 4096 measurements, 28 bytes each) are staged in LDS.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"

namespace {

constexpr int WAVES = 4;
constexpr int BLOCK = 64 * WAVES;
constexpr int NANG = PEM_NANGLE;
constexpr int TILE = 16 * NANG;          // doubles of a 16-sample tile (contiguous in j_ion)
constexpr int UN = (TILE / 2 + 63) / 64; // 16-byte pieces per lane (12)

typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A wave owns 16 consecutive samples per tile: their 16 x 91 profile values are one contiguous block, moved to LDS
// with 16-byte-per-lane loads (two tiles ahead in registers, as in svd_compress_kernel: a gather straight from
// global memory ran at 3.7 TB/s).  Lane (s = lane & 15, q = lane >> 4) then takes measurements q, q+4, ... of sample s
// from LDS; the four partial sums meet by shuffles.
__global__ __launch_bounds__(BLOCK) void jion_loglik_kernel(long long n, int n_cond, int n_ang,
                                                            const int32_t* __restrict__ kidx,
                                                            const double* __restrict__ wgt, const double* __restrict__ y,
                                                            const double* __restrict__ inv_std,
                                                            const double* __restrict__ j_ion, double* __restrict__ ll) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* tile = reinterpret_cast<double*>(smem_raw) + wave * (TILE + 2);
    // measurement tables in LDS: vmcnt is one in-order counter, so a table load from global memory inside the
    // loop would wait for every prefetched tile load issued before it (355 us per 1.25e6 samples that way)
    const int nent = n_cond * n_ang;
    double* tab_w = reinterpret_cast<double*>(smem_raw) + WAVES * (TILE + 2);
    double* tab_y = tab_w + nent;
    double* tab_is = tab_y + nent;
    int32_t* tab_k = reinterpret_cast<int32_t*>(tab_is + nent);
    for (int i = tid; i < nent; i += BLOCK) {
        tab_w[i] = wgt[i];
        tab_y[i] = y[i];
        tab_is[i] = inv_std[i];
        tab_k[i] = kidx[i];
    }
    __syncthreads();
    const int s = lane & 15, q = lane >> 4;
    const long long ntiles = (n + 15) / 16;
    const long long stride = (long long)gridDim.x * WAVES;
    const bool vec_ok = (((uintptr_t)j_ion) & 15) == 0;

    auto fetch = [&](long long t, f64x2 (&v)[UN]) {
        const double* tp = j_ion + t * TILE;
        const long long rest = (n - t * 16) * NANG;
        const int len = rest < TILE ? (int)rest : TILE;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = 2 * (u * 64 + lane);
            if (vec_ok && i + 1 < len) {
                v[u] = *reinterpret_cast<const f64x2*>(tp + i);
            } else {
                v[u].x = i < len ? tp[i] : 0.0;
                v[u].y = i + 1 < len ? tp[i + 1] : 0.0;
            }
        }
    };

    long long t = (long long)blockIdx.x * WAVES + wave;
    f64x2 nxt[UN], nxt2[UN];
    if (t < ntiles) fetch(t, nxt);
    if (t + stride < ntiles) fetch(t + stride, nxt2);
    for (; t < ntiles; t += stride) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = 2 * (u * 64 + lane);
            if (i < TILE) *reinterpret_cast<f64x2*>(tile + i) = nxt[u];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) nxt[u] = nxt2[u];
        if (t + 2 * stride < ntiles) fetch(t + 2 * stride, nxt2);
        wave_lds_sync();

        const long long i = t * 16 + s;
        const int base = (int)(i % n_cond) * n_ang;
        const double* row = tile + s * NANG;
        double acc = 0.0;
        for (int a = q; a < n_ang; a += 4) {
            const int k = tab_k[base + a];
            const double w = tab_w[base + a];
            const double lo = row[k], hi = row[k + 1];
            const double model = fma(w, hi - lo, lo);
            const double z = (tab_y[base + a] - model) * tab_is[base + a];
            acc = fma(-0.5 * z, z, acc);
        }
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        if (q == 0 && i < n) ll[i] = acc;
        wave_lds_sync();
    }
}

// ---- marginalisation over nuisance draws (mcmc.py:100-107) and the prior (mcmc.py:110-121) ----------------------
// One workgroup per chain k: s[m] = sum_e ll[k][m][e] (+ the discharge-current weight of mcmc.py:102-104 with the
// test double's I_d = (q/m_i) mdot_a / (1 - 2 a_1), tests/sim_hallthruster.jl:35-40), then a streaming
// log-sum-exp over m held as (max, sum of exp(s - max)) pairs.  NaN in any s makes the result NaN, all -inf gives -inf
// -- as numpy's max-shifted form in the reference.  With `log_prior` given the output is the log posterior:
// prior + likelihood, -inf where the prior is -inf or the likelihood NaN.
struct Lse {
    double mx, acc;
};

__device__ __forceinline__ Lse lse_push(Lse a, double s) {
    if (s > a.mx) {
        a.acc = (a.acc == 0.0 ? 0.0 : a.acc * exp(a.mx - s)) + 1.0;
        a.mx = s;
    } else if (s != -__builtin_inf()) {
        a.acc += exp(s - a.mx);   // NaN s lands here and poisons acc
    }
    return a;
}

__device__ __forceinline__ Lse lse_merge(Lse a, Lse b) {
    const double m = fmax(a.mx, b.mx);
    const double ta = a.acc == 0.0 ? 0.0 : a.acc * exp(a.mx - m);
    const double tb = b.acc == 0.0 ? 0.0 : b.acc * exp(b.mx - m);
    return Lse{m, ta + tb};
}

__global__ __launch_bounds__(BLOCK) void loglik_marginal_kernel(int n_draws, int n_cond, const double* __restrict__ ll,
                                                                const double* __restrict__ mdot_a,
                                                                const double* __restrict__ a_1, double discharge,
                                                                double inv_sigma, const double* __restrict__ log_prior,
                                                                double* __restrict__ out) {
    __shared__ Lse part[WAVES];
    const long long base = (long long)blockIdx.x * n_draws * n_cond;
    Lse st{-__builtin_inf(), 0.0};
    for (int m = threadIdx.x; m < n_draws; m += BLOCK) {
        const long long row = base + (long long)m * n_cond;
        double s = 0.0;
        for (int e = 0; e < n_cond; ++e) s += ll[row + e];
        if (mdot_a) {
            double w = 0.0;
            for (int e = 0; e < n_cond; ++e) {
                const double i_d = (1.6e-19 / 2.18e-25) * mdot_a[row + e] / (1.0 - a_1[row + e] * 2.0);
                const double z = (discharge - i_d) * inv_sigma;
                w = fma(-0.5 * z, z, w);
            }
            s += w;
        }
        st = lse_push(st, s);
    }
    for (int sh = 32; sh >= 1; sh >>= 1) st = lse_merge(st, Lse{__shfl_xor(st.mx, sh), __shfl_xor(st.acc, sh)});
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = st;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < WAVES; ++w) st = lse_merge(st, part[w]);
        double r = st.mx + log(st.acc);            // all -inf: -inf + log(0) = -inf
        if (st.acc != st.acc) r = st.acc;          // NaN
        if (log_prior) {
            const double lp = log_prior[blockIdx.x];
            r = (lp - lp == 0.0 && r == r) ? lp + r : -__builtin_inf();
        }
        out[blockIdx.x] = r;
    }
}

struct PriorTable {
    int32_t kind[PEM_SAMPLE_MAX_DIM];
    double a[PEM_SAMPLE_MAX_DIM], b[PEM_SAMPLE_MAX_DIM];
};

__global__ void log_prior_kernel(long long n, int ndim, PriorTable t, const double* __restrict__ theta,
                                 double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double lp = 0.0;
    for (int d = 0; d < ndim; ++d) {
        const double x = theta[i * ndim + d], a = t.a[d], b = t.b[d];
        double v;
        if (t.kind[d] == PEM_DIST_UNIFORM) {
            v = (x >= a && x <= b) ? -log(b - a) : -__builtin_inf();
        } else if (t.kind[d] == PEM_DIST_LOGUNIFORM) {   // density of 10^U(a, b): 1 / (x ln10 (b - a))
            v = (x >= pow(10.0, a) && x <= pow(10.0, b)) ? -log(x) - log(2.302585092994045684 * (b - a)) : -__builtin_inf();
        } else {
            const double z = (x - a) / b;
            v = -0.5 * z * z - log(b * 2.5066282746310002);
        }
        lp += v;
    }
    out[i] = lp;
}

}  // namespace

extern "C" int pem_loglik_marginal_f64_dev(size_t n_chains, int n_draws, int n_cond, const double* loglik,
                                           const double* mdot_a, const double* a_1, double discharge_current,
                                           double discharge_sigma, const double* log_prior, double* out,
                                           pem_stream_t stream) {
    if (n_draws < 1 || n_cond < 1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_loglik_marginal: need n_draws, n_cond >= 1");
    if (n_chains == 0) return PEM_OK;
    if (!loglik || !out || (mdot_a && !a_1)) return pem::fail(PEM_ERR_INVALID_ARG, "pem_loglik_marginal: NULL array");
    if (mdot_a && !(discharge_sigma > 0.0)) return pem::fail(PEM_ERR_INVALID_ARG, "pem_loglik_marginal: discharge_sigma must be > 0");
    if (n_chains > 0x7fffffffull) return pem::fail(PEM_ERR_INVALID_ARG, "pem_loglik_marginal: too many chains");
    if (int rc = pem::check_device()) return rc;
    hipLaunchKernelGGL(loglik_marginal_kernel, dim3((unsigned)n_chains), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                       n_draws, n_cond, loglik, mdot_a, a_1, discharge_current, mdot_a ? 1.0 / discharge_sigma : 0.0, log_prior,
                       out);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

extern "C" int pem_log_prior_f64_dev(size_t n, int ndim, const int32_t* kind, const double* a, const double* b,
                                     const double* theta, double* out, pem_stream_t stream) {
    if (ndim < 1 || ndim > PEM_SAMPLE_MAX_DIM) return pem::fail(PEM_ERR_INVALID_ARG, "pem_log_prior: 1 <= ndim <= %d", PEM_SAMPLE_MAX_DIM);
    if (!kind || !a || !b) return pem::fail(PEM_ERR_INVALID_ARG, "pem_log_prior: NULL prior table");
    PriorTable t{};
    for (int d = 0; d < ndim; ++d) {
        if (kind[d] < PEM_DIST_UNIFORM || kind[d] > PEM_DIST_NORMAL) return pem::fail(PEM_ERR_INVALID_ARG, "pem_log_prior: unknown distribution %d", kind[d]);
        t.kind[d] = kind[d];
        t.a[d] = a[d];
        t.b[d] = b[d];
    }
    if (n == 0) return PEM_OK;
    if (!theta || !out) return pem::fail(PEM_ERR_INVALID_ARG, "pem_log_prior: NULL array");
    if (int rc = pem::check_device()) return rc;
    hipLaunchKernelGGL(log_prior_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (long long)n, ndim, t, theta, out);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

extern "C" int pem_jion_loglik_f64_dev(size_t n, int n_cond, int n_ang, const int32_t* kidx, const double* weight,
                                       const double* y, const double* inv_std, const double* j_ion, double* loglik,
                                       pem_stream_t stream) {
    if (n_cond < 1 || n_ang < 1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_jion_loglik: need at least one condition and one angle");
    if (n == 0) return PEM_OK;
    if (!kidx || !weight || !y || !inv_std || !j_ion || !loglik) return pem::fail(PEM_ERR_INVALID_ARG, "pem_jion_loglik: NULL array");
    if (int rc = pem::check_device()) return rc;
    if ((long long)n_cond * n_ang > PEM_LOGLIK_MAX_MEASUREMENTS)
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_jion_loglik: more than %d measurements (n_cond * n_ang)", PEM_LOGLIK_MAX_MEASUREMENTS);
    size_t blocks = ((n + 15) / 16 + WAVES - 1) / WAVES;
    // persistent: >= 47 KB of LDS per workgroup.  (Grids of 2 / 4 x the resident workgroups or one tile group per wave, which help
    // the write-heavy kernels, measured 173 / 176 / 194 us against 171 here: the reads are prefetched two tiles deep, r03o.)
    if (blocks > 256 * 2) blocks = 256 * 2;
    const size_t lds = (size_t)WAVES * (TILE + 2) * 8 + (size_t)n_cond * n_ang * 28 + 16;
    static pem::LdsAttrOnce attr;
    HIP_TRY(attr.ensure(reinterpret_cast<const void*>(jion_loglik_kernel)));
    hipLaunchKernelGGL(jion_loglik_kernel, dim3((unsigned)blocks), dim3(BLOCK), lds, static_cast<hipStream_t>(stream),
                       (long long)n, n_cond, n_ang, kidx, weight, y, inv_std, j_ion, loglik);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}
