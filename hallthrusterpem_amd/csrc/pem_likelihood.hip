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
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"

namespace {

constexpr int BLOCK = 256;

// 16 lanes per sample: lane q of a group takes measurements q, q+16, ...; DPP-sized xor reduction over the group
__global__ __launch_bounds__(BLOCK) void jion_loglik_kernel(long long n, int n_cond, int n_ang,
                                                            const int32_t* __restrict__ kidx,
                                                            const double* __restrict__ wgt, const double* __restrict__ y,
                                                            const double* __restrict__ inv_std,
                                                            const double* __restrict__ j_ion, double* __restrict__ ll) {
    const int q = threadIdx.x & 15;
    const long long group = ((long long)blockIdx.x * BLOCK + threadIdx.x) >> 4;
    const long long ngroups = ((long long)gridDim.x * BLOCK) >> 4;
    for (long long i = group; i < n; i += ngroups) {
        const int e = (int)(i % n_cond);
        const double* row = j_ion + i * PEM_NANGLE;
        const int base = e * n_ang;
        double acc = 0.0;
        for (int a = q; a < n_ang; a += 16) {
            const int k = kidx[base + a];
            const double w = wgt[base + a];
            const double model = fma(w, row[k + 1] - row[k], row[k]);
            const double z = (y[base + a] - model) * inv_std[base + a];
            acc = fma(-0.5 * z, z, acc);
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        acc += __shfl_xor(acc, 8);
        if (q == 0) ll[i] = acc;
    }
}

}  // namespace

extern "C" int pem_jion_loglik_f64_dev(size_t n, int n_cond, int n_ang, const int32_t* kidx, const double* weight,
                                       const double* y, const double* inv_std, const double* j_ion, double* loglik,
                                       pem_stream_t stream) {
    if (n_cond < 1 || n_ang < 1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_jion_loglik: need at least one condition and one angle");
    if (n == 0) return PEM_OK;
    if (!kidx || !weight || !y || !inv_std || !j_ion || !loglik) return pem::fail(PEM_ERR_INVALID_ARG, "pem_jion_loglik: NULL array");
    if (int rc = pem::check_device()) return rc;
    size_t blocks = (n * 16 + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(jion_loglik_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                       (long long)n, n_cond, n_ang, kidx, weight, y, inv_std, j_ion, loglik);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}
