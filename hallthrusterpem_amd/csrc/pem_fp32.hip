// pem_fp32.hip -- the coupled PEM-v0 evaluation in SINGLE-PRECISION arithmetic, reduced QoIs only (V_cc, div_angle, T_c),
// and the Saltelli design of BASELINE configs[4] fused around it (gfx950).
//
// Why it exists (SURVEY.md section 8d config 5, section 8b "fp32/mixed entry points optional with a tolerance report"):
// the reduced-QoI fp64 kernel is bound by VALU issue, not by HBM, and fp64 VALU instructions issue
// at half the fp32 rate on this chip (tools/microbench/mfma_f64_rate.hip).  A Sobol' analysis needs its QoIs to ~1e-4;
// fp32 gives ~1e-6 (measured per QoI against the fp64 kernel on identical inputs: hallthrusterpem_amd/fp32.py,
// tests/test_fp32.py, bench.py --fp32).  Which steps could have needed fp64 and do not:
//   * 1 - exp(-r n sigma) and the beam amplitude I_B0 exp(-r n sigma): they scale numerator and denominator of
//     cos_div alike and cancel -- they decide only whether a sample is "plain" (amplitudes >= 0, j_cex > 0);
//   * arccos(cos_div): d(angle) = d(cos) / sin(angle); with cos_div good to ~3e-7 the angle is good to 1e-6 .. 1e-5
//     relative over the PEM-v0 priors (angles 0.1 .. 1 rad).  Kept in fp32; the report states the measured tail.
// What is computed is the reference's formulas (cathode.py:24-38, tests/sim_hallthruster.jl:35-48, plume.py:39-140) with
// the same table method as the fp64 kernel (pem_kernels.hip): D(a) and the two Simpson functionals Qd(a), Qn(a) from
// degree-6 polynomial tables in LDS (tools/gen_tables_f32.py), one lane per sample.  A sample that is not "plain" (NaN
// or negative amplitudes, beams narrower than 0.03 rad) takes the literal 91-term sums in fp32 instead.
//
// The model itself is csrc/pem_model_f32.h; this file holds the explicit-input kernel.  The Saltelli design fused around
// the model (pem_saltelli_f32_dev) is csrc/pem_saltelli.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"
#include "pem_model_f32.h"

namespace {

using namespace pem_model32;

// ---------------------------------------------------------------------------------------------------------------
// explicit inputs: x [15][ld] floats -> qoi [3][ldq] floats (V_cc, div_angle, T_c) (+ invalid)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void coupled_f32_kernel(long long n, float k, float rad, const float* __restrict__ x, size_t ld,
                                                          float* __restrict__ qoi, size_t ldq, uint8_t* __restrict__ invalid) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const Tab32 t = stage_tables(lds, threadIdx.x, 256);
    __syncthreads();
    const float inv_r2 = 1.0f / (rad * rad), inv_2pi_r2 = 1.0f / (2.0f * F_PI * (rad * rad));
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float in[NIN];
#pragma unroll
        for (int d = 0; d < NIN; ++d) in[d] = x[(size_t)d * ld + i];
        const Qoi32 o = coupled_f32(in, k, rad, inv_r2, inv_2pi_r2, t);
        qoi[i] = o.V_cc;
        qoi[ldq + i] = o.div;
        qoi[2 * ldq + i] = o.T_c;
        if (invalid) invalid[i] = (uint8_t)o.invalid;
    }
}

}  // namespace

extern "C" {

int pem_coupled_f32_dev(size_t n, float torr2pa, float radius, const float* x, size_t ld, float* qoi, size_t ldq,
                        uint8_t* invalid, pem_stream_t stream) {
    if (n == 0) return PEM_OK;
    if (!x || !qoi) return pem::fail(PEM_ERR_INVALID_ARG, "pem_coupled_f32: NULL array");
    if (ld < n || ldq < n) return pem::fail(PEM_ERR_INVALID_ARG, "pem_coupled_f32: leading dimension smaller than n");
    if (int rc = pem::check_device()) return rc;
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(coupled_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), (long long)n,
                       torr2pa, radius, x, ld, qoi, ldq, invalid);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // extern "C"
