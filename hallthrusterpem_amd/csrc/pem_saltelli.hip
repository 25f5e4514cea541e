// pem_saltelli.hip -- the Saltelli design of BASELINE configs[4] as ONE launch per shard (gfx950).
//
// What it stands in for: scripts/pem_v0/sobol.py:46-118 -- sample the A and B matrices, run the model on A, B and on A
// with column d from B for every varied input d, and form the sums inside `uq.sobol_sa(..., compute_s2=False)` (uqtils,
// third-party and absent: parity UNPINNED; the estimators are stated in hallthrusterpem_amd/drivers.py).
//
// The block-by-block driver (drivers.sobol_indices, fused=False) launches d + 2 coupled evaluations per batch, each
// regenerating all 15 inputs (8 Philox blocks per evaluation), writes QoIs to HBM and reduces them with a second kernel.
// Here, per base sample, the rows A and B of the counter-based design are generated once (16 Philox blocks; the numbers of
// pem_sample_f64_dev), the model runs in a rolled loop around ONE model body -- one sample per lane -- and the six
// estimator terms of an evaluation are summed over the wave at once (a transposing butterfly: 10 additions instead of
// 48) and added to per-wave fp64 accumulators in LDS; only [workgroups][2 + 2 d][3] partial sums reach HBM.
// Two models share the kernel:
//   Model32  csrc/pem_model_f32.h: single-precision arithmetic on the design rounded to float (pem_saltelli_f32_dev)
//   Model64  the fp64 model, lane per sample: the scalar stages and tables of csrc/pem_model.h with the expression of the
//            reduced-QoI table path of plume_r1_kernel (pem_kernels.hip) -- bit-identical to it for "plain" samples --
//            and the literal 91-term sums for the others (pem_saltelli_f64_dev)
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"
#include "pem_model.h"
#include "pem_model_f32.h"
#include "pem_philox.h"

namespace {

constexpr int NIN = 15;   // P_b V_a T_e V_vac Pstar P_T mdot_a a_1 c0..c5 sigma_cex
constexpr int NQ = 3;     // V_cc, div_angle, T_c

struct SaltelliArg {
    unsigned long long seed, first;
    unsigned int stream;
    int nv;                  // varied inputs
    int varied[NIN];         // their indices (0..14)
    int kind[NIN];
    double a[NIN], b[NIN];
};

struct Eval {                // what the estimator needs from one evaluation
    double f[NQ];
    bool bad_thruster, invalid;
};

// ---- the two models -------------------------------------------------------------------------------------------------
struct Model32 {
    using real = float;
    static constexpr int LDS_BYTES = pem_model32::LDS_FLOATS * 4;
    pem_model32::Tab32 t;
    float k, rad, inv_r2, inv_2pi_r2;
    __device__ void stage(void* lds, int tid, int nthreads, double torr2pa, double radius) {
        t = pem_model32::stage_tables(static_cast<float*>(lds), tid, nthreads);
        k = (float)torr2pa;
        rad = (float)radius;
        inv_r2 = 1.0f / (rad * rad);
        inv_2pi_r2 = 1.0f / (2.0f * pem_model32::F_PI * (rad * rad));
    }
    __device__ __forceinline__ Eval eval(const float (&x)[NIN]) const {
        const pem_model32::Qoi32 o = pem_model32::coupled_f32(x, k, rad, inv_r2, inv_2pi_r2, t);
        return Eval{{(double)o.V_cc, (double)o.div, (double)o.T_c}, o.T < 0.0f || o.I_B0 < 0.0f, o.invalid};
    }
};

// the reference's literal 91-term sums (plume.py:99-123) for a sample outside the tabulated range.  Out of line: never taken
// under the PEM-v0 priors, and inlined its two exp() bodies cost the rolled evaluation loop registers it does not have
struct LiteralSums {
    double den, num, lo;
};
__device__ __attribute__((noinline)) LiteralSums literal_sums(const double2* simpson, double X1a, double X2a, double j_cex, double a1, double a2) {
#pragma clang fp contract(off)
    using namespace pem_model;
    double dd = 0.0, nn = 0.0, lo = __builtin_inf();
    for (int kk = 0; kk < NANG; ++kk) {
        const double alpha = kk == NANG - 1 ? HALF_PI : (double)kk * GRID_H;
        const double t1 = alpha / a1, t2 = alpha / a2;
        const double f = X1a * exp(-(t1 * t1)) + X2a * exp(-(t2 * t2));
        lo = fmin(lo, f + j_cex);
        dd = __builtin_fma(simpson[kk].x, f, dd);
        nn = __builtin_fma(simpson[kk].y, f, nn);
    }
    return LiteralSums{dd, nn, lo};
}

struct Model64 {
    using real = double;
    static constexpr int QPOLY_DOUBLES = (PEM_NDI + PEM_NQB) * PEM_NDC * 2;
    static constexpr int LDS_BYTES = (PEM_NDI * PEM_NDC + QPOLY_DOUBLES + 2 * 96) * 8;
    const double* dpoly;
    const double2* qpoly;
    const double2* simpson;
    double k, rad, inv_r2, inv_2pi_r2;
    __device__ void stage(void* lds, int tid, int nthreads, double torr2pa, double radius) {
        double* d = static_cast<double*>(lds);
        double* q = d + PEM_NDI * PEM_NDC;
        double* s = q + QPOLY_DOUBLES;
        for (int i = tid; i < PEM_NDI * PEM_NDC; i += nthreads) d[i] = PEM_DPOLY[i];
        for (int i = tid; i < QPOLY_DOUBLES; i += nthreads) q[i] = PEM_QPOLY[i];
        for (int i = tid; i < 96; i += nthreads) {
            s[2 * i] = i < PEM_NANGLE ? PEM_SIMPSON_CDEN[i] : 0.0;
            s[2 * i + 1] = i < PEM_NANGLE ? PEM_SIMPSON_CNUM[i] : 0.0;
        }
        dpoly = d;
        qpoly = reinterpret_cast<const double2*>(q);
        simpson = reinterpret_cast<const double2*>(s);
        k = torr2pa;
        rad = radius;
        inv_r2 = 1.0 / (rad * rad);
        inv_2pi_r2 = 1.0 / (2.0 * pem_model::PEM_PI * (rad * rad));
    }
    // cathode.py:24-38 -> tests/sim_hallthruster.jl:35-48 -> plume.py:39-140, reduced QoIs.  The plain branch is, operation
    // for operation, process_tile<..., JMODE 0> of pem_kernels.hip; the other branch is the reference's literal sums.
    __device__ __forceinline__ Eval eval(const double (&x)[NIN]) const {
        using namespace pem_model;
        const double V_cc = cathode_vcc(x[0], x[1], x[2], x[3], x[4], x[5], k);
        const ThrusterQoI th = thruster_stage(x[1], V_cc, x[6], x[7]);
        const PlumeSetup ps = plume_setup(x[0], x[9], x[10], x[11], x[12], x[13], k);
        const double a1 = ps.a1, a2 = ps.a2;
        const double u1 = 1.0 / (a1 * a1), u2 = 1.0 / (a2 * a2);
        const double A1 = (1.0 - x[8]) / normaliser(a1, u1, dpoly);
        const double A2 = x[8] / normaliser(a2, u2, dpoly);
        const double decay = exp(-rad * ps.n_neutral * x[14]);
        const double j_cex = th.I_B0 * (1.0 - decay) * inv_2pi_r2;
        const double base = th.I_B0 * decay * inv_r2;
        const double X1a = base * A1, X2a = base * A2;
        const bool plain = fabs(a1) >= PEM_QA_MIN && fabs(a2) >= PEM_QA_MIN && X1a >= 0.0 && X2a >= 0.0 && j_cex > 0.0 &&
                           (fmax(X1a, X2a) >= 1e-280 || (X1a == 0.0 && X2a == 0.0));
        double d1, n1, d2, n2;
        simpson_functionals(qpoly, fabs(a1), u1, d1, n1);
        simpson_functionals(qpoly, fabs(a2), u2, d2, n2);
        double den = fma(base * A1, d1, (base * A2) * d2), num = fma(base * A1, n1, (base * A2) * n2);
        bool invalid = a1 <= 0.0;
        if (!plain) {
            const LiteralSums ls = literal_sums(simpson, X1a, X2a, j_cex, a1, a2);
            den = ls.den;
            num = ls.num;
            invalid = invalid || ls.lo <= 0.0;
        }
        double cos_div = num / den;
        if (cos_div == __builtin_inf()) cos_div = __builtin_nan("");
        return Eval{{V_cc, acos(cos_div), th.T * cos_div}, th.T < 0.0 || th.I_B0 < 0.0, invalid};
    }
};

// ---- the design -----------------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) double transform_call_s(int kind, double a, double b, double u) {
    return pem::transform(kind, a, b, u);
}

// one row of the design (stream `st`) for global base sample g: bit-identical to pem_sample_f64_dev (then rounded to the
// model's real type)
template <class real>
__device__ __forceinline__ void design_row(const SaltelliArg& s, const int* lds_kind, const double* lds_ab, unsigned long long g,
                                           unsigned int st, real (&x)[NIN]) {
    const unsigned int k0 = (unsigned int)s.seed, k1 = (unsigned int)(s.seed >> 32);
    double u[16];
#pragma unroll
    for (int pair = 0; pair < 8; ++pair) {
        const pem::Philox4 r = pem::philox4x32_10((unsigned int)g, (unsigned int)(g >> 32), (unsigned int)pair, st, k0, k1);
        u[2 * pair] = pem::u53(r.x, r.y);
        u[2 * pair + 1] = pem::u53(r.z, r.w);
    }
#pragma unroll
    for (int d = 0; d < NIN; ++d) {
        const int kd = __builtin_amdgcn_readfirstlane(lds_kind[d]);      // wave-uniform: a scalar branch inside
        x[d] = (real)transform_call_s(kd, lds_ab[2 * d], lds_ab[2 * d + 1], u[d]);
    }
}

// Sum eight per-lane values over the 64 lanes of a wave, TRANSPOSING on the way: after the three halving steps each
// lane carries one of the eight sums, so the whole reduction costs 4 + 2 + 1 + 3 = 10 additions (and shuffles) instead
// of 8 x 6.  On return lane l holds the wave total of v[4 (l & 1) + 2 ((l >> 1) & 1) + ((l >> 2) & 1)].
__device__ __forceinline__ double wave_sum8(const double (&v)[8], int lane) {
    double w4[4], w2[2], w;
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double send = b0 ? v[k] : v[k + 4], keep = b0 ? v[k + 4] : v[k];
        w4[k] = keep + __shfl_xor(send, 1);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double send = b1 ? w4[k] : w4[k + 2], keep = b1 ? w4[k + 2] : w4[k];
        w2[k] = keep + __shfl_xor(send, 2);
    }
    {
        const double send = b2 ? w2[0] : w2[1], keep = b2 ? w2[1] : w2[0];
        w = keep + __shfl_xor(send, 4);
    }
    w += __shfl_xor(w, 8);
    w += __shfl_xor(w, 16);
    w += __shfl_xor(w, 32);
    return w;
}
// which of the eight values lane l ends up with
__device__ __forceinline__ int wave_sum8_slot(int lane) { return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1); }

// partial: [gridDim.x][2 + 2 nv][NQ]: rows 0,1 = sum fA + fB, sum fA^2 + fB^2; rows 2+2j, 3+2j = sum fB (fAB_j - fA),
// sum (fA - fAB_j)^2 for varied input j.  flags: [gridDim.x][2] = non-physical thruster results (T < 0 or I_B0 < 0,
// thruster.py:490-493) and invalid plume samples among all evaluations.
// One model body, a rolled loop over the nv + 2 evaluations of a base sample (A, B, then A with one column of B); a lane
// carries two design rows and one evaluation, not 2 (nv + 1) x 3 running sums.
template <class Model>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))   // two waves per SIMD: <= 256 registers (the fp64 model takes 277 unconstrained)
void saltelli_kernel(long long n, SaltelliArg s, double torr2pa, double radius,
                                                       double* __restrict__ partial, uint64_t* __restrict__ flags) {
    using real = typename Model::real;
    __shared__ __attribute__((aligned(16))) unsigned char lds_model[Model::LDS_BYTES];
    __shared__ int lds_kind[NIN + 1], lds_varied[NIN + 1];
    __shared__ double lds_ab[2 * NIN];
    __shared__ double acc[4][NIN + 1][8];           // [wave][evaluation slot: 0 = the A/B statistics, 1 + j = varied input j][value]
    // row B of the design lives in LDS, one column per input (lane-consecutive: conflict-free): an evaluation needs one value of
    // it (all fifteen only for the B evaluation itself), and held in registers beside row A and the model's own temporaries it
    // pushed the fp64 instantiation past 256 VGPRs (20 spilled, 96 bytes of scratch per lane)
    __shared__ real xb_lds[NIN][256];
    __shared__ unsigned int bad[4][2];
    Model model;
    model.stage(lds_model, threadIdx.x, 256, torr2pa, radius);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < NIN) {
        lds_kind[threadIdx.x] = s.kind[threadIdx.x];
        lds_varied[threadIdx.x] = s.varied[threadIdx.x];
        lds_ab[2 * threadIdx.x] = s.a[threadIdx.x];
        lds_ab[2 * threadIdx.x + 1] = s.b[threadIdx.x];
    }
    for (int i = threadIdx.x; i < 4 * (NIN + 1) * 8; i += 256) (&acc[0][0][0])[i] = 0.0;
    __syncthreads();
    const int nv = s.nv;
    const int my_slot = wave_sum8_slot(lane);
    unsigned int bad_thruster = 0, bad_plume = 0;
    // every wave runs the same number of iterations (the reductions inside need all 64 lanes): lanes past n evaluate
    // the last sample and contribute zeros
    const long long stride = (long long)gridDim.x * 256;
    const long long iters = (n + stride - 1) / stride;
    for (long long it = 0; it < iters; ++it) {
        const long long i = it * stride + (long long)blockIdx.x * 256 + threadIdx.x;
        const bool live = i < n;
        const unsigned long long g = s.first + (unsigned long long)(live ? i : n - 1);
        real xa[NIN];
        {
            real xb[NIN];
            design_row(s, lds_kind, lds_ab, g, s.stream + 1u, xb);
#pragma unroll
            for (int c = 0; c < NIN; ++c) xb_lds[c][threadIdx.x] = xb[c];     // read back by this lane only: no barrier
            // (a compiler barrier: without it the stores are forwarded to the loads below and row B is back in registers)
            asm volatile("" ::: "memory");
        }
        design_row(s, lds_kind, lds_ab, g, s.stream, xa);
        double fa[NQ] = {0.0, 0.0, 0.0}, fb[NQ] = {0.0, 0.0, 0.0};
        for (int e = 0; e < nv + 2; ++e) {                    // 0: A, 1: B, 2 + j: A with column varied[j] from B
            const int d = e >= 2 ? __builtin_amdgcn_readfirstlane(lds_varied[e - 2]) : -1;
            real x[NIN];
#pragma unroll
            for (int c = 0; c < NIN; ++c) x[c] = (e == 1 || c == d) ? xb_lds[c][threadIdx.x] : xa[c];
            const Eval o = model.eval(x);
            if (live) {
                bad_thruster += o.bad_thruster;
                bad_plume += o.invalid;
            }
            if (e == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) fa[q] = o.f[q];
                continue;
            }
            double v[8];
            if (e == 1) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    fb[q] = o.f[q];
                    const double a = fa[q], b = o.f[q];
                    v[q] = a + b;
                    v[NQ + q] = fma(a, a, b * b);
                }
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const double a = fa[q], b = fb[q], ab = o.f[q];
                    v[q] = b * (ab - a);
                    v[NQ + q] = (a - ab) * (a - ab);
                }
            }
            v[6] = v[7] = 0.0;
            if (!live) {
#pragma unroll
                for (int q = 0; q < 6; ++q) v[q] = 0.0;
            }
            const double tot = wave_sum8(v, lane);
            if (lane < 8) acc[wave][e - 1][my_slot] += tot;   // lanes 0..7 carry the eight sums, one each
        }
    }
    {
        double v[8] = {(double)bad_thruster, (double)bad_plume, 0, 0, 0, 0, 0, 0};
        const double tot = wave_sum8(v, lane);
        if (lane < 8 && my_slot < 2) bad[wave][my_slot] = (unsigned int)tot;
    }
    __syncthreads();
    // one partial per workgroup, in a fixed order (deterministic): row 2 j' + {0, 1} x NQ + q  <-  acc[.][j'][{0, 1} NQ + q]
    const int rows = (2 + 2 * nv) * NQ;
    if ((int)threadIdx.x < rows) {
        const int row = threadIdx.x / NQ, q = threadIdx.x - row * NQ, e = row >> 1, which = row & 1;
        partial[(size_t)blockIdx.x * rows + threadIdx.x] =
            acc[0][e][which * NQ + q] + acc[1][e][which * NQ + q] + acc[2][e][which * NQ + q] + acc[3][e][which * NQ + q];
    }
    if (threadIdx.x < 2)
        flags[(size_t)blockIdx.x * 2 + threadIdx.x] = (uint64_t)bad[0][threadIdx.x] + bad[1][threadIdx.x] + bad[2][threadIdx.x] + bad[3][threadIdx.x];
}

template <class Model>
int launch(const char* who, size_t n_base, uint64_t first_index, uint64_t seed, uint32_t stream_id, const int32_t* kind,
           const double* a, const double* b, int n_varied, const int32_t* varied, double torr2pa, double radius, double* partial,
           uint64_t* flags, int n_blocks, pem_stream_t stream) {
    if (!kind || !a || !b || !varied || !partial || !flags) return pem::fail(PEM_ERR_INVALID_ARG, "%s: NULL array", who);
    if (n_varied < 1 || n_varied > NIN) return pem::fail(PEM_ERR_INVALID_ARG, "%s: 1 <= n_varied <= %d", who, NIN);
    if (n_blocks < 1) return pem::fail(PEM_ERR_INVALID_ARG, "%s: n_blocks must be positive", who);
    if (int rc = pem::check_device()) return rc;
    if (n_base == 0) {   // an empty shard contributes zero to every sum: the caller adds the partials up whatever n_base was
        HIP_TRY(hipMemsetAsync(partial, 0, (size_t)n_blocks * (2 + 2 * n_varied) * NQ * sizeof(double), static_cast<hipStream_t>(stream)));
        HIP_TRY(hipMemsetAsync(flags, 0, (size_t)n_blocks * 2 * sizeof(uint64_t), static_cast<hipStream_t>(stream)));
        return PEM_OK;
    }
    SaltelliArg s{};
    s.seed = seed;
    s.first = first_index;
    s.stream = stream_id;
    s.nv = n_varied;
    for (int d = 0; d < NIN; ++d) {
        if (kind[d] < PEM_DIST_UNIFORM || kind[d] > PEM_DIST_NORMAL)
            return pem::fail(PEM_ERR_INVALID_ARG, "%s: unknown distribution kind %d for input %d", who, kind[d], d);
        s.kind[d] = kind[d];
        s.a[d] = a[d];
        s.b[d] = b[d];
        s.varied[d] = 0;
    }
    for (int j = 0; j < n_varied; ++j) {
        if (varied[j] < 0 || varied[j] >= NIN) return pem::fail(PEM_ERR_INVALID_ARG, "%s: varied[%d] = %d out of range", who, j, varied[j]);
        s.varied[j] = varied[j];
    }
    hipLaunchKernelGGL(saltelli_kernel<Model>, dim3((unsigned)n_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), (long long)n_base, s,
                       torr2pa, radius, partial, flags);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // namespace

extern "C" {

int pem_saltelli_f32_dev(size_t n_base, uint64_t first_index, uint64_t seed, uint32_t stream_id, const int32_t* kind,
                         const double* a, const double* b, int n_varied, const int32_t* varied, float torr2pa, float radius,
                         double* partial, uint64_t* flags, int n_blocks, pem_stream_t stream) {
    return launch<Model32>("pem_saltelli_f32", n_base, first_index, seed, stream_id, kind, a, b, n_varied, varied, torr2pa, radius,
                           partial, flags, n_blocks, stream);
}

int pem_saltelli_f64_dev(size_t n_base, uint64_t first_index, uint64_t seed, uint32_t stream_id, const int32_t* kind,
                         const double* a, const double* b, int n_varied, const int32_t* varied, double torr2pa, double radius,
                         double* partial, uint64_t* flags, int n_blocks, pem_stream_t stream) {
    return launch<Model64>("pem_saltelli_f64", n_base, first_index, seed, stream_id, kind, a, b, n_varied, varied, torr2pa, radius,
                           partial, flags, n_blocks, stream);
}

}  // extern "C"
