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
// Fused Saltelli kernel (pem_saltelli_f32_dev): for base sample i the rows A_i and B_i of the counter-based design are
// generated once (the SAME numbers as the fp64 design of pem_sample_f64_dev, rounded to float), the model is evaluated
// on A_i, B_i and A_i with column d from B_i for every varied input d, and the sums behind the first-order and total
// Sobol' estimators (hallthrusterpem_amd/drivers.py) are accumulated in fp64 registers: no input or QoI ever touches
// HBM, 16 Philox blocks per base sample instead of 8 per evaluation.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"
#include "pem_philox.h"

#define PEM_TABLE_DECL static __device__ const
#include "pem_tables_f32.h"

namespace {

constexpr int NANG = PEM_NANGLE;
constexpr int NIN = 15;                       // P_b V_a T_e V_vac Pstar P_T mdot_a a_1 c0..c5 sigma_cex
constexpr int NQ = 3;                         // V_cc, div_angle, T_c
constexpr float F_PI = 3.14159265358979323846f;
constexpr float F_HALF_PI = 1.57079632679489661923f;
constexpr float F_GRID_H = 1.57079632679489661923f / 90.0f;
constexpr float F_ALPHA_OVERFLOW = 53.28349511409265f;   // |a| beyond which the reference's erfi bracket is NaN
constexpr int ROW = 8;                        // D rows padded to 8 floats, Q rows to 8 float2: 16-byte LDS reads
constexpr int NQROWS = PEM32_NDI + PEM32_NQB;
constexpr int LDS_FLOATS = PEM32_NDI * ROW + NQROWS * ROW * 2 + 2 * 96;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Tab32 {
    const float* dpoly;     // [32][8]
    const f32x2* qpoly;     // [96][8] {Qd, Qn}
    const f32x2* simpson;   // [96] {cden, cnum}, zero past angle 90
};

__device__ __forceinline__ Tab32 stage_tables(float* lds, int tid, int nthreads) {
    float* d = lds;
    float* q = lds + PEM32_NDI * ROW;
    float* s = q + NQROWS * ROW * 2;
    for (int i = tid; i < PEM32_NDI * ROW; i += nthreads) {
        const int r = i / ROW, j = i - r * ROW;
        d[i] = j < PEM32_NDC ? PEM32_DPOLY[r * PEM32_NDC + j] : 0.0f;
    }
    for (int i = tid; i < NQROWS * ROW * 2; i += nthreads) {
        const int r = i / (ROW * 2), j = (i - r * ROW * 2) >> 1, w = i & 1;
        q[i] = j < PEM32_NDC ? PEM32_QPOLY[(r * PEM32_NDC + j) * 2 + w] : 0.0f;
    }
    for (int i = tid; i < 2 * 96; i += nthreads) s[i] = i < 2 * NANG ? PEM32_SIMPSON[i] : 0.0f;
    return Tab32{d, reinterpret_cast<const f32x2*>(q), reinterpret_cast<const f32x2*>(s)};
}

__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

// D(a), u = 1/a^2 (see pem_kernels.hip::normaliser): table for |a| >= 0.25, series below, NaN where the reference is NaN
__device__ __forceinline__ float normaliser32(float a, float u, const float* dpoly) {
#pragma clang fp contract(off)
    int i = (int)(2.0f * u);
    i = i < 0 ? 0 : (i > PEM32_NDI - 1 ? PEM32_NDI - 1 : i);
    const float x = fmaf(4.0f, u, -(float)(2 * i + 1));
    const f32x4* c = reinterpret_cast<const f32x4*>(dpoly + i * ROW);
    const f32x4 c0 = c[0], c1 = c[1];
    float d = c1.z;                              // degree 6: c[6]
    d = fmaf(d, x, c1.y);
    d = fmaf(d, x, c1.x);
    d = fmaf(d, x, c0.w);
    d = fmaf(d, x, c0.z);
    d = fmaf(d, x, c0.y);
    d = fmaf(d, x, c0.x);
    const float a2 = a * a, y = 0.5f * a2;
    float s = PEM32_DAWSON[PEM32_NDAW - 1];
#pragma unroll
    for (int j = PEM32_NDAW - 2; j >= 0; --j) s = fmaf(s, y, PEM32_DAWSON[j]);
    float D = (fabsf(a) < 0.25f) ? F_PI * a2 * s : d;
    if (!(fabsf(a) <= F_ALPHA_OVERFLOW) || a == 0.0f) D = __builtin_nanf("");
    return D;
}

// {Qd(a), Qn(a)}: the two divergence integrals of one beam (pem_kernels.hip::simpson_functionals), |a| >= QA_MIN
__device__ __forceinline__ f32x2 functionals32(const f32x2* qpoly, float aa, float u) {
#pragma clang fp contract(off)
    const bool wide = aa >= 0.25f;
    const float t = wide ? 2.0f * u : (aa - PEM32_QA_MIN) * PEM32_QB_SCALE;
    const int last = wide ? PEM32_NDI - 1 : PEM32_NQB - 1;
    int i = (int)t;
    i = i > last ? last : (i < 0 ? 0 : i);
    const float x = 2.0f * (t - (float)i) - 1.0f;
    const f32x4* p = reinterpret_cast<const f32x4*>(qpoly + ((wide ? 0 : PEM32_NDI) + i) * ROW);
    const f32x4 p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];    // {d0 n0 d1 n1} {d2 n2 d3 n3} {d4 n4 d5 n5} {d6 n6 - -}
    const f32x2 xx = {x, x};
    f32x2 acc = {p3.x, p3.y};
    acc = __builtin_elementwise_fma(acc, xx, f32x2{p2.z, p2.w});
    acc = __builtin_elementwise_fma(acc, xx, f32x2{p2.x, p2.y});
    acc = __builtin_elementwise_fma(acc, xx, f32x2{p1.z, p1.w});
    acc = __builtin_elementwise_fma(acc, xx, f32x2{p1.x, p1.y});
    acc = __builtin_elementwise_fma(acc, xx, f32x2{p0.z, p0.w});
    acc = __builtin_elementwise_fma(acc, xx, f32x2{p0.x, p0.y});
    return acc;
}

struct Qoi32 {
    float V_cc, div, T_c, I_B0, T;
    bool invalid;
};

// One sample.  x: the 15 coupled inputs in the order of COUPLED_INPUTS.  Every fused multiply-add is written out and
// the compiler is kept from forming others: the explicit-input kernel and the fused Saltelli kernel then evaluate the
// same operations and give the same bits (tests/test_fp32.py holds the fused launch to the block-by-block pipeline).
__device__ __forceinline__ Qoi32 coupled_f32(const float (&x)[NIN], float k, float rad, float inv_r2, float inv_2pi_r2, const Tab32& t) {
#pragma clang fp contract(off)
    const float P_b = x[0], V_a = x[1], T_e = x[2], V_vac = x[3], Pstar = x[4], P_T = x[5], mdot = x[6], a_1 = x[7];
    const float c0 = x[8], c1 = x[9], c2 = x[10], c3 = x[11], c4 = x[12], c5 = x[13], sigma = x[14];
    Qoi32 o;
    // cathode.py:26-37
    const float PB = P_b * k, PS = Pstar * k, PT = P_T * k;
    float V = fmaf(T_e, __logf(1.0f + PB * frcp(PT)), V_vac);
    V = fmaf(-(T_e * frcp(PT + PS)), PB, V);
    if (V < 0.0f) V = 0.0f;
    if (V > V_a) V = V_a;
    o.V_cc = V;
    // sim_hallthruster.jl:37-41
    constexpr float q_over_m = (float)(1.6e-19 / 2.18e-25);
    o.I_B0 = q_over_m * mdot;
    o.T = mdot * __builtin_amdgcn_sqrtf(2.0f * q_over_m * (V_a - V));
    // plume.py:40-61
    const float n_neutral = fmaf(c4, PB, c5);
    float a1 = fmaf(c2, PB, c3);
    if (a1 > F_HALF_PI) a1 = F_HALF_PI;
    const float a2 = a1 * frcp(c1);
    const float u1 = frcp(a1 * a1), u2 = frcp(a2 * a2);
    const float A1 = (1.0f - c0) * frcp(normaliser32(a1, u1, t.dpoly));
    const float A2 = c0 * frcp(normaliser32(a2, u2, t.dpoly));
    // plume.py:95-100
    const float decay = __expf(-rad * n_neutral * sigma);
    const float j_cex = o.I_B0 * (1.0f - decay) * inv_2pi_r2;
    const float base = o.I_B0 * decay * inv_r2;
    const float X1 = base * A1, X2 = base * A2;
    const float aa1 = fabsf(a1), aa2 = fabsf(a2);
    const bool plain = aa1 >= PEM32_QA_MIN && aa2 >= PEM32_QA_MIN && X1 >= 0.0f && X2 >= 0.0f && j_cex > 0.0f &&
                       (fmaxf(X1, X2) >= 1e-30f || (X1 == 0.0f && X2 == 0.0f));
    float den, num;
    bool invalid = a1 <= 0.0f;
    const f32x2 q1 = functionals32(t.qpoly, aa1, u1), q2 = functionals32(t.qpoly, aa2, u2);
    den = fmaf(X1, q1.x, X2 * q2.x);
    num = fmaf(X1, q1.y, X2 * q2.y);
    if (!plain) {
        // the literal sums of plume.py:99-123 (rare: never under the PEM-v0 priors)
        float d = 0.0f, nn = 0.0f, lo = __builtin_inff();
        for (int kk = 0; kk < NANG; ++kk) {
            const float alpha = kk == NANG - 1 ? F_HALF_PI : (float)kk * F_GRID_H;
            const float t1 = alpha * frcp(a1), t2 = alpha * frcp(a2);
            const float f = fmaf(X1, __expf(-(t1 * t1)), X2 * __expf(-(t2 * t2)));
            lo = fminf(lo, f + j_cex);
            const f32x2 w = t.simpson[kk];
            d = fmaf(w.x, f, d);
            nn = fmaf(w.y, f, nn);
        }
        den = d;
        num = nn;
        invalid = invalid || lo <= 0.0f;
    }
    float cos_div = num * frcp(den);               // den == 0: rcp = inf, so x/0 = +-inf and 0/0 = NaN as in the reference
    if (cos_div == __builtin_inff()) cos_div = __builtin_nanf("");
    o.div = acosf(cos_div);
    o.T_c = o.T * cos_div;
    o.invalid = invalid;
    return o;
}

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

// ---------------------------------------------------------------------------------------------------------------
// fused Saltelli design
// ---------------------------------------------------------------------------------------------------------------
struct SaltelliArg {
    unsigned long long seed, first;
    unsigned int stream;
    int nv;                  // varied inputs
    int varied[NIN];         // their indices (0..14)
    int kind[NIN];
    double a[NIN], b[NIN];
};

__device__ __attribute__((noinline)) double transform_call32(int kind, double a, double b, double u) {
    return pem::transform(kind, a, b, u);
}

// one row of the design (stream `st`) for global base sample g, rounded to float: bit-identical to
// (float) pem_sample_f64_dev(...)
__device__ __forceinline__ void design_row(const SaltelliArg& s, const int* lds_kind, const double* lds_ab, unsigned long long g,
                                           unsigned int st, float (&x)[NIN]) {
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
        x[d] = (float)transform_call32(kd, lds_ab[2 * d], lds_ab[2 * d + 1], u[d]);
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
// One model body, a rolled loop over the nv + 2 evaluations of a base sample (A, B, then A with one column of B); the six
// estimator terms of an evaluation are summed over the wave at once (wave_sum8) and added to the wave's accumulators in
// LDS, so a lane carries two design rows and one evaluation, not 2 (nv + 1) x 3 running sums: ~110 registers.
constexpr int NROWS_MAX = 2 + 2 * NIN;
__global__ __launch_bounds__(256) void saltelli_f32_kernel(long long n, SaltelliArg s, float k, float rad,
                                                           double* __restrict__ partial, uint64_t* __restrict__ flags) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ int lds_kind[NIN + 1], lds_varied[NIN + 1];
    __shared__ double lds_ab[2 * NIN];
    __shared__ double acc[4][NIN + 1][8];           // [wave][evaluation slot: 0 = the A/B statistics, 1 + j = varied input j][value]
    __shared__ unsigned int bad[4][2];
    const Tab32 t = stage_tables(lds, threadIdx.x, 256);
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
    const float inv_r2 = 1.0f / (rad * rad), inv_2pi_r2 = 1.0f / (2.0f * F_PI * (rad * rad));
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
        float xa[NIN], xb[NIN];
        design_row(s, lds_kind, lds_ab, g, s.stream, xa);
        design_row(s, lds_kind, lds_ab, g, s.stream + 1u, xb);
        float fa[NQ] = {0.0f, 0.0f, 0.0f}, fb[NQ] = {0.0f, 0.0f, 0.0f};
        for (int e = 0; e < nv + 2; ++e) {                    // 0: A, 1: B, 2 + j: A with column varied[j] from B
            const int d = e >= 2 ? __builtin_amdgcn_readfirstlane(lds_varied[e - 2]) : -1;
            float x[NIN];
#pragma unroll
            for (int c = 0; c < NIN; ++c) x[c] = (e == 1 || c == d) ? xb[c] : xa[c];
            const Qoi32 o = coupled_f32(x, k, rad, inv_r2, inv_2pi_r2, t);
            const float f[NQ] = {o.V_cc, o.div, o.T_c};
            if (live) {
                bad_thruster += (o.T < 0.0f || o.I_B0 < 0.0f);
                bad_plume += o.invalid;
            }
            if (e == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) fa[q] = f[q];
                continue;
            }
            double v[8];
            if (e == 1) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    fb[q] = f[q];
                    const double a = fa[q], b = f[q];
                    v[q] = a + b;
                    v[NQ + q] = fma(a, a, b * b);
                }
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const double a = fa[q], b = fb[q], ab = f[q];
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

int pem_saltelli_f32_dev(size_t n_base, uint64_t first_index, uint64_t seed, uint32_t stream_id, const int32_t* kind,
                         const double* a, const double* b, int n_varied, const int32_t* varied, float torr2pa, float radius,
                         double* partial, uint64_t* flags, int n_blocks, pem_stream_t stream) {
    if (!kind || !a || !b || !varied || !partial || !flags) return pem::fail(PEM_ERR_INVALID_ARG, "pem_saltelli_f32: NULL array");
    if (n_varied < 1 || n_varied > NIN) return pem::fail(PEM_ERR_INVALID_ARG, "pem_saltelli_f32: 1 <= n_varied <= %d", NIN);
    if (n_blocks < 1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_saltelli_f32: n_blocks must be positive");
    if (n_base == 0) return PEM_OK;
    if (int rc = pem::check_device()) return rc;
    SaltelliArg s{};
    s.seed = seed;
    s.first = first_index;
    s.stream = stream_id;
    s.nv = n_varied;
    for (int d = 0; d < NIN; ++d) {
        if (kind[d] < PEM_DIST_UNIFORM || kind[d] > PEM_DIST_NORMAL)
            return pem::fail(PEM_ERR_INVALID_ARG, "pem_saltelli_f32: unknown distribution kind %d for input %d", kind[d], d);
        s.kind[d] = kind[d];
        s.a[d] = a[d];
        s.b[d] = b[d];
        s.varied[d] = 0;
    }
    for (int j = 0; j < n_varied; ++j) {
        if (varied[j] < 0 || varied[j] >= NIN) return pem::fail(PEM_ERR_INVALID_ARG, "pem_saltelli_f32: varied[%d] = %d out of range", j, varied[j]);
        s.varied[j] = varied[j];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(saltelli_f32_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, (long long)n_base, s, torr2pa, radius, partial, flags);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // extern "C"
