// pem_model_f32.h -- the coupled PEM-v0 model in single-precision arithmetic, reduced QoIs only (V_cc, div_angle, T_c): one
// sample per lane.  Shared by the explicit-input kernel (pem_fp32.hip) and the fused Saltelli kernel (pem_saltelli.hip);
// pem_fp32.hip's header states what it is for and how accurate it is.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_hip.h"

#ifndef PEM_TABLE_DECL
#define PEM_TABLE_DECL static __device__ const
#endif
#include "pem_tables_f32.h"

namespace pem_model32 {

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

}  // namespace pem_model32
