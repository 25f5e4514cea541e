// pem_model.h -- the per-sample scalar pieces of the fp64 model, shared by the tile kernels (pem_kernels.hip) and the
// lane-per-sample Saltelli kernel (pem_saltelli.hip): cathode stage, thruster test double, plume set-up, the beam
// normaliser D(a) and the tabulated Simpson functionals Qd(a), Qn(a).  Reference lines are cited at each function.
#pragma once
#include <hip/hip_runtime.h>

#include "pem_hip.h"

#ifndef PEM_TABLE_DECL
#define PEM_TABLE_DECL static __device__ const
#endif
#include "pem_tables.h"

namespace pem_model {

constexpr int NANG = PEM_NANGLE;
constexpr double PEM_PI = 3.14159265358979323846264338327950288;
constexpr double HALF_PI = PEM_PI / 2;
// |a| beyond which scipy.special.erfi(a/2) overflows in the reference bracket (plume.py:64-85):
// the reference result is NaN there (found by bisection on the reference; tests/golden plume_edges).
constexpr double ALPHA_OVERFLOW = 53.28349511409265;
constexpr double SERIES_BELOW = 0.25;          // |a| < 0.25  <=>  u = 1/a^2 > 16: series instead of table
constexpr double GRID_H = HALF_PI / 90.0;      // step of np.linspace(0, pi/2, 91), plume.py:53

// ---------------------------------------------------------------------------------------------
// per-sample scalar stages
// ---------------------------------------------------------------------------------------------

// cathode.py:26-37.  numpy rounds every product and sum on its own; V_cc is a difference of
// nearly equal terms when V_vac ~ 0, so the operation order and the absence of FMA are kept.
__device__ __forceinline__ double cathode_vcc(double P_b, double V_a, double T_e, double V_vac, double Pstar,
                                              double P_T, double k) {
#pragma clang fp contract(off)
    const double PB = P_b * k;
    const double PS = Pstar * k;
    const double PT = P_T * k;
    const double lg = log(1.0 + PB / PT);
    double V = V_vac + T_e * lg;
    V = V - (T_e / (PT + PS)) * PB;
    if (V < 0.0) V = 0.0;  // NaN compares false and stays NaN, as V_cc[V_cc < 0] = 0 leaves it
    if (V > V_a) V = V_a;
    return V;
}

struct ThrusterQoI {
    double I_B0, I_d, T, eta_c, eta_m, eta_v, eta_a, v_exh;
};

// sim_hallthruster.jl:35-48 -- q and m_ion are that script's literals.
__device__ __forceinline__ ThrusterQoI thruster_stage(double V_a, double V_cc, double mdot, double a1) {
#pragma clang fp contract(off)
    constexpr double q = 1.6e-19, m_ion = 2.18e-25;
    ThrusterQoI o;
    o.I_B0 = (q / m_ion) * mdot;
    o.eta_c = 1.0 - a1 * 2.0;
    o.I_d = o.I_B0 / o.eta_c;
    o.v_exh = sqrt(2.0 * q * (V_a - V_cc) / m_ion);
    o.T = mdot * o.v_exh;
    o.eta_m = 1.0 - a1 * 5.0;
    o.eta_v = 1.0 - a1 * 2.0;
    o.eta_a = 0.5 * (o.T * o.T) / (mdot * V_a * o.I_d);
    return o;
}

// exp(x) for x <= 0 without the special-case handling of the library exp: 2^n * P(r), P = degree-13
// Taylor polynomial on |r| <= ln2/2 (truncation 1.3e-17), n applied with v_ldexp_f64 so that results
// below the normal range denormalise and then flush to 0 like exp() does.  Arguments below -800 give 0;
// a NaN argument gives 0 as well -- callers carry NaN through the beam amplitudes instead.
__device__ __forceinline__ double exp_nonpos(double x) {
    x = fmax(x, -800.0);
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821614599e-10;  // 1/13!
    p = fma(p, r, 2.0876756987868098979e-09);
    p = fma(p, r, 2.5052108385441718775e-08);
    p = fma(p, r, 2.7557319223985890653e-07);
    p = fma(p, r, 2.7557319223985890653e-06);
    p = fma(p, r, 2.4801587301587301587e-05);
    p = fma(p, r, 1.9841269841269841270e-04);
    p = fma(p, r, 1.3888888888888888889e-03);
    p = fma(p, r, 8.3333333333333333333e-03);
    p = fma(p, r, 4.1666666666666666667e-02);
    p = fma(p, r, 1.6666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// D(a) with u = 1/a^2: polynomial table for |a| >= 0.25, series below, NaN where the reference is NaN.
// `poly` points at the 32 x 12 coefficient table (LDS in the fast kernel, global memory otherwise).
__device__ __forceinline__ double normaliser(double a, double u, const double* poly) {
    int i = (int)(2.0 * u);  // u >= 0; NaN -> 0, +inf saturates
    i = i < 0 ? 0 : (i > PEM_NDI - 1 ? PEM_NDI - 1 : i);
    const double x = fma(4.0, u, -(double)(2 * i + 1));
    const double* c = poly + i * PEM_NDC;
    double d = c[PEM_NDC - 1];
#pragma unroll
    for (int j = PEM_NDC - 2; j >= 0; --j) d = fma(d, x, c[j]);
    const double a2 = a * a, y = 0.5 * a2;
    double s = PEM_DAWSON[PEM_NDAW - 1];
#pragma unroll
    for (int j = PEM_NDAW - 2; j >= 0; --j) s = fma(s, y, PEM_DAWSON[j]);
    double D = (fabs(a) < SERIES_BELOW) ? PEM_PI * a2 * s : d;
    if (!(fabs(a) <= ALPHA_OVERFLOW) || a == 0.0) D = __builtin_nan("");
    return D;
}


// plume.py:40, 56-61: the pressure in Pa, the neutral density and the two beam widths.  numpy rounds c4*P_B and c2*P_B
// before adding c5 / c3; a contracted fma() would not, and when c2*P_B cancels c3 the reference's alpha1 is exactly 0
// (invalid sample, NaN normaliser) where the fma leaves a tiny number of either sign.  Shared by every plume kernel so
// that one sample gets one answer whatever path evaluates it.
struct PlumeSetup {
    double n_neutral, a1, a2;
};
__device__ __forceinline__ PlumeSetup plume_setup(double P_b, double c1, double c2, double c3, double c4, double c5, double k) {
#pragma clang fp contract(off)
    PlumeSetup o;
    const double P_B = P_b * k;
    o.n_neutral = c4 * P_B + c5;
    double a1 = c2 * P_B + c3;
    if (a1 > HALF_PI) a1 = HALF_PI;   // upper clip only (plume.py:60); NaN stays NaN
    o.a1 = a1;
    o.a2 = a1 / c1;
    return o;
}

// The two divergence integrals as functions of one beam width (tools/gen_tables.py, QPOLY): with
// f_k = X1 e_k(a1) + X2 e_k(a2) the Simpson sums of plume.py:117-123 are X1 Qd(a1) + X2 Qd(a2) and X1 Qn(a1) + X2 Qn(a2).
// Region A: |a| >= 0.25, row floor(2u), u = 1/a^2;  region B: QA_MIN <= |a| < 0.25, row NDI + floor(t),
// t = (|a| - QA_MIN) * QB_SCALE;  x = 2 (t - row) - 1 in both.  Worst relative error 3.8e-16 (generator self-check).
__device__ __forceinline__ void simpson_functionals(const double2* qpoly, double aa, double u, double& qd, double& qn) {
    const bool wide = aa >= 0.25;
    const double t = wide ? 2.0 * u : (aa - PEM_QA_MIN) * PEM_QB_SCALE;
    const int last = wide ? PEM_NDI - 1 : PEM_NQB - 1;
    int i = (int)t;                      // NaN -> 0; a sample that is not plain evaluates a row it will not use
    i = i > last ? last : (i < 0 ? 0 : i);
    const double x = 2.0 * (t - (double)i) - 1.0;
    const double2* p = qpoly + ((wide ? 0 : PEM_NDI) + i) * PEM_NDC;
    double2 acc = p[PEM_NDC - 1];
#pragma unroll
    for (int j = PEM_NDC - 2; j >= 0; --j) {
        const double2 c = p[j];
        acc.x = fma(acc.x, x, c.x);
        acc.y = fma(acc.y, x, c.y);
    }
    qd = acc.x;
    qn = acc.y;
}

}  // namespace pem_model
