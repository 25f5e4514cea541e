// pem_math.h -- fp64 log10 / 10^x for the `norm: log10` of field QoIs (pem_v0_SPT-100.yml:207-214), written for
// throughput: with the library log10() the compress kernel is bound by 91 calls per sample (1.7 TB/s), not by HBM.
//
// pem_log10: argument reduction x = 2^e * m, m in [sqrt(1/2), sqrt(2)); log(m) = log(1+f) by the classical
//   s = f/(2+f) series 2s + 2s^3/3 + ... summed with the published fdlibm minimax coefficients (|error| < 2^-58.45);
//   log10(x) = e*log10(2) [hi + lo] + log(m)/ln(10).  Held to <= 2 ulp of numpy's log10 over 5e-324 .. 1.8e308
//   (tests/test_compression.py).  Branch-free: denormals through v_frexp, 0 / negative / inf / NaN by selects.
// pem_exp10: n = rint(y*log2(10)), r = y - n*log10(2) [hi + lo], 10^r = exp(r ln10) by a degree-13 Taylor polynomial
//   (|r ln10| <= ln2/2, truncation 1.3e-17), scaled by 2^n with ldexp; the argument is clamped to [-330, 310] where
//   ldexp already gives 0 / inf.  Branch-free.
//
// pem_log10_tab: the same function from a 1024-entry table staged in LDS (tools/gen_log_table.py, csrc/pem_log_table.h):
//   i = top 10 mantissa bits, r = fma(m, c_i, -1), log10 x = (e - [i < 424]) log10(2) + T_i + r P5(r).  No division, a
//   degree-5 instead of a degree-7 polynomial, the special cases decided on the high dword with v_cmp_class: ~13 fp64 +
//   ~12 fp32-rate instructions against ~40 fp64 ones (fp64 VALU issues at half the fp32 rate on this chip: measured 7.5
//   cycles per wave instruction).  Largest error 1.3 ulp against a 64-bit-mantissa log10 (libm's own: 1.6 ulp).
#pragma once
#include <hip/hip_runtime.h>

#ifndef PEM_TABLE_DECL
#define PEM_TABLE_DECL static __device__ const
#endif
#include "pem_log_table.h"

namespace pem {

typedef double log_f64x2 __attribute__((ext_vector_type(2)));
constexpr int LOG_TABLE_DOUBLES = 2 * PEM_LOG_N;

// copy the table into LDS (called by every thread of the workgroup before a __syncthreads)
__device__ __forceinline__ void load_log_table(double* lds_tab, int tid, int nthreads) {
    for (int i = tid; i < LOG_TABLE_DOUBLES; i += nthreads) lds_tab[i] = PEM_LOG10_TAB[i];
}

__device__ __forceinline__ double pem_log10(double x);
__device__ __forceinline__ double pem_log10_tab(double x, const double* lds_tab) {
#ifdef PEM_LOG10_NO_TABLE                                     // A/B builds: the series version everywhere (tools/build_variant.sh)
    return pem_log10(x);
#endif
    const double m = __builtin_amdgcn_frexp_mant(x);         // [1/2, 1); denormals normalised in hardware
    int e = __builtin_amdgcn_frexp_exp(x);
    const unsigned mh = (unsigned)__double2hiint(m);
    const int i = (int)((mh >> (20 - PEM_LOG_NBITS)) & (PEM_LOG_N - 1));
    e -= i < PEM_LOG_LOW_BELOW ? 1 : 0;
    const log_f64x2 ct = reinterpret_cast<const log_f64x2*>(lds_tab)[i];       // one ds_read_b128
    const double r = fma(m, ct.x, -1.0);
    double p = PEM_LOG_A6;
    p = fma(p, r, PEM_LOG_A5);
    p = fma(p, r, PEM_LOG_A4);
    p = fma(p, r, PEM_LOG_A3);
    p = fma(p, r, PEM_LOG_A2);
    p = fma(p, r, PEM_LOG_A1);
    const double de = (double)e;
    const double res = fma(de, 3.01029995663611771306e-01, ct.y) + fma(de, 3.69423907715893078616e-13, p * r);
    // +-0 -> -inf, +inf -> +inf, negative and NaN -> NaN: the three constants differ in their high dword only
    const bool ok = __builtin_amdgcn_class(x, 0x180);        // +subnormal | +normal
    const unsigned sp = __builtin_amdgcn_class(x, 0x060) ? 0xfff00000u : (__builtin_amdgcn_class(x, 0x200) ? 0x7ff00000u : 0x7ff80000u);
    return __hiloint2double(ok ? __double2hiint(res) : (int)sp, ok ? __double2loint(res) : 0);
}

// pem_log10_tab for a POSITIVE FINITE argument (normal or subnormal): the same value bit for bit, without the special
// cases -- the caller has excluded them (csrc/pem_latent.hip decides them once per sample instead of once per angle).
__device__ __forceinline__ double pem_log10_tab_pos(double x, const double* lds_tab) {
    const double m = __builtin_amdgcn_frexp_mant(x);
    int e = __builtin_amdgcn_frexp_exp(x);
    const unsigned mh = (unsigned)__double2hiint(m);
    const int i = (int)((mh >> (20 - PEM_LOG_NBITS)) & (PEM_LOG_N - 1));
    e -= i < PEM_LOG_LOW_BELOW ? 1 : 0;
    const log_f64x2 ct = reinterpret_cast<const log_f64x2*>(lds_tab)[i];
    const double r = fma(m, ct.x, -1.0);
    double p = PEM_LOG_A6;
    p = fma(p, r, PEM_LOG_A5);
    p = fma(p, r, PEM_LOG_A4);
    p = fma(p, r, PEM_LOG_A3);
    p = fma(p, r, PEM_LOG_A2);
    p = fma(p, r, PEM_LOG_A1);
    const double de = (double)e;
    return fma(de, 3.01029995663611771306e-01, ct.y) + fma(de, 3.69423907715893078616e-13, p * r);
}

__device__ __forceinline__ double pem_log10(double x) {
    // v_frexp_{mant,exp}_f64 normalise denormals in hardware: m in [1/2, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654757;
    m = low ? m + m : m;                 // m in [sqrt(1/2), sqrt(2))
    e = low ? e - 1 : e;
    const double f = m - 1.0;            // exact
#ifdef PEM_LOG10_IEEE_DIV
    const double s = f / (2.0 + f);
#else
    // s = f / (2 + f) by v_rcp_f64 + two Newton steps + one residual correction (8 instructions against ~15 for the
    // IEEE division sequence; s within 1 ulp, which the 2-ulp bound of the result absorbs)
    const double d = 2.0 + f;
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    double s = f * y;
    s = fma(fma(-d, s, f), y, s);
#endif
    const double z = s * s;
    double R = 1.479819860511658591e-01;
    R = fma(R, z, 1.531383769920937332e-01);
    R = fma(R, z, 1.818357216161805012e-01);
    R = fma(R, z, 2.222219843214978396e-01);
    R = fma(R, z, 2.857142874366239149e-01);
    R = fma(R, z, 3.999999999940941908e-01);
    R = fma(R, z, 6.666666666666735130e-01);
    R *= z;
    const double hfsq = 0.5 * f * f;
    const double log_m = f - (hfsq - s * (hfsq + R));   // log(1 + f)
    const double de = (double)e;
    const double r = fma(de, 3.01029995663611771306e-01, fma(de, 3.69423907715893078616e-13, 4.34294481903251816668e-01 * log_m));
    // branch-free specials (a branch per element would stop the compiler interleaving the unrolled evaluations):
    // +0/-0 -> -inf, +inf -> +inf, negative and NaN -> NaN
    const double special = x == 0.0 ? -__builtin_inf() : (x == __builtin_inf() ? x : __builtin_nan(""));
    return (x > 0.0 && x < __builtin_inf()) ? r : special;
}

__device__ __forceinline__ double pem_exp10(double y) {
    const double yc = fmin(fmax(y, -330.0), 310.0);      // beyond: 0 and inf through ldexp; NaN restored at the end
    const double n = rint(yc * 3.32192809488736234787);
    double r = fma(n, -3.01029995663611771306e-01, yc);  // n * hi is exact: |n| < 2^11, hi has 13 trailing zero bits
    r = fma(n, -3.69423907715893078616e-13, r);
    const double t = r * 2.30258509299404568402;         // |t| <= ln2/2 + rounding
    double p = 1.6059043836821613e-10;                   // 1/13!
    p = fma(p, t, 2.08767569878681e-09);
    p = fma(p, t, 2.505210838544172e-08);
    p = fma(p, t, 2.755731922398589e-07);
    p = fma(p, t, 2.7557319223985893e-06);
    p = fma(p, t, 2.48015873015873e-05);
    p = fma(p, t, 1.984126984126984e-04);
    p = fma(p, t, 1.3888888888888889e-03);
    p = fma(p, t, 8.333333333333333e-03);
    p = fma(p, t, 4.1666666666666664e-02);
    p = fma(p, t, 1.6666666666666666e-01);
    p = fma(p, t, 0.5);
    p = fma(p, t, 1.0);
    p = fma(p, t, 1.0);
    const double v = ldexp(p, (int)n);                   // one rounding, also into the denormal range
    return y != y ? y : v;
}

}  // namespace pem
