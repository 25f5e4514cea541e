// pem_kernels.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI of libpem_hip.so.
//
// What is computed (reference file:line, upstream repository root):
//   cathode stage    src/hallmd/models/cathode.py:24-38
//   thruster stage   tests/sim_hallthruster.jl:35-48 (the reference's analytic test double)
//   plume stage      src/hallmd/models/plume.py:39-140
//
// Kernel shape (see DESIGN.md for the measurements behind it).  The path is a streaming map with
// 120 B in and 752 B out per sample, HBM-write-bound once the per-angle transcendentals are
// removed, so the design is about (1) few fp64 transcendentals and (2) full-line coalesced stores.
//   * L lanes share one sample (L = 4 by default); a 64-lane wave owns S = 64/L consecutive
//     samples.  Lane (s, c) computes angles k = c*CH .. c*CH+CH-1, CH = ceil(91/L).
//   * The two Gaussians exp(-(k h / a)^2) are advanced along k by the two-term recurrence
//     e_{k+1} = e_k r_k, r_{k+1} = r_k q  (q = exp(-2 (h/a)^2)), restarted with direct exp() at
//     every chunk start -- 3 exp per beam per lane instead of 91 per beam per sample; the error
//     grows as ~CH^2/2 ulp (3e-14 for CH = 23).
//   * The normaliser D(a) = 2 pi Int_0^{pi/2} exp(-(t/a)^2) sin t dt (identical to the six complex
//     erfi of plume.py:64-85) is a 24-point Gauss-Legendre sum split over the L lanes of a sample
//     and combined with wavefront shuffles; below |a| = 0.25 a 10-term series takes over.
//   * The Simpson sums of plume.py:117-123 are accumulated in the same k loop with folded weights
//     from an LDS table and combined with the same shuffles.
//   * A wave's S x 91 profile block is contiguous in j_ion (R = 1), so it is staged in LDS in its
//     final order and written out with 16-byte-per-lane, 1-KiB-per-instruction stores.
//
// This file is written for gfx950 only: 64-wide waves, 160 KiB LDS, no portability layer.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "pem_hip.h"

#define PEM_TABLE_DECL static __device__ const
#include "pem_tables.h"

namespace {

constexpr int NANG = PEM_NANGLE;
constexpr double PEM_PI = 3.14159265358979323846264338327950288;
constexpr double HALF_PI = PEM_PI / 2;
// |a| beyond which scipy.special.erfi(a/2) overflows in the reference bracket (plume.py:64-85):
// the reference result is NaN there (found by bisection on the reference; tests/golden plume_edges).
constexpr double ALPHA_OVERFLOW = 53.28349511409265;
constexpr double SERIES_BELOW = 0.25;
constexpr int BLOCK = 256;
// waves per workgroup of the fast kernel: the LDS tile of one wave is (64/L)*91*8 bytes
template <int L>
constexpr int waves_per_block() { return L == 1 ? 2 : 4; }
constexpr int TABLE_DOUBLES = 2 * NANG + 2 * PEM_NGL;  // {cden,cnum}[91] then {t2,ws}[24]

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

// D(a) from a finished Gauss-Legendre sum: series below 0.25, NaN where the reference is NaN.
__device__ __forceinline__ double finish_normaliser(double a, double gl_sum) {
    const double a2 = a * a;
    const double y = 0.5 * a2;
    double s = PEM_DAWSON[PEM_NDAW - 1];
#pragma unroll
    for (int i = PEM_NDAW - 2; i >= 0; --i) s = fma(s, y, PEM_DAWSON[i]);
    double D = (fabs(a) < SERIES_BELOW) ? PEM_PI * a2 * s : gl_sum;
    if (!(fabs(a) <= ALPHA_OVERFLOW) || a == 0.0) D = __builtin_nan("");
    return D;
}

template <int MASK_LO>
__device__ __forceinline__ double xor_reduce_add(double v) {
#pragma unroll
    for (int m = MASK_LO; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

template <int MASK_LO>
__device__ __forceinline__ int xor_reduce_or(int v) {
#pragma unroll
    for (int m = MASK_LO; m < 64; m <<= 1) v |= __shfl_xor(v, m);
    return v;
}

// ---------------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------------
struct PlumeIO {
    long long n;
    double torr2pa;
    double radius;  // R = 1 fast path
    const double *P_b, *c0, *c1, *c2, *c3, *c4, *c5, *sigma, *I_B0, *T;
    double *j_ion, *div, *Tc;
    uint8_t* invalid;
};

struct CoupledIO {
    const double *V_a, *T_e, *V_vac, *Pstar, *P_T, *mdot_a, *a_1;
    double *V_cc, *I_B0, *T;
};

// ---------------------------------------------------------------------------------------------
// fast path: R = 1, L lanes per sample, LDS-staged coalesced profile stores
//   COUPLED: cathode + thruster stages are evaluated in front of the plume (inputs from CoupledIO)
//   WRITE_J: stage and store the 91-point profile (false = reduced-QoI mode)
// ---------------------------------------------------------------------------------------------
template <int L, bool COUPLED, bool WRITE_J>
__global__ __launch_bounds__(64 * waves_per_block<L>()) void plume_r1_kernel(PlumeIO io, CoupledIO cio) {
    static_assert(L == 1 || L == 2 || L == 4 || L == 8, "lanes per sample");
    constexpr int WAVES = waves_per_block<L>();
    constexpr int NTHREADS = 64 * WAVES;
    constexpr int S = 64 / L;               // samples per wave
    constexpr int CH = (NANG + L - 1) / L;  // angles per lane
    constexpr int NPL = PEM_NGL / L;        // Gauss-Legendre nodes per lane
    constexpr int TILE = S * NANG;          // doubles per wave tile
    constexpr int UNROLL = CH <= 23 ? CH : 2;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* tab = reinterpret_cast<double*>(smem_raw);
    double2* tab_simpson = reinterpret_cast<double2*>(tab);            // [91] {cden, cnum}
    double2* tab_gl = reinterpret_cast<double2*>(tab + 2 * NANG);      // [24] {t2, ws}
    double* tiles = tab + TABLE_DOUBLES;

    const int tid = threadIdx.x;
    for (int i = tid; i < NANG; i += NTHREADS) tab_simpson[i] = make_double2(PEM_SIMPSON_CDEN[i], PEM_SIMPSON_CNUM[i]);
    if (tid < PEM_NGL) tab_gl[tid] = make_double2(PEM_GL_T2[tid], PEM_GL_WS[tid]);
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int s = lane % S, c = lane / S;
    const long long tile_base = ((long long)blockIdx.x * WAVES + wave) * S;  // first sample of this wave
    const long long g = tile_base + s;
    const bool live = g < io.n;
    const long long gi = live ? g : io.n - 1;  // dead lanes recompute the last sample, write nothing

    // ---- inputs (SoA: S consecutive doubles per array per wave, L lanes share an address) ----
    const double P_b = io.P_b[gi];
    const double c0 = io.c0[gi], c1 = io.c1[gi], c2 = io.c2[gi], c3 = io.c3[gi];
    const double c4 = io.c4[gi], c5 = io.c5[gi], sigma = io.sigma[gi];
    double I_B0, thrust = 0.0, V_cc = 0.0;
    bool have_T;
    if constexpr (COUPLED) {
        const double V_a = cio.V_a[gi];
        V_cc = cathode_vcc(P_b, V_a, cio.T_e[gi], cio.V_vac[gi], cio.Pstar[gi], cio.P_T[gi], io.torr2pa);
        const ThrusterQoI th = thruster_stage(V_a, V_cc, cio.mdot_a[gi], cio.a_1[gi]);
        I_B0 = th.I_B0;
        thrust = th.T;
        have_T = true;
    } else {
        I_B0 = io.I_B0[gi];
        have_T = io.T != nullptr;
        if (have_T) thrust = io.T[gi];
    }

    // ---- plume.py:40-61 ----
    const double P_B = P_b * io.torr2pa;
    const double n_neutral = c4 * P_B + c5;
    double a1 = c2 * P_B + c3;
    if (a1 > HALF_PI) a1 = HALF_PI;
    const double a2 = a1 / c1;
    const double u1 = 1.0 / (a1 * a1), u2 = 1.0 / (a2 * a2);

    // ---- normalisers: this lane's share of the 24 Gauss-Legendre nodes, then shuffle-reduce ----
    double gl1 = 0.0, gl2 = 0.0;
#pragma unroll
    for (int m = 0; m < NPL; ++m) {
        const double2 node = tab_gl[c + L * m];
        gl1 = fma(node.y, exp(-node.x * u1), gl1);
        gl2 = fma(node.y, exp(-node.x * u2), gl2);
    }
    if constexpr (L > 1) {
        gl1 = xor_reduce_add<S>(gl1);
        gl2 = xor_reduce_add<S>(gl2);
    }
    const double A1 = (1.0 - c0) / finish_normaliser(a1, gl1);  // plume.py:64-73
    const double A2 = c0 / finish_normaliser(a2, gl2);          // plume.py:75-85

    // ---- plume.py:95-100 at the single radius ----
    const double rad = io.radius;
    const double decay = exp(-rad * n_neutral * sigma);
    const double j_cex = I_B0 * (1.0 - decay) / (2.0 * PEM_PI * (rad * rad));
    const double base = I_B0 * decay / (rad * rad);
    const double B1 = base * A1, B2 = base * A2;

    // ---- Gaussian recurrences, started exactly at this lane's first angle ----
    constexpr double H = HALF_PI / 90.0;  // grid step of np.linspace(0, pi/2, 91)
    const double s1 = (H * H) * u1, s2 = (H * H) * u2;
    const int k0 = c * CH;
    const double dk0 = (double)k0;
    double e1 = exp(-(dk0 * dk0) * s1), r1 = exp(-(2.0 * dk0 + 1.0) * s1);
    double e2 = exp(-(dk0 * dk0) * s2), r2 = exp(-(2.0 * dk0 + 1.0) * s2);
    const double q1 = exp(-2.0 * s1), q2 = exp(-2.0 * s2);

    double* tile = tiles + wave * TILE;
    double den = 0.0, num = 0.0;
    int invalid = (a1 <= 0.0) ? 1 : 0;  // plume.py:105, first term
#pragma unroll UNROLL
    for (int j = 0; j < CH; ++j) {
        const int k = k0 + j;
        if (L == 1 || k < NANG) {
            const double2 w = tab_simpson[k];
            const double f = B1 * e1 + B2 * e2;  // j_beam + j_scat
            const double ji = f + j_cex;         // plume.py:102
            if constexpr (WRITE_J) tile[s * NANG + k] = ji;
            invalid |= (ji <= 0.0) ? 1 : 0;      // plume.py:105, second term
            den = fma(w.x, f, den);
            num = fma(w.y, f, num);
            e1 *= r1;
            r1 *= q1;
            e2 *= r2;
            r2 *= q2;
        }
    }
    if constexpr (L > 1) {
        den = xor_reduce_add<S>(den);
        num = xor_reduce_add<S>(num);
        invalid = xor_reduce_or<S>(invalid);
    }

    // ---- plume.py:124-140 ----
    double cos_div = num / den;
    if (cos_div == __builtin_inf()) cos_div = __builtin_nan("");
    if (live && c == 0) {
        io.div[g] = acos(cos_div);
        if (have_T) io.Tc[g] = thrust * cos_div;
        if (io.invalid) io.invalid[g] = (uint8_t)invalid;
        if constexpr (COUPLED) {
            cio.V_cc[g] = V_cc;
            if (cio.I_B0) cio.I_B0[g] = I_B0;
            if (cio.T) cio.T[g] = thrust;
        }
    }

    if constexpr (WRITE_J) {
        if (invalid) {  // plume.py:106: the whole profile of an invalid sample becomes 1e-20 (rare)
            for (int j = 0; j < CH; ++j)
                if (k0 + j < NANG) tile[s * NANG + k0 + j] = 1e-20;
        }
        __syncthreads();
        // the wave's S*91 doubles are one contiguous, 16-byte aligned block of j_ion
        long long valid = (io.n - tile_base) * NANG;  // doubles of this tile that exist
        if (valid > TILE) valid = TILE;
        if (valid > 0) {
            double* dst = io.j_ion + tile_base * NANG;
            const double2* src2 = reinterpret_cast<const double2*>(tile);
            double2* dst2 = reinterpret_cast<double2*>(dst);
            const int pairs = (int)(valid >> 1);
#pragma unroll 4
            for (int i = lane; i < pairs; i += 64) dst2[i] = src2[i];
            if ((valid & 1) && lane == 0) dst[valid - 1] = tile[valid - 1];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// general path: any number of radii, one lane per sample, plain strided stores.  Used for
// sweep_radius arrays (tests/test_plume.py:31 uses 25 radii); not the benchmarked configuration.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void plume_generic_kernel(PlumeIO io, const double* __restrict__ radii, int R) {
    const long long g = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= io.n) return;
    const double P_B = io.P_b[g] * io.torr2pa;
    const double c0 = io.c0[g], c1 = io.c1[g];
    const double n_neutral = io.c4[g] * P_B + io.c5[g];
    const double sigma = io.sigma[g], I_B0 = io.I_B0[g];
    double a1 = io.c2[g] * P_B + io.c3[g];
    if (a1 > HALF_PI) a1 = HALF_PI;
    const double a2 = a1 / c1;
    const double u1 = 1.0 / (a1 * a1), u2 = 1.0 / (a2 * a2);
    double gl1 = 0.0, gl2 = 0.0;
    for (int i = 0; i < PEM_NGL; ++i) {
        gl1 = fma(PEM_GL_WS[i], exp(-PEM_GL_T2[i] * u1), gl1);
        gl2 = fma(PEM_GL_WS[i], exp(-PEM_GL_T2[i] * u2), gl2);
    }
    const double A1 = (1.0 - c0) / finish_normaliser(a1, gl1);
    const double A2 = c0 / finish_normaliser(a2, gl2);
    constexpr double H = HALF_PI / 90.0;
    const double s1 = (H * H) * u1, s2 = (H * H) * u2;
    const double r10 = exp(-s1), r20 = exp(-s2), q1 = exp(-2.0 * s1), q2 = exp(-2.0 * s2);
    const bool have_T = io.T != nullptr;
    const double thrust = have_T ? io.T[g] : 0.0;

    int invalid = (a1 <= 0.0) ? 1 : 0;
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: integrals + invalid flag; pass 1: profile stores
        for (int r = 0; r < R; ++r) {
            const double rad = radii[r];
            const double decay = exp(-rad * n_neutral * sigma);
            const double j_cex = I_B0 * (1.0 - decay) / (2.0 * PEM_PI * (rad * rad));
            const double base = I_B0 * decay / (rad * rad);
            const double B1 = base * A1, B2 = base * A2;
            double e1 = (a1 == 0.0) ? __builtin_nan("") : 1.0, e2 = e1, r1 = r10, r2 = r20, den = 0.0, num = 0.0;
            for (int k = 0; k < NANG; ++k) {
                const double f = B1 * e1 + B2 * e2;
                const double ji = f + j_cex;
                if (pass == 0) {
                    invalid |= (ji <= 0.0) ? 1 : 0;
                    den = fma(PEM_SIMPSON_CDEN[k], f, den);
                    num = fma(PEM_SIMPSON_CNUM[k], f, num);
                } else {
                    io.j_ion[((size_t)g * NANG + k) * R + r] = invalid ? 1e-20 : ji;
                }
                e1 *= r1;
                r1 *= q1;
                e2 *= r2;
                r2 *= q2;
            }
            if (pass == 0) {
                double cos_div = num / den;
                if (cos_div == __builtin_inf()) cos_div = __builtin_nan("");
                io.div[(size_t)g * R + r] = acos(cos_div);
                if (have_T) io.Tc[(size_t)g * R + r] = thrust * cos_div;
            }
        }
    }
    if (io.invalid) io.invalid[g] = (uint8_t)invalid;
}

__global__ __launch_bounds__(BLOCK) void cathode_kernel(long long n, const double* __restrict__ P_b,
                                                        const double* __restrict__ V_a, const double* __restrict__ T_e,
                                                        const double* __restrict__ V_vac,
                                                        const double* __restrict__ Pstar,
                                                        const double* __restrict__ P_T, double k,
                                                        double* __restrict__ V_cc) {
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride)
        V_cc[i] = cathode_vcc(P_b[i], V_a[i], T_e[i], V_vac[i], Pstar[i], P_T[i], k);
}

struct ThrusterOut {
    double *I_B0, *I_d, *T, *eta_c, *eta_m, *eta_v, *eta_a, *v_exh;
};

__global__ __launch_bounds__(BLOCK) void thruster_kernel(long long n, const double* __restrict__ V_a,
                                                         const double* __restrict__ V_cc,
                                                         const double* __restrict__ mdot,
                                                         const double* __restrict__ a_1, ThrusterOut o) {
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const ThrusterQoI t = thruster_stage(V_a[i], V_cc[i], mdot[i], a_1[i]);
        if (o.I_B0) o.I_B0[i] = t.I_B0;
        if (o.I_d) o.I_d[i] = t.I_d;
        if (o.T) o.T[i] = t.T;
        if (o.eta_c) o.eta_c[i] = t.eta_c;
        if (o.eta_m) o.eta_m[i] = t.eta_m;
        if (o.eta_v) o.eta_v[i] = t.eta_v;
        if (o.eta_a) o.eta_a[i] = t.eta_a;
        if (o.v_exh) o.v_exh[i] = t.v_exh;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";
int g_lanes = 4;
double g_angle_grid[NANG];
std::once_flag g_grid_once;

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorNoDevice ? PEM_ERR_NO_DEVICE : PEM_ERR_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                                \
    } while (0)

int check_device() {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt == 0) {
        (void)hipGetLastError();
        return fail(PEM_ERR_NO_DEVICE, "no HIP device available (%s); libpem_hip has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    return PEM_OK;
}

template <int L, bool COUPLED, bool WRITE_J>
int launch_r1(const PlumeIO& io, const CoupledIO& cio, hipStream_t st) {
    constexpr int S = 64 / L;
    constexpr int WAVES = waves_per_block<L>();
    const size_t lds = (size_t)TABLE_DOUBLES * 8 + (WRITE_J ? (size_t)WAVES * S * NANG * 8 : 0);
    const long long per_block = (long long)WAVES * S;
    const long long blocks = (io.n + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return fail(PEM_ERR_INVALID_ARG, "n = %lld needs more than 2^31 workgroups", io.n);
    auto kern = plume_r1_kernel<L, COUPLED, WRITE_J>;
    if (lds > 48 * 1024) {  // opt in to more than the default dynamic-LDS limit, once per instantiation
        static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        HIP_TRY(attr);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * WAVES), lds, st, io, cio);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

template <bool COUPLED, bool WRITE_J>
int dispatch_lanes(const PlumeIO& io, const CoupledIO& cio, hipStream_t st) {
    switch (g_lanes) {
        case 1: return launch_r1<1, COUPLED, WRITE_J>(io, cio, st);
        case 2: return launch_r1<2, COUPLED, WRITE_J>(io, cio, st);
        case 8: return launch_r1<8, COUPLED, WRITE_J>(io, cio, st);
        default: return launch_r1<4, COUPLED, WRITE_J>(io, cio, st);
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// device workspace of the host-pointer entry points
struct Workspace {
    std::mutex mu;
    void* buf = nullptr;
    size_t cap = 0;
    int device = -1;
    int reserve(size_t bytes) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (buf && (cap < bytes || dev != device)) {
            (void)hipFree(buf);
            buf = nullptr;
            cap = 0;
        }
        if (!buf) {
            HIP_TRY(hipMalloc(&buf, bytes));
            cap = bytes;
            device = dev;
        }
        return PEM_OK;
    }
} g_ws;

// carve 256-byte aligned arrays out of the workspace
struct Carver {
    unsigned char* base;
    size_t off = 0;
    explicit Carver(void* b) : base(static_cast<unsigned char*>(b)) {}
    template <class T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base + off);
        off += (count * sizeof(T) + 255) & ~size_t(255);
        return p;
    }
};
size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* pem_version(void) { return "hallthrusterpem_amd libpem_hip 0.1.0 (gfx950)"; }

const char* pem_last_error(void) { return g_err; }

int pem_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return cnt;
}

int pem_init(int device) {
    if (int rc = check_device()) return rc;
    HIP_TRY(hipSetDevice(device));
    return PEM_OK;
}

int pem_synchronize(pem_stream_t stream) {
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PEM_OK;
}

int pem_set_lanes_per_sample(int lanes) {
    if (lanes == 0) lanes = 4;
    if (lanes == 1 || lanes == 2 || lanes == 4 || lanes == 8) g_lanes = lanes;
    return g_lanes;
}

const double* pem_angle_grid(void) {
    std::call_once(g_grid_once, [] {
        // np.linspace(0, pi/2, 91): k * ((pi/2) / 90), last point exactly pi/2 (plume.py:53)
        const double step = HALF_PI / 90.0;
        for (int k = 0; k < NANG; ++k) g_angle_grid[k] = (double)k * step;
        g_angle_grid[NANG - 1] = HALF_PI;
    });
    return g_angle_grid;
}

// ---- cathode ---------------------------------------------------------------------------------
int pem_cathode_f64_dev(size_t n, const double* P_b, const double* V_a, const double* T_e, const double* V_vac,
                        const double* Pstar, const double* P_T, double torr2pa, double* V_cc, pem_stream_t stream) {
    if (n == 0) return PEM_OK;
    if (!P_b || !V_a || !T_e || !V_vac || !Pstar || !P_T || !V_cc) return fail(PEM_ERR_INVALID_ARG, "pem_cathode: NULL array");
    if (int rc = check_device()) return rc;
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 8 * 4) blocks = 256 * 8 * 4;  // grid-stride beyond a few waves per SIMD
    hipLaunchKernelGGL(cathode_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                       (long long)n, P_b, V_a, T_e, V_vac, Pstar, P_T, torr2pa, V_cc);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

// ---- thruster test double ----------------------------------------------------------------------
int pem_thruster_f64_dev(size_t n, const double* V_a, const double* V_cc, const double* mdot_a, const double* a_1,
                         double* I_B0, double* I_d, double* T, double* eta_c, double* eta_m, double* eta_v,
                         double* eta_a, double* v_exh, pem_stream_t stream) {
    if (n == 0) return PEM_OK;
    if (!V_a || !V_cc || !mdot_a || !a_1) return fail(PEM_ERR_INVALID_ARG, "pem_thruster: NULL input array");
    if (int rc = check_device()) return rc;
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 8 * 4) blocks = 256 * 8 * 4;
    ThrusterOut o{I_B0, I_d, T, eta_c, eta_m, eta_v, eta_a, v_exh};
    hipLaunchKernelGGL(thruster_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                       (long long)n, V_a, V_cc, mdot_a, a_1, o);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

// ---- plume -------------------------------------------------------------------------------------
int pem_plume_f64_dev(size_t n, int n_radii, const double* radii, double torr2pa, const double* P_b, const double* c0,
                      const double* c1, const double* c2, const double* c3, const double* c4, const double* c5,
                      const double* sigma_cex, const double* I_B0, const double* T, double* j_ion, double* div_angle,
                      double* T_c, uint8_t* invalid, pem_stream_t stream) {
    if (n_radii < 1 || !radii) return fail(PEM_ERR_INVALID_ARG, "pem_plume: need at least one sweep radius");
    if (n == 0) return PEM_OK;
    if (!P_b || !c0 || !c1 || !c2 || !c3 || !c4 || !c5 || !sigma_cex || !I_B0 || !j_ion || !div_angle)
        return fail(PEM_ERR_INVALID_ARG, "pem_plume: NULL array");
    if ((T == nullptr) != (T_c == nullptr)) return fail(PEM_ERR_INVALID_ARG, "pem_plume: T and T_c go together");
    if (int rc = check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    PlumeIO io{(long long)n, torr2pa, radii[0], P_b, c0, c1, c2, c3, c4, c5, sigma_cex, I_B0, T, j_ion, div_angle, T_c, invalid};
    if (n_radii == 1 && aligned16(j_ion)) return dispatch_lanes<false, true>(io, CoupledIO{}, st);

    // general path: radii go to the device through a small stream-ordered allocation
    double* d_radii = nullptr;
    HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&d_radii), sizeof(double) * n_radii, st));
    HIP_TRY(hipMemcpyAsync(d_radii, radii, sizeof(double) * n_radii, hipMemcpyHostToDevice, st));
    const size_t blocks = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(plume_generic_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, st, io, d_radii, n_radii);
    hipError_t le = hipGetLastError();
    HIP_TRY(hipFreeAsync(d_radii, st));
    HIP_TRY(le);
    // `radii` is host memory owned by the caller: make sure the copy has left it before returning
    HIP_TRY(hipStreamSynchronize(st));
    return PEM_OK;
}

// ---- coupled -----------------------------------------------------------------------------------
int pem_coupled_f64_dev(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a, const double* T_e,
                        const double* V_vac, const double* Pstar, const double* P_T, const double* mdot_a,
                        const double* a_1, const double* c0, const double* c1, const double* c2, const double* c3,
                        const double* c4, const double* c5, const double* sigma_cex, double* V_cc, double* I_B0,
                        double* T, double* j_ion, double* div_angle, double* T_c, uint8_t* invalid, pem_stream_t stream) {
    if (n == 0) return PEM_OK;
    if (!P_b || !V_a || !T_e || !V_vac || !Pstar || !P_T || !mdot_a || !a_1 || !c0 || !c1 || !c2 || !c3 || !c4 || !c5 ||
        !sigma_cex || !V_cc || !div_angle || !T_c)
        return fail(PEM_ERR_INVALID_ARG, "pem_coupled: NULL array");
    if (j_ion && !aligned16(j_ion)) return fail(PEM_ERR_INVALID_ARG, "pem_coupled: j_ion must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    PlumeIO io{(long long)n, torr2pa, radius, P_b, c0, c1, c2, c3, c4, c5, sigma_cex, nullptr, nullptr, j_ion, div_angle, T_c, invalid};
    CoupledIO cio{V_a, T_e, V_vac, Pstar, P_T, mdot_a, a_1, V_cc, I_B0, T};
    return j_ion ? dispatch_lanes<true, true>(io, cio, st) : dispatch_lanes<true, false>(io, cio, st);
}

// =============================================================================================
// host-pointer entry points: stage through the device workspace in chunks
// =============================================================================================
int pem_cathode_f64(size_t n, const double* P_b, const double* V_a, const double* T_e, const double* V_vac,
                    const double* Pstar, const double* P_T, double torr2pa, double* V_cc) {
    if (n == 0) return PEM_OK;
    if (!P_b || !V_a || !T_e || !V_vac || !Pstar || !P_T || !V_cc) return fail(PEM_ERR_INVALID_ARG, "pem_cathode: NULL array");
    if (int rc = check_device()) return rc;
    std::lock_guard<std::mutex> lock(g_ws.mu);
    const size_t chunk = n < (size_t(1) << 24) ? n : (size_t(1) << 24);
    if (int rc = g_ws.reserve(7 * padded(chunk * 8))) return rc;
    const double* in[6] = {P_b, V_a, T_e, V_vac, Pstar, P_T};
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = (n - off < chunk) ? n - off : chunk;
        Carver cv(g_ws.buf);
        double* d[7];
        for (auto& p : d) p = cv.take<double>(chunk);
        for (int i = 0; i < 6; ++i) HIP_TRY(hipMemcpyAsync(d[i], in[i] + off, m * 8, hipMemcpyHostToDevice, nullptr));
        if (int rc = pem_cathode_f64_dev(m, d[0], d[1], d[2], d[3], d[4], d[5], torr2pa, d[6], nullptr)) return rc;
        HIP_TRY(hipMemcpyAsync(V_cc + off, d[6], m * 8, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return PEM_OK;
}

int pem_thruster_f64(size_t n, const double* V_a, const double* V_cc, const double* mdot_a, const double* a_1,
                     double* I_B0, double* I_d, double* T, double* eta_c, double* eta_m, double* eta_v, double* eta_a,
                     double* v_exh) {
    if (n == 0) return PEM_OK;
    if (!V_a || !V_cc || !mdot_a || !a_1) return fail(PEM_ERR_INVALID_ARG, "pem_thruster: NULL input array");
    if (int rc = check_device()) return rc;
    std::lock_guard<std::mutex> lock(g_ws.mu);
    const size_t chunk = n < (size_t(1) << 24) ? n : (size_t(1) << 24);
    if (int rc = g_ws.reserve(12 * padded(chunk * 8))) return rc;
    const double* in[4] = {V_a, V_cc, mdot_a, a_1};
    double* out[8] = {I_B0, I_d, T, eta_c, eta_m, eta_v, eta_a, v_exh};
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = (n - off < chunk) ? n - off : chunk;
        Carver cv(g_ws.buf);
        double *di[4], *dout[8];
        for (auto& p : di) p = cv.take<double>(chunk);
        for (int i = 0; i < 8; ++i) dout[i] = out[i] ? cv.take<double>(chunk) : nullptr;
        for (int i = 0; i < 4; ++i) HIP_TRY(hipMemcpyAsync(di[i], in[i] + off, m * 8, hipMemcpyHostToDevice, nullptr));
        if (int rc = pem_thruster_f64_dev(m, di[0], di[1], di[2], di[3], dout[0], dout[1], dout[2], dout[3], dout[4],
                                          dout[5], dout[6], dout[7], nullptr))
            return rc;
        for (int i = 0; i < 8; ++i)
            if (out[i]) HIP_TRY(hipMemcpyAsync(out[i] + off, dout[i], m * 8, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return PEM_OK;
}

int pem_plume_f64(size_t n, int n_radii, const double* radii, double torr2pa, const double* P_b, const double* c0,
                  const double* c1, const double* c2, const double* c3, const double* c4, const double* c5,
                  const double* sigma_cex, const double* I_B0, const double* T, double* j_ion, double* div_angle,
                  double* T_c, uint8_t* invalid) {
    if (n_radii < 1 || !radii) return fail(PEM_ERR_INVALID_ARG, "pem_plume: need at least one sweep radius");
    if (n == 0) return PEM_OK;
    if (!P_b || !c0 || !c1 || !c2 || !c3 || !c4 || !c5 || !sigma_cex || !I_B0 || !j_ion || !div_angle)
        return fail(PEM_ERR_INVALID_ARG, "pem_plume: NULL array");
    if ((T == nullptr) != (T_c == nullptr)) return fail(PEM_ERR_INVALID_ARG, "pem_plume: T and T_c go together");
    if (int rc = check_device()) return rc;
    std::lock_guard<std::mutex> lock(g_ws.mu);
    const size_t R = (size_t)n_radii;
    // bound the profile chunk to ~256 MiB of device memory
    size_t chunk = (size_t(1) << 28) / (NANG * R * 8);
    if (chunk < 1024) chunk = 1024;
    if (chunk > n) chunk = n;
    chunk = (chunk + 63) & ~size_t(63);
    const size_t need = 10 * padded(chunk * 8) + padded(chunk * NANG * R * 8) + 2 * padded(chunk * R * 8) + padded(chunk);
    if (int rc = g_ws.reserve(need)) return rc;
    const double* in[10] = {P_b, c0, c1, c2, c3, c4, c5, sigma_cex, I_B0, T};
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = (n - off < chunk) ? n - off : chunk;
        Carver cv(g_ws.buf);
        double* d[10];
        for (auto& p : d) p = cv.take<double>(chunk);
        double* dj = cv.take<double>(chunk * NANG * R);
        double* ddiv = cv.take<double>(chunk * R);
        double* dtc = cv.take<double>(chunk * R);
        uint8_t* dinv = cv.take<uint8_t>(chunk);
        for (int i = 0; i < 10; ++i)
            if (in[i]) HIP_TRY(hipMemcpyAsync(d[i], in[i] + off, m * 8, hipMemcpyHostToDevice, nullptr));
        if (int rc = pem_plume_f64_dev(m, n_radii, radii, torr2pa, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8],
                                       T ? d[9] : nullptr, dj, ddiv, T ? dtc : nullptr, invalid ? dinv : nullptr, nullptr))
            return rc;
        HIP_TRY(hipMemcpyAsync(j_ion + off * NANG * R, dj, m * NANG * R * 8, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipMemcpyAsync(div_angle + off * R, ddiv, m * R * 8, hipMemcpyDeviceToHost, nullptr));
        if (T) HIP_TRY(hipMemcpyAsync(T_c + off * R, dtc, m * R * 8, hipMemcpyDeviceToHost, nullptr));
        if (invalid) HIP_TRY(hipMemcpyAsync(invalid + off, dinv, m, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return PEM_OK;
}

int pem_coupled_f64(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a, const double* T_e,
                    const double* V_vac, const double* Pstar, const double* P_T, const double* mdot_a, const double* a_1,
                    const double* c0, const double* c1, const double* c2, const double* c3, const double* c4,
                    const double* c5, const double* sigma_cex, double* V_cc, double* I_B0, double* T, double* j_ion,
                    double* div_angle, double* T_c, uint8_t* invalid) {
    if (n == 0) return PEM_OK;
    const double* in[15] = {P_b, V_a, T_e, V_vac, Pstar, P_T, mdot_a, a_1, c0, c1, c2, c3, c4, c5, sigma_cex};
    for (auto p : in)
        if (!p) return fail(PEM_ERR_INVALID_ARG, "pem_coupled: NULL input array");
    if (!V_cc || !div_angle || !T_c) return fail(PEM_ERR_INVALID_ARG, "pem_coupled: NULL output array");
    if (int rc = check_device()) return rc;
    std::lock_guard<std::mutex> lock(g_ws.mu);
    size_t chunk = (size_t(1) << 28) / (NANG * 8);
    if (chunk > n) chunk = n;
    chunk = (chunk + 63) & ~size_t(63);
    const size_t need = 20 * padded(chunk * 8) + padded(chunk * NANG * 8) + padded(chunk);
    if (int rc = g_ws.reserve(need)) return rc;
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = (n - off < chunk) ? n - off : chunk;
        Carver cv(g_ws.buf);
        double* d[15];
        for (auto& p : d) p = cv.take<double>(chunk);
        double* dvcc = cv.take<double>(chunk);
        double* dib0 = cv.take<double>(chunk);
        double* dT = cv.take<double>(chunk);
        double* ddiv = cv.take<double>(chunk);
        double* dtc = cv.take<double>(chunk);
        double* dj = cv.take<double>(chunk * NANG);
        uint8_t* dinv = cv.take<uint8_t>(chunk);
        for (int i = 0; i < 15; ++i) HIP_TRY(hipMemcpyAsync(d[i], in[i] + off, m * 8, hipMemcpyHostToDevice, nullptr));
        if (int rc = pem_coupled_f64_dev(m, torr2pa, radius, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9],
                                         d[10], d[11], d[12], d[13], d[14], dvcc, I_B0 ? dib0 : nullptr, T ? dT : nullptr,
                                         j_ion ? dj : nullptr, ddiv, dtc, invalid ? dinv : nullptr, nullptr))
            return rc;
        HIP_TRY(hipMemcpyAsync(V_cc + off, dvcc, m * 8, hipMemcpyDeviceToHost, nullptr));
        if (I_B0) HIP_TRY(hipMemcpyAsync(I_B0 + off, dib0, m * 8, hipMemcpyDeviceToHost, nullptr));
        if (T) HIP_TRY(hipMemcpyAsync(T + off, dT, m * 8, hipMemcpyDeviceToHost, nullptr));
        if (j_ion) HIP_TRY(hipMemcpyAsync(j_ion + off * NANG, dj, m * NANG * 8, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipMemcpyAsync(div_angle + off, ddiv, m * 8, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipMemcpyAsync(T_c + off, dtc, m * 8, hipMemcpyDeviceToHost, nullptr));
        if (invalid) HIP_TRY(hipMemcpyAsync(invalid + off, dinv, m, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    return PEM_OK;
}

}  // extern "C"
