// pem_latent.hip -- coupled PEM-v0 evaluation + SVD compression of the profile in one launch, ONE LANE PER SAMPLE (gfx950).
//
//   latent[i][r] = sum_k norm(j_ion[i][k]) basis[k][r]        (compression.py-shaped map of pem_v0_SPT-100.yml:207-214;
//                                                              plume.py:39-140 for j_ion, div_angle, T_c)
//
// Why not a mode of plume_r1_kernel (csrc/pem_kernels.hip), which spreads a sample over four lanes (it was one until
// round 2, 386-406 us per 1.25e6 samples against 185-198 us for this kernel):
// with four lanes per sample the four chunk lanes of a wave instruction sit at four different angles, so the basis row
// of "this angle" is per-lane data -- four ds_read_b128 per lane and angle, and the loop was bound by those LDS reads
// (a per-lane log10 table gather on top of them made it slower; profiles/svd_probe_r02c.txt).  With one lane per
// sample every lane of a wave is at the SAME angle: the basis row and the Simpson weights are wave-uniform and come
// through the scalar cache into SGPRs (s_load, no LDS, no VGPRs), which leaves the LDS to the table log10 of
// pem_math.h (13 fp64 + 12 fp32-rate instructions against ~40 fp64 for the series) and drops the cross-lane sums.
// The matrix pipe is no help here: v_mfma_f64 shares the SIMD's fp64 units with the VALU on this chip (times add,
// profiles/mfma_valu_overlap_r02w.txt) and a rank <= 8 contraction pads to 16 columns.
//
// The Gaussians are the recurrences of plume_r1_kernel, restarted every 23 angles from the same coarse recurrence
// (chunk starts 0, 23, 46, 69), and the Simpson sums are accumulated per chunk and added in chunk order, so j_ion,
// den and num see the operations of the four-lane kernel.  A sample whose recurrence runs into the deep tail (smallest
// j_ion below 1e-290, or a non-finite amplitude) is redone literally -- direct exp() per angle, every product and sum
// rounded separately as plume.py:99-102 does -- by literal_sample(); under the PEM-v0 priors that never runs.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"
#include "pem_math.h"
#include "pem_model.h"

namespace {

using namespace pem_model;

struct LatentArgs {
    long long n;
    double torr2pa, radius;
    const double *P_b, *V_a, *T_e, *V_vac, *Pstar, *P_T, *mdot_a, *a_1, *c0, *c1, *c2, *c3, *c4, *c5, *sigma;
    double *V_cc, *div, *Tc;
    uint8_t* invalid;
};

constexpr int CHUNK = 23;          // angles between restarts of the recurrences = the chunk of plume_r1_kernel<4, ...>
// tuning knobs (tools/build_variant.sh -DPEM_LAT_...=; #pragma unroll wants a constant expression, not a macro)
#ifndef PEM_LAT_BLOCK
#define PEM_LAT_BLOCK 256
#endif
#ifndef PEM_LAT_UNROLL
#define PEM_LAT_UNROLL 2
#endif
#ifndef PEM_LAT_WAVES
#define PEM_LAT_WAVES 4      // waves per SIMD the register allocator leaves room for
#endif
constexpr int LAT_BLOCK = PEM_LAT_BLOCK;
constexpr int LAT_UNROLL = PEM_LAT_UNROLL;
constexpr int OUT_STRIDE = 65;     // doubles per column of a wave's output staging block (odd: the transposed reads spread over the banks)
typedef double lat_f64x2 __attribute__((ext_vector_type(2)));

// The slow, literal evaluation of one sample (see exact_chunk / exact_latents in pem_kernels.hip): chunk-wise partial
// Simpson sums added in chunk order, like the fast loop.  Out of line: it must not cost the fast loop registers.
// (round 4: inlined at its one call site and writing straight into the caller's accumulators.  Out of line it handed its 3 + RANK
// results back through a struct on the stack -- 96 bytes of scratch per lane in every instantiation, for a path the priors never
// take; inlined it reuses the fast loop's registers, which are dead by then.)
template <int RANK, bool LOGN>
__device__ __forceinline__ void literal_sample(double X1a, double X2a, double jcex, double a1, double a2,
                                               const double* __restrict__ basis, double& den, double& num, double& lo, double (&lat)[RANK]) {
#pragma clang fp contract(off)
    den = 0.0;
    num = 0.0;
    lo = __builtin_inf();
#pragma unroll
    for (int q = 0; q < RANK; ++q) lat[q] = 0.0;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        double dc = 0.0, nc = 0.0;
#pragma unroll 1
        for (int j = 0; j < CHUNK; ++j) {
            const int k = c * CHUNK + j;
            if (k >= NANG) break;
            const double alpha = k == NANG - 1 ? HALF_PI : (double)k * GRID_H;   // np.linspace(0, pi/2, 91)
            const double t1 = alpha / a1, t2 = alpha / a2;
            const double f = X1a * exp(-(t1 * t1)) + X2a * exp(-(t2 * t2));
            const double ji = f + jcex;
            lo = fmin(lo, f);
            dc = __builtin_fma(PEM_SIMPSON_CDEN[k], f, dc);
            nc = __builtin_fma(PEM_SIMPSON_CNUM[k], f, nc);
            const double lj = LOGN ? pem::pem_log10(ji) : ji;
#pragma unroll
            for (int q = 0; q < RANK; ++q) lat[q] = __builtin_fma(lj, basis[k * RANK + q], lat[q]);
        }
        den += dc;
        num += nc;
    }
}

template <int RANK, bool LOGN>
__global__ __launch_bounds__(LAT_BLOCK) __attribute__((amdgpu_waves_per_eu(PEM_LAT_WAVES))) void coupled_latent_kernel(LatentArgs a, const double* __restrict__ basis,
                                                                    double* __restrict__ latent) {
    __shared__ __attribute__((aligned(16))) double logtab[LOGN ? pem::LOG_TABLE_DOUBLES : 2];
    __shared__ __attribute__((aligned(16))) double outbuf[LAT_BLOCK / 64][RANK * OUT_STRIDE];   // per wave: [column][sample]
    const long long g = (long long)blockIdx.x * LAT_BLOCK + threadIdx.x;
    const bool live = g < a.n;
    const long long gi = live ? g : a.n - 1;   // dead lanes recompute the last sample and store nothing
    // the sample's inputs are requested before the table copy so that the two latencies overlap
    const double P_b = a.P_b[gi], V_a = a.V_a[gi], T_e = a.T_e[gi], V_vac = a.V_vac[gi], Pstar = a.Pstar[gi], P_T = a.P_T[gi];
    const double mdot = a.mdot_a[gi], a_1 = a.a_1[gi], c0 = a.c0[gi], c1 = a.c1[gi], c2 = a.c2[gi], c3 = a.c3[gi], c4 = a.c4[gi];
    const double c5 = a.c5[gi], sigma = a.sigma[gi];
    if constexpr (LOGN) {
        pem::load_log_table(logtab, threadIdx.x, LAT_BLOCK);
        __syncthreads();
    }

    // ------------------------------ prelude: cathode, thruster, plume set-up (as process_tile) ------------------------------
    const double V_cc = cathode_vcc(P_b, V_a, T_e, V_vac, Pstar, P_T, a.torr2pa);
    const ThrusterQoI th = thruster_stage(V_a, V_cc, mdot, a_1);
    const PlumeSetup ps = plume_setup(P_b, c1, c2, c3, c4, c5, a.torr2pa);
    const double a1 = ps.a1, a2 = ps.a2;
    const double u1 = 1.0 / (a1 * a1), u2 = 1.0 / (a2 * a2);
    const double A1 = (1.0 - c0) / normaliser(a1, u1, PEM_DPOLY);   // plume.py:64-73
    const double A2 = c0 / normaliser(a2, u2, PEM_DPOLY);           // plume.py:75-85
    const double rad = a.radius;
    const double inv_r2 = 1.0 / (rad * rad), inv_2pi_r2 = 1.0 / (2.0 * PEM_PI * (rad * rad));
    const double decay = exp(-rad * ps.n_neutral * sigma);   // plume.py:95-100 at the single radius
    const double jcex = th.I_B0 * (1.0 - decay) * inv_2pi_r2;
    const double base = th.I_B0 * decay * inv_r2;
    const double X1a = base * A1, X2a = base * A2;

    // Gaussian recurrences e_k = exp(-k^2 s), s = (h/a)^2, restarted at k = 0, 23, 46, 69 by the coarse recurrence
    //   e_{23c} = E^(c^2), r_{23c} = exp(-(2*23c + 1) s) = r0 G^c           (plume_r1_kernel, "ROUNDS")
    const double s1 = (GRID_H * GRID_H) * u1, s2 = (GRID_H * GRID_H) * u2;
    const double r01 = exp_nonpos(-s1), G1 = exp_nonpos(-(2.0 * CHUNK) * s1), E1 = exp_nonpos(-(double)(CHUNK * CHUNK) * s1);
    const double r02 = exp_nonpos(-s2), G2 = exp_nonpos(-(2.0 * CHUNK) * s2), E2 = exp_nonpos(-(double)(CHUNK * CHUNK) * s2);
    const double q1 = r01 * r01, q2 = r02 * r02, E1sq = E1 * E1, E2sq = E2 * E2;

    double lat[RANK];
#pragma unroll
    for (int r = 0; r < RANK; ++r) lat[r] = 0.0;
    double den = 0.0, num = 0.0, lo = __builtin_inf();
    // The loop takes log10 of j_ion without that function's special cases, so everything but "positive and finite at every
    // angle" has to end in the literal evaluation: |f_k| <= |X1a| + |X2a| (the Gaussians are <= 1), so a bounded start
    // keeps every j_ion finite; NaN amplitudes and the non-positive ones are caught by the tests behind the loop.
    bool uncertain = !(fabs(X1a) + fabs(X2a) + fabs(jcex) < 1e300);
    double Xc1 = X1a, Xc2 = X2a, rho1 = E1, rho2 = E2, rc1 = r01, rc2 = r02;
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        double X1 = Xc1, X2 = Xc2, rr1 = rc1, rr2 = rc2, dc = 0.0, nc = 0.0;
        const int k0 = c * CHUNK, nk = c < 3 ? CHUNK : NANG - 3 * CHUNK;
#pragma unroll LAT_UNROLL
        for (int j = 0; j < nk; ++j) {
            const int k = k0 + j;                       // wave-uniform: weights and basis row are scalar loads
            const double f = X1 + X2;                   // j_beam + j_scat
            const double ji = f + jcex;                 // plume.py:102
            lo = fmin(lo, f);
            dc = fma(PEM_SIMPSON_CDEN[k], f, dc);
            nc = fma(PEM_SIMPSON_CNUM[k], f, nc);
            const double lj = LOGN ? pem::pem_log10_tab_pos(ji, logtab) : ji;
#pragma unroll
            for (int r = 0; r < RANK; ++r) lat[r] = fma(lj, basis[k * RANK + r], lat[r]);
            X1 *= rr1;
            rr1 *= q1;
            X2 *= rr2;
            rr2 *= q2;
        }
        // (an infinite amplitude -- exp(+x) overflow for a negative density -- must turn into NaN where the reference's
        // exp() is exactly zero: the literal evaluation decides)
        uncertain = uncertain || !__builtin_isfinite(X1) || !__builtin_isfinite(X2);
        den += dc;
        num += nc;
        Xc1 *= rho1;
        rho1 *= E1sq;
        rc1 *= G1;
        Xc2 *= rho2;
        rho2 *= E2sq;
        rc2 *= G2;
    }
    uncertain = uncertain || !((lo + jcex) >= 1e-290);   // (written so that a NaN is uncertain as well)
    if (__ballot(uncertain)) {   // rare; under the PEM-v0 priors j_cex > 1e-6 and this never runs
        if (uncertain) {
            literal_sample<RANK, LOGN>(X1a, X2a, jcex, a1, a2, basis, den, num, lo, lat);
        }
    }
    // plume.py:105: invalid if alpha1 <= 0 or any j_ion <= 0 (NaN compares false); min_k fl(f_k + c) = fl(min_k f_k + c)
    const bool bad = a1 <= 0.0 || (lo + jcex) <= 0.0;
    if (__ballot(bad)) {         // plume.py:106: the profile of an invalid sample is 1e-20 everywhere -> norm(1e-20) x column sums
        if (bad) {
            const double fill = LOGN ? -20.0 : 1e-20;
#pragma unroll
            for (int r = 0; r < RANK; ++r) {
                double sum = 0.0;
                for (int k = 0; k < NANG; ++k) sum += basis[k * RANK + r];
                lat[r] = fill * sum;
            }
        }
    }

    // ------------------------------ epilogue ------------------------------
    double cos_div = num / den;  // plume.py:124-127
    if (cos_div == __builtin_inf()) cos_div = __builtin_nan("");
    // latents: the wave's 64 x RANK block is contiguous in memory; through LDS it leaves as whole 16-byte pieces
    const int lane = threadIdx.x & 63;
    const long long first = g - lane;
    if (first + 64 <= a.n && (reinterpret_cast<uintptr_t>(latent) & 15) == 0) {
        double* ob = outbuf[threadIdx.x >> 6];
#pragma unroll
        for (int r = 0; r < RANK; ++r) ob[r * OUT_STRIDE + lane] = lat[r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        lat_f64x2* dst = reinterpret_cast<lat_f64x2*>(latent + first * RANK);
#pragma unroll
        for (int p = lane; p < 32 * RANK; p += 64) {
            const int e0 = 2 * p, e1 = 2 * p + 1;
            lat_f64x2 v;
            v.x = ob[(e0 % RANK) * OUT_STRIDE + e0 / RANK];
            v.y = ob[(e1 % RANK) * OUT_STRIDE + e1 / RANK];
            __builtin_nontemporal_store(v, &dst[p]);
        }
    } else if (live) {
#pragma unroll
        for (int r = 0; r < RANK; ++r) latent[g * RANK + r] = lat[r];
    }
    if (live) {
        a.div[g] = acos(cos_div);
        a.Tc[g] = th.T * cos_div;
        a.V_cc[g] = V_cc;
        if (a.invalid) a.invalid[g] = (uint8_t)bad;
    }
}

template <int RANK>
int launch_rank(const LatentArgs& a, const double* basis, double* latent, bool log_norm, hipStream_t st) {
    const unsigned blocks = (unsigned)((a.n + LAT_BLOCK - 1) / LAT_BLOCK);
    if (log_norm) hipLaunchKernelGGL((coupled_latent_kernel<RANK, true>), dim3(blocks), dim3(LAT_BLOCK), 0, st, a, basis, latent);
    else hipLaunchKernelGGL((coupled_latent_kernel<RANK, false>), dim3(blocks), dim3(LAT_BLOCK), 0, st, a, basis, latent);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // namespace

extern "C" {

int pem_coupled_latent_f64_dev(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a,
                               const double* T_e, const double* V_vac, const double* Pstar, const double* P_T,
                               const double* mdot_a, const double* a_1, const double* c0, const double* c1, const double* c2,
                               const double* c3, const double* c4, const double* c5, const double* sigma_cex, int rank,
                               int norm, const double* basis, double* latent, double* V_cc, double* div_angle, double* T_c,
                               uint8_t* invalid, pem_stream_t stream) {
    if (rank < 1 || rank > PEM_FUSED_LATENT_MAX_RANK)
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_coupled_latent: 1 <= rank <= %d", PEM_FUSED_LATENT_MAX_RANK);
    if (norm != PEM_NORM_NONE && norm != PEM_NORM_LOG10)
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_coupled_latent: norm must be PEM_NORM_NONE or PEM_NORM_LOG10");
    if (n == 0) return PEM_OK;
    if (!P_b || !V_a || !T_e || !V_vac || !Pstar || !P_T || !mdot_a || !a_1 || !c0 || !c1 || !c2 || !c3 || !c4 || !c5 ||
        !sigma_cex || !basis || !latent || !V_cc || !div_angle || !T_c)
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_coupled_latent: NULL array");
    if (int rc = pem::check_device()) return rc;
    const LatentArgs a{(long long)n, torr2pa, radius, P_b, V_a, T_e, V_vac, Pstar, P_T, mdot_a, a_1, c0, c1, c2, c3, c4, c5,
                       sigma_cex, V_cc, div_angle, T_c, invalid};
    const bool log_norm = norm == PEM_NORM_LOG10;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (rank) {
        case 1: return launch_rank<1>(a, basis, latent, log_norm, st);
        case 2: return launch_rank<2>(a, basis, latent, log_norm, st);
        case 3: return launch_rank<3>(a, basis, latent, log_norm, st);
        case 4: return launch_rank<4>(a, basis, latent, log_norm, st);
        case 5: return launch_rank<5>(a, basis, latent, log_norm, st);
        case 6: return launch_rank<6>(a, basis, latent, log_norm, st);
        case 7: return launch_rank<7>(a, basis, latent, log_norm, st);
        default: return launch_rank<8>(a, basis, latent, log_norm, st);
    }
}

}  // extern "C"
