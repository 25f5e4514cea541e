// pem_kernels.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI of libpem_hip.so.
//
// What is computed (reference file:line, upstream repository root):
//   cathode stage    src/hallmd/models/cathode.py:24-38
//   thruster stage   tests/sim_hallthruster.jl:35-48 (the reference's analytic test double)
//   plume stage      src/hallmd/models/plume.py:39-140
//
// Kernel shape (DESIGN.md holds the measurements behind each choice).  The path is a streaming map,
// 120 B in and 752 B out per sample; the first version was VALU-bound on fp64 transcendentals
// (1246 VALU instructions per 16 samples), so the design removes them until HBM writes are the bound:
//   * One 64-lane wave owns 64 consecutive samples per tile and is persistent over tiles; a workgroup is 4 such
//     waves that share two LDS tables and nothing else (8 waves per CU).
//   * PRELUDE, one lane per sample, nothing redundant: cathode + thruster stages, alpha1/alpha2, the two
//     normalisers, beam amplitudes.  The normaliser D(a) = 2 pi Int_0^{pi/2} exp(-(t/a)^2) sin t dt
//     (identical to the six complex erfi of plume.py:64-85) is a degree-11 polynomial in u = 1/a^2 from a
//     32-interval LDS table (|a| >= 0.25) or a 10-term series (|a| < 0.25) -- no quadrature, no erfi.
//   * ROUNDS: the wave then walks its 64 samples in L rounds of S = 64/L samples; in a round lane
//     (s, c) produces angles k = c*CH .. c*CH+CH-1 (CH = ceil(91/L)) of sample s.  The two Gaussians
//     exp(-(k h / a)^2) advance along k by the two-term recurrence e_{k+1} = e_k r_k, r_{k+1} = r_k q,
//     and the chunk starts come from the same recurrence at stride CH -- so a sample costs 7 exp in
//     total (3 per beam + the CEX decay) instead of 182 + 6 erfi.
//   * The Simpson sums of plume.py:117-123 ride in the same loop (folded weights from an LDS table).
//   * A round's S x 91 profile block is contiguous in j_ion (R = 1): it is staged in LDS in final order
//     and leaves as 16-byte-per-lane, 1-KiB-per-instruction stores.
//   * EPILOGUE, one lane per sample again: cos_div, arccos, T_c, coalesced 512-byte stores.
//   * Without a profile (reduced-QoI mode) the rounds are skipped: the divergence integrals come from tabulated
//     Simpson functionals of the beam width (simpson_functionals), per sample, wherever that is exact to rounding.
//   * Where the reference's own exp() has left the normal range (deep tail of a narrow beam with j_cex = 0, infinite
//     amplitudes) a chunk is re-evaluated literally (exact_chunk): "equal to the reference" beats "accurate" there.
//   * The same tile loop carries the fused modes: Monte-Carlo inputs generated in the prelude (MC) and the likelihood of
//     measured current densities (JMODE 3) consuming the profile on chip.  (The fused SVD compression is a lane-per-sample
//     kernel of its own, csrc/pem_latent.hip.)
//   Other kernels in this file: plume_rfew_kernel (2..8 sweep radii: this design generalised), plume_radii_kernel /
//   plume_generic_kernel (more radii), cathode, thruster, u_ion profile and the post-run filters.  The per-sample scalar
//   stages and tables are csrc/pem_model.h (shared with the lane-per-sample kernels, csrc/pem_saltelli.hip and csrc/pem_latent.hip).
//
// This file is written for gfx950 only: 64-wide waves, 160 KiB LDS, no portability layer.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <functional>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>

#include "pem_common.h"
#include "pem_math.h"
#include "pem_hip.h"
#include "pem_philox.h"
#include "pem_qfused.h"

#include "pem_model.h"

namespace {

using namespace pem_model;

constexpr int BLOCK = 256;                     // elementwise kernels
constexpr int WAVE = 64;

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
    float* j_ion_f32;  // mixed mode: the profile is computed in fp64 and stored as fp32
    // fused likelihood mode (JMODE 3): measurement tables [n_cond][n_ang] and the per-sample result
    const int32_t* m_kidx;
    const double *m_wgt, *m_y, *m_inv_std;
    double* loglik;
    int n_cond, n_ang;
    // Where sample g of an input array lives: ptr[(g / 64) * in_tile_stride + g % 64].  64 = plain SoA arrays (every entry
    // point but one); 15 * 64 = the tile-interleaved layout of pem_coupled_tiled_f64_dev, whose 15 "arrays" are the rows of
    // one [tiles][15][64] block (R = 1 fast path only: the other plume kernels index the arrays directly).
    long long in_tile_stride = 64;
    // counting modes (JMODE 4 / 5): brackets in, counts and records out (csrc/pem_qfused.h)
    pem::CountIO q;
};

struct CoupledIO {
    const double *V_a, *T_e, *V_vac, *Pstar, *P_T, *mdot_a, *a_1;
    double *V_cc, *I_B0, *T;
};

// fused Monte-Carlo mode: the 15 coupled inputs are generated in registers from the counter-based design instead
// of being read from HBM (order of COUPLED inputs: P_b V_a T_e V_vac Pstar P_T mdot_a a_1 c0..c5 sigma_cex)
struct McDesign {
    unsigned long long seed, first;
    unsigned int stream;
    int swap_dim;           // Saltelli blocks, as pem_sample_f64_dev: -1 plain, -2 all from stream+1, d >= 0 column d
    int kind[15];
    double a[15], b[15];
    double* x_out;          // optional [15][ld] copy of the generated inputs
    unsigned long long ld;
};

// the per-sample inputs of one lane, prefetched one tile ahead
template <bool COUPLED>
struct SampleIn {
    double P_b, c0, c1, c2, c3, c4, c5, sigma;
    double x0, x1, x2, x3, x4, x5, x6;  // COUPLED: V_a T_e V_vac Pstar P_T mdot_a a_1 ; else I_B0, T, unused
};

// read-once input stream
__device__ __forceinline__ double stream_load(const double* p) {
#if defined(PEM_NT_LOADS) && PEM_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

__device__ __forceinline__ void stream_store1(double v, double* p) {
#if defined(PEM_NT_QOI) && PEM_NT_QOI
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

template <bool COUPLED>
__device__ __forceinline__ SampleIn<COUPLED> load_sample(const PlumeIO& io, const CoupledIO& cio, long long gi) {
    SampleIn<COUPLED> v;
    gi = (gi >> 6) * io.in_tile_stride + (gi & 63);
    v.P_b = stream_load(io.P_b + gi);
    v.c0 = stream_load(io.c0 + gi);
    v.c1 = stream_load(io.c1 + gi);
    v.c2 = stream_load(io.c2 + gi);
    v.c3 = stream_load(io.c3 + gi);
    v.c4 = stream_load(io.c4 + gi);
    v.c5 = stream_load(io.c5 + gi);
    v.sigma = stream_load(io.sigma + gi);
    if constexpr (COUPLED) {
        v.x0 = stream_load(cio.V_a + gi);
        v.x1 = stream_load(cio.T_e + gi);
        v.x2 = stream_load(cio.V_vac + gi);
        v.x3 = stream_load(cio.Pstar + gi);
        v.x4 = stream_load(cio.P_T + gi);
        v.x5 = stream_load(cio.mdot_a + gi);
        v.x6 = stream_load(cio.a_1 + gi);
    } else {
        v.x0 = stream_load(io.I_B0 + gi);
        v.x1 = io.T ? stream_load(io.T + gi) : 0.0;
        v.x2 = v.x3 = v.x4 = v.x5 = v.x6 = 0.0;
    }
    return v;
}

// Out of line on purpose: inlined fifteen times, the library exp / normcdfinv bodies of the transforms pushed the
// fused kernel past 256 VGPRs (204 bytes of scratch per lane).
__device__ __attribute__((noinline)) double transform_call(int kind, double a, double b, double u) {
    return pem::transform(kind, a, b, u);
}

// inputs of global sample `g` from the design: bit-identical to pem_sample_f64_dev followed by a load
// `design`: the LDS copy of the design's 15 x (a, b, kind) made at kernel start (MC_LDS_DOUBLES: a[15] | b[15] | pad |
// kind[15] as ints).  Read from the kernel arguments inside the tile loop instead, those 75 SGPRs were spilled to VGPR
// lanes and restored around every out-of-line transform call: ~370 v_readlane / v_writelane per tile (310 SGPR spills).
constexpr int MC_LDS_DOUBLES = 48;
__device__ __forceinline__ SampleIn<true> generate_sample(const McDesign& mc, const double* design, long long g_local) {
    const int* design_kind = reinterpret_cast<const int*>(design + 32);
    const unsigned long long g = mc.first + (unsigned long long)g_local;
    const unsigned int k0 = (unsigned int)mc.seed, k1 = (unsigned int)(mc.seed >> 32);
    double x[16];
    // All eight Philox blocks first: they are independent chains of quarter-rate 32x32 multiplies, and the out-of-line
    // transform calls between them would keep the scheduler from interleaving them.
    double u[16];
#pragma unroll
    for (int pair = 0; pair < 8; ++pair) {
        // the two dimensions of a pair may come from different streams in a Saltelli block (wave-uniform choice)
        const unsigned int st0 = mc.stream + ((mc.swap_dim == -2 || mc.swap_dim == 2 * pair) ? 1u : 0u);
        const unsigned int st1 = mc.stream + ((mc.swap_dim == -2 || mc.swap_dim == 2 * pair + 1) ? 1u : 0u);
        const pem::Philox4 r = pem::philox4x32_10((unsigned int)g, (unsigned int)(g >> 32), (unsigned int)pair, st0, k0, k1);
        pem::Philox4 r1 = r;
        if (2 * pair + 1 < 15 && st1 != st0)
            r1 = pem::philox4x32_10((unsigned int)g, (unsigned int)(g >> 32), (unsigned int)pair, st1, k0, k1);
        u[2 * pair] = pem::u53(r.x, r.y);
        u[2 * pair + 1] = pem::u53(r1.z, r1.w);
    }
#pragma unroll
    for (int d = 0; d < 15; ++d) x[d] = transform_call(__builtin_amdgcn_readfirstlane(design_kind[d]), design[d], design[15 + d], u[d]);
    if (mc.x_out) {
#pragma unroll
        for (int d = 0; d < 15; ++d) mc.x_out[(size_t)d * mc.ld + g_local] = x[d];
    }
    SampleIn<true> v;
    v.P_b = x[0]; v.x0 = x[1]; v.x1 = x[2]; v.x2 = x[3]; v.x3 = x[4]; v.x4 = x[5]; v.x5 = x[6]; v.x6 = x[7];
    v.c0 = x[8]; v.c1 = x[9]; v.c2 = x[10]; v.c3 = x[11]; v.c4 = x[12]; v.c5 = x[13]; v.sigma = x[14];
    return v;
}

// ---------------------------------------------------------------------------------------------
// fast path: R = 1.  Each wave is persistent over 64-sample tiles; WPB independent waves per workgroup.
//   L        lanes that share a sample during the rounds (2, 4 or 8)
//   COUPLED  cathode + thruster stages are evaluated in front of the plume (inputs from CoupledIO)
//   JMODE    0: reduced-QoI mode, no profile;  1: stage and store the 91-point profile as fp64;
//            2: mixed mode -- same fp64 arithmetic, profile rounded once to fp32 when it is staged
//            3: fused likelihood -- the profile is staged in LDS only and reduced against measured current
//               densities there (csrc/pem_likelihood.hip's formula); nothing but scalars leaves the chip
//            (the fused compression mode -- latent = norm(j_ion) @ basis -- is a kernel of its own, one lane per sample:
//            csrc/pem_latent.hip)
//            4: as 1, and the staged profile is COUNTED against the brackets of a percentile selection on its way out (count_round);
//            5: the same without the stores -- the percentiles of a profile that is never written
// LDS map (doubles): shared by the workgroup: simpson[96][2] | dpoly[32*12];  per wave: params[NROWS][64] |
// tile[S*91] | 2 (sink).  The Simpson table is padded with zero weights to L*CH <= 96 entries so the angle loop
// needs no branch.  The den/num partial sums of a round reuse the rows of `params` that the round has consumed.
// ---------------------------------------------------------------------------------------------
constexpr int NPARAM = 9;   // X1 X2 jcex | r0 G E (beam 1) | r0 G E (beam 2)
constexpr int NSIMP = 96;   // >= L*CH for L in {2, 4, 8}
constexpr int WPB = 4;      // waves per workgroup (they share the two tables and nothing else)
template <int L>
constexpr int param_rows() { return 2 * L > NPARAM ? 2 * L : NPARAM; }   // rows 2c, 2c+1 are reused for the Simpson partials
constexpr int TABLE_DOUBLES = 2 * NSIMP + PEM_NDI * PEM_NDC;
constexpr int QPOLY_DOUBLES = (PEM_NDI + PEM_NQB) * PEM_NDC * 2;
template <int L, int JMODE>
constexpr int wave_lds_doubles() {
    return param_rows<L>() * WAVE +
           ((JMODE == 1 || JMODE == 3 || JMODE == 4 || JMODE == 5) ? (WAVE / L) * NANG + 2 : JMODE == 2 ? ((WAVE / L) * NANG + 4) / 2 : 0);
}
template <int L, int JMODE>
constexpr int fast_lds_doubles() { return TABLE_DOUBLES + WPB * wave_lds_doubles<L, JMODE>(); }

// Order LDS traffic inside ONE wave (after the table load a wave only ever reads LDS it wrote itself): the LDS
// unit executes a wave's DS instructions in issue order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// 16-byte store of the write-once profile stream.  Non-temporal: measured 227.6 -> 190.7 us per 1.25e6-sample
// launch against plain stores, interleaved A/B (tools/ab_bench.py); the same hint on the small input
// loads or on the per-sample QoI stores is slower and is not used.
#ifndef PEM_NT_STORES
#define PEM_NT_STORES 1
#endif
// tuning knobs of the fused modes (file scope: a #define inside a function body does not survive -save-temps)
#ifndef PEM_LOGLIK_MU
#define PEM_LOGLIK_MU 2       // measurement records in flight per lane in the fused likelihood mode
#endif
__device__ __forceinline__ void stream_store(f64x2 v, f64x2* dst) {
#if PEM_NT_STORES
    __builtin_nontemporal_store(v, dst);
#else
    *dst = v;
#endif
}

// LDS views of one wave
struct WaveLds {
    const double* meas;      // fused likelihood: [n_cond*n_ang] records {weight, y, inv_std, k (integer bits)}, or nullptr
    const double2* simpson;  // [96] {cden, cnum}
    const double* poly;      // [32*12]
    const double2* qpoly;    // reduced-QoI mode: [(32+64)*12] {Qd, Qn} coefficients of the Simpson functionals, or nullptr
    double* params;          // [9][64]
    double* tile;            // [S*91] + 2
    unsigned tile_off;       // its byte offset in the workgroup's dynamic LDS (counting modes)
};

// The slow, literal evaluation of one lane's chunk of angles: direct exp() of -(alpha_k / alpha)^2 per beam, every
// product and sum rounded separately, as plume.py:99-102 (and the oracle) do.  The recurrence of the angle loop is
// accurate to ~1e-14 relative -- also where the reference's own exp() has run into the denormal range (argument
// below -708) or rounded to zero (below -745.13), which is where "accurate" and "equal to the reference" part ways:
// with j_cex = 0 the reference sees j_ion = 0 there and flags the sample invalid (plume.py:105).  A chunk whose
// smallest value is below 1e-290 (or <= 0) is therefore recomputed this way; under the PEM-v0 priors j_cex > 1e-6
// and this never runs.  Out of line: it must not cost the angle loop registers.
struct ChunkSums {
    double den, num, lo;
};
template <typename JT, bool WRITE_J>
__device__ __attribute__((noinline)) ChunkSums exact_chunk(double X1a, double X2a, double jcex, double a1, double a2, int k0,
                                                           int nk, const double2* w, JT* tile_row) {
#pragma clang fp contract(off)
    ChunkSums r{0.0, 0.0, __builtin_inf()};
    for (int j = 0; j < nk; ++j) {
        const int k = k0 + j;
        if (k >= NANG) break;
        const double alpha = k == NANG - 1 ? HALF_PI : (double)k * GRID_H;   // np.linspace(0, pi/2, 91)
        const double t1 = alpha / a1, t2 = alpha / a2;
        const double f = X1a * exp(-(t1 * t1)) + X2a * exp(-(t2 * t2));
        const double ji = f + jcex;
        if (WRITE_J) tile_row[j] = (JT)ji;
        r.lo = fmin(r.lo, WRITE_J ? ji : f);
        r.den = __builtin_fma(w[j].x, f, r.den);
        r.num = __builtin_fma(w[j].y, f, r.num);
    }
    return r;
}

// ---- counting modes (JMODE 4 / 5): the round tile against the brackets of a percentile selection -------------------------------
// The profile's percentiles over the samples (gen_data.py:163-168, monte_carlo.py:363-658) need, per (angle, quantile), the
// number of values below a bracket and the values inside it (csrc/pem_quantile.hip: the pilot form).  A round's S x 91 values sit
// in LDS in final order, so the wave changes roles once more: lane = angle (two slots: angles 0..63 and 64..90), down the S
// samples of the round.  Per (value, bracket) one subtraction of high words -- its borrow is "below", t <= words is "inside"
// (brackets end on whole words and do not overlap: the selection falls back otherwise) -- with the counts in registers for a round
// and in per-workgroup LDS counters between rounds.  The few per cent of values inside a bracket are noted as one bit per (slot,
// sample) and leave as 16-byte records {key, angle * nq + quantile}, appended to the wave's own region of the record buffer
// (no atomics: the count lives in a scalar register) by however many lanes have one left, until none has.
// Its LDS operands are byte offsets into the workgroup's dynamic LDS (as an out-of-line function it must not take generic pointers:
// they would turn its LDS traffic into flat accesses).
// What count_round needs and the rounds do not: kept in the wave's own LDS words (a record of 48 bytes written once per launch),
// so that the call carries four arguments and nothing of this stays live in the caller's registers across the rounds (as
// arguments they cost the counting kernels 8-56 bytes of stack for registers saved around the call).
struct CountCtx {
    pem::Record* rec;            // this wave's region of the record buffer
    uint8_t* row_certain;        // the premask's per-sample counts (whole arrays), or nullptr
    uint8_t* row_uncertain;
    unsigned cap;                // records the region holds
    unsigned tab_off;            // LDS [91][NQ] {loh, words}
    unsigned below_off;          // LDS [NQ][91]
    unsigned pm_off;             // LDS [91] uint4: the premask's thresholds
};
struct QCount {
    unsigned ctx_off;            // LDS: this wave's CountCtx
    unsigned below_off;          // LDS [NQ][91] (the kernel's final flush)
    unsigned cap, cnt;           // region size; records produced so far (wave-uniform)
};
struct NoCount {};

// PM: the outlier test of gen_data.py:163-168 rides along (the "premask").  A sample is an outlier of the profile when more than
// int(0.75 * 91) of its values lie outside [p25 - f iqr, p75 + f iqr] per angle -- bounds that are not known until the selection
// is done.  But p25 and p75 are known to lie in their brackets, so the bounds lie in intervals [lo_min, lo_max], [hi_min, hi_max]
// (rounded as numpy rounds them; every operation is monotone), and a value is either OUTSIDE FOR CERTAIN (below lo_min or above
// hi_max), INSIDE FOR CERTAIN, or uncertain (in one of the two intervals: about one value in a hundred).  Per sample the two
// counts leave as bytes; the caller settles the few samples whose verdict the uncertain values could change.  Four comparisons of
// high words per value, kept as wave masks; a row's counts are population counts of those masks.
// Inlined into the (rolled) rounds.  Its first version, inlined with every loop unrolled, cost the kernel 140 registers and one wave
// per SIMD, so it went out of line -- where the premask variant saved two callee-saved registers on the stack per call (8 bytes
// of scratch).  With the sample loops rolled in groups of four the inlined form fits two waves per SIMD (209-228 registers, no
// scratch) and measures the same (5.13-5.21 against 5.18-5.22 ms per 1e7-sample campaign): -DPEM_COUNT_INLINE=0 is the other form.
#ifndef PEM_COUNT_INLINE
#define PEM_COUNT_INLINE 1
#endif
#if PEM_COUNT_INLINE
#define PEM_COUNT_LINKAGE __forceinline__
#else
#define PEM_COUNT_LINKAGE __attribute__((noinline))
#endif
template <int NQ, int S, bool FULL, bool PM>
__device__ PEM_COUNT_LINKAGE unsigned count_round(unsigned ctx_off, unsigned tile_off, unsigned cnt, int lane, int rows, long long first) {
#if defined(PEM_COUNT_EXP) && PEM_COUNT_EXP == 1
    return cnt;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const CountCtx ctx = *reinterpret_cast<const CountCtx*>(smem_raw + ctx_off);
    const uint2* tab = reinterpret_cast<const uint2*>(smem_raw + ctx.tab_off);
    unsigned* below = reinterpret_cast<unsigned*>(smem_raw + ctx.below_off);
    const double* tile = reinterpret_cast<const double*>(smem_raw + tile_off);
    const uint4* pm = reinterpret_cast<const uint4*>(smem_raw + ctx.pm_off);   // [91] {certainly below, possibly below, possibly above, certainly above}
    pem::Record* rec = ctx.rec;
    const unsigned cap = ctx.cap;
    uint8_t* row_certain = PM ? ctx.row_certain + first : nullptr;
    uint8_t* row_uncertain = PM ? ctx.row_uncertain + first : nullptr;
    unsigned bits = 0;           // bit 16 slot + s: value s of the slot's angle lies inside a bracket
    unsigned row_c = 0, row_u = 0;   // lane s: values of row s outside for certain / uncertain
    static_assert(S <= 16, "one bit per sample and slot");
    // Two slots: angles 0..63, one row per wave instruction (lane = angle), and angles 64..90, TWO rows per instruction in full
    // tiles (lanes 0..26 and 32..58: 24 instructions' worth of values per round instead of 32).
    // (Rolled loops on purpose: fully unrolled, the scheduler hoisted every read of the round and the function took 248 registers,
    // or turned every borrow into a 0 / 1 register.  Comparisons are kept as wave masks -- v_cmp into a scalar pair, s_or, and back
    // as a lane condition: written with plain bools every one of them was materialised, 37 instructions per value instead of 19.)
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int half = (FULL && slot == 1) ? lane >> 5 : 0;      // which of the instruction's two rows this lane looks at
        const int sub = slot == 1 ? (lane & 31) : lane;
        const int col = 64 * slot + sub;
        const bool on = slot == 0 || (sub < NANG - 64 && (FULL || lane < 32));
        const int cc = on ? col : NANG - 1;           // idle lanes of the second slot repeat the last angle and keep nothing
        uint2 tb[NQ];
        unsigned nb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            tb[q] = tab[cc * NQ + q];
            nb[q] = 0;
        }
        const int* hw = reinterpret_cast<const int*>(tile) + 2 * cc + 1 + half * 2 * NANG;   // the high words of this lane's values
        uint4 th = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (PM) th = pm[cc];
        constexpr unsigned long long PART = (1ull << (NANG - 64)) - 1;
        const unsigned long long lanes_on = slot == 0 ? ~0ull : (FULL ? (PART | (PART << 32)) : PART);
        unsigned mine = 0;
        // one value per lane: the brackets' below-counts and "inside any" (then `bit` is noted); the premask's two wave masks
        auto one = [&](int row, unsigned bit, unsigned long long& certain, unsigned long long& maybe) {
            const unsigned kh = pem::order_key_high(hw[row * 2 * NANG]);
            unsigned long long in = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                unsigned t;
                nb[q] += __builtin_sub_overflow(kh, tb[q].x, &t) ? 1u : 0u;
                in |= __builtin_amdgcn_uicmp(t, tb[q].y, 37 /* ICMP_ULE */);   // (a borrow leaves t above every `words`)
            }
            mine |= __builtin_amdgcn_inverse_ballot_w64(in) ? bit : 0u;
            if constexpr (PM) {
                certain = (__builtin_amdgcn_uicmp(kh, th.x, 36 /* ULT */) | __builtin_amdgcn_uicmp(kh, th.w, 34 /* UGT */)) & lanes_on;
                maybe = (__builtin_amdgcn_uicmp(kh, th.y, 37 /* ULE */) | __builtin_amdgcn_uicmp(kh, th.z, 35 /* UGE */)) & lanes_on & ~certain;
            }
        };
        if constexpr (FULL) {
            // four rows a turn (explicit groups: the wave-level builtins keep the compiler from unrolling a counted loop with a
            // remainder).  The premask's counts of the four rows -- scalars, at most 64 + 27 each over the two slots -- are packed
            // into one word and kept by lane `turn`: two vector instructions per four rows instead of six per row.
            static_assert(S % 4 == 0, "rows in groups of four");
            const unsigned bit0 = 1u << (16 * slot + half);
#pragma unroll 1
            for (int turn = 0; turn < S / 4; ++turn) {
                unsigned pc = 0, pu = 0;
                if (slot == 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        unsigned long long certain = 0, maybe = 0;
                        one(4 * turn + u, bit0 << (4 * turn + u), certain, maybe);
                        if constexpr (PM) {
                            pc |= (unsigned)__popcll(certain) << (8 * u);
                            pu |= (unsigned)__popcll(maybe) << (8 * u);
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {              // rows 4 turn + 2 u (lanes 0..31) and 4 turn + 2 u + 1 (lanes 32..63)
                        unsigned long long certain = 0, maybe = 0;
                        one(4 * turn + 2 * u, bit0 << (4 * turn + 2 * u), certain, maybe);
                        if constexpr (PM) {
                            pc |= ((unsigned)__popc((unsigned)certain) << (16 * u)) | ((unsigned)__popc((unsigned)(certain >> 32)) << (16 * u + 8));
                            pu |= ((unsigned)__popc((unsigned)maybe) << (16 * u)) | ((unsigned)__popc((unsigned)(maybe >> 32)) << (16 * u + 8));
                        }
                    }
                }
                if constexpr (PM) {
                    row_c += lane == turn ? pc : 0u;
                    row_u += lane == turn ? pu : 0u;
                }
            }
        } else {
            for (int r = 0; r < rows; ++r) {          // the ragged last tile of a launch: row by row, one byte per lane
                unsigned long long certain = 0, maybe = 0;
                one(r, 1u << (16 * slot + r), certain, maybe);
                if constexpr (PM) {
                    row_c += lane == r ? (unsigned)__popcll(certain) : 0u;
                    row_u += lane == r ? (unsigned)__popcll(maybe) : 0u;
                }
            }
        }
        bits |= on ? mine : 0u;
        if (on) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (nb[q]) atomicAdd(&below[q * NANG + col], nb[q]);
        }
    }
    if constexpr (PM) {
        if constexpr (FULL) {                          // lane `turn` holds the bytes of rows 4 turn .. 4 turn + 3
            if (lane < S / 4) {
                reinterpret_cast<unsigned*>(row_certain)[lane] = row_c;
                reinterpret_cast<unsigned*>(row_uncertain)[lane] = row_u;
            }
        } else if (lane < rows) {
            row_certain[lane] = (uint8_t)row_c;
            row_uncertain[lane] = (uint8_t)row_u;
        }
    }
#if defined(PEM_COUNT_EXP) && PEM_COUNT_EXP == 2
    bits = 0;
#endif
    // The records: {key, angle} (which of the angle's brackets holds the key is found again by the passes over the records: they
    // see 4 % of the values).  Two per lane and turn -- the reads of the tile, and the stores, of both are in flight together --
    // appended behind the wave's count by however many lanes have one.
    while (__ballot(bits != 0)) {
        const bool has0 = bits != 0;
        const int b0 = has0 ? __builtin_ctz(bits) : 0;
        bits &= bits - 1;                              // (0 stays 0)
        const bool has1 = bits != 0;
        const int b1 = has1 ? __builtin_ctz(bits) : 0;
        bits &= bits - 1;
        const int col0 = (b0 >> 4) ? 64 + (lane & 31) : lane, col1 = (b1 >> 4) ? 64 + (lane & 31) : lane;
        const double x0 = tile[(b0 & 15) * NANG + (has0 ? col0 : 0)], x1 = tile[(b1 & 15) * NANG + (has1 ? col1 : 0)];
        const unsigned long long e0 = __ballot(has0), e1 = __ballot(has1);
        const unsigned at0 = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(e0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)e0, 0u));
        const unsigned n0 = (unsigned)__popcll(e0);
        const unsigned at1 = cnt + n0 + __builtin_amdgcn_mbcnt_hi((unsigned)(e1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)e1, 0u));
        if (has0 && at0 < cap) {
            f64x2 r;
            r.x = __longlong_as_double((long long)pem::order_key(x0));
            r.y = __longlong_as_double((long long)col0);
            *reinterpret_cast<f64x2*>(rec + at0) = r;
        }
        if (has1 && at1 < cap) {
            f64x2 r;
            r.x = __longlong_as_double((long long)pem::order_key(x1));
            r.y = __longlong_as_double((long long)col1);
            *reinterpret_cast<f64x2*>(rec + at1) = r;
        }
        cnt += n0 + (unsigned)__popcll(e1);
    }
    return cnt;
}

// One 64-sample tile.  FULL = every sample of the tile exists (the steady state of the persistent loop:
// no bounds checks and a fixed number of stores, so the compiler can count them); FULL = false is the
// ragged last tile of a batch.
template <int L, bool COUPLED, int JMODE, bool FULL, int NQ = 0, bool PM = false, class QC = NoCount>
__device__ __forceinline__ void process_tile(const PlumeIO& io, const CoupledIO& cio, const WaveLds& m,
                                             const SampleIn<COUPLED>& in, long long t, int lane, double rad,
                                             double inv_r2, double inv_2pi_r2, QC& qc) {
    constexpr int S = WAVE / L;             // samples per round
    constexpr int CH = (NANG + L - 1) / L;  // angles per lane
    constexpr int TILE = S * NANG;          // profile values per round tile
    constexpr bool WRITE_J = JMODE != 0;
    using JT = typename std::conditional<JMODE == 2, float, double>::type;   // element type of the stored profile
    constexpr int PER16 = 16 / (int)sizeof(JT);                              // values per 16-byte piece
    constexpr int PAIRS = TILE / PER16;     // 16-byte pieces of a full round tile (TILE divides evenly)
    static_assert(TILE % PER16 == 0, "a round tile is a whole number of 16-byte pieces");
    const int s = lane % S, c = lane / S;   // role during the rounds
    const int k0 = c * CH;
    const double2* my_w = m.simpson + k0;   // this lane's folded Simpson weights
    double* params = m.params;
    JT* tile = reinterpret_cast<JT*>(m.tile);
    const long long g = t * WAVE + lane;
    const bool live = FULL || g < io.n;

    // ------------------------------ PRELUDE: one lane per sample ------------------------------
    double I_B0, thrust = 0.0, V_cc = 0.0;
    bool have_T;
    if constexpr (COUPLED) {
        V_cc = cathode_vcc(in.P_b, in.x0, in.x1, in.x2, in.x3, in.x4, io.torr2pa);
        const ThrusterQoI th = thruster_stage(in.x0, V_cc, in.x5, in.x6);
        I_B0 = th.I_B0;
        thrust = th.T;
        have_T = true;
    } else {
        I_B0 = in.x0;
        thrust = in.x1;
        have_T = io.T != nullptr;
    }
    // plume.py:40-61
    const PlumeSetup ps = plume_setup(in.P_b, in.c1, in.c2, in.c3, in.c4, in.c5, io.torr2pa);
    const double n_neutral = ps.n_neutral, a1 = ps.a1, a2 = ps.a2;
    const double u1 = 1.0 / (a1 * a1), u2 = 1.0 / (a2 * a2);
    const double A1 = (1.0 - in.c0) / normaliser(a1, u1, m.poly);  // plume.py:64-73
    const double A2 = in.c0 / normaliser(a2, u2, m.poly);          // plume.py:75-85
    // plume.py:95-100 at the single radius
    const double decay = exp(-rad * n_neutral * in.sigma);
    const double j_cex = I_B0 * (1.0 - decay) * inv_2pi_r2;
    const double base = I_B0 * decay * inv_r2;
    const unsigned long long a1_nonpos = __ballot(a1 <= 0.0);  // plume.py:105, first term
    unsigned long long inv_mask = 0;
    double den = 0.0, num = 0.0;
    // Reduced-QoI mode: nothing needs the 91 profile values themselves.  The two Simpson sums are linear in the beam
    // amplitudes and depend on each beam only through its width -> two table look-ups per beam (simpson_functionals).
    // What the loop would still decide is plume.py:105's `any(j_ion <= 0)`; with both amplitudes >= 0 and j_cex > 0 every
    // j_ion[k] >= j_cex > 0, so the flag is `alpha1 <= 0` alone.  Such a "plain" sample takes its sums from the tables,
    // any other one (NaN or denormal amplitudes, negative c0 or 1 - c0, beams narrower than QA_MIN) from the loop: a sample's result
    // depends on that sample alone, however batches and shards cut the design.  The loop is skipped -- a wave-uniform
    // branch -- when all 64 samples of the tile are plain, which under the PEM-v0 priors is every tile.
    bool plain = false, table_tile = false;
    if constexpr (JMODE == 0) {
        // (amplitudes in the denormal range are left to the loop as well: there every w_k f_k underflows to zero and the
        // reference's 0/0 = NaN must come out, where the tabulated sum would still see a few bits)
        const double X1a = base * A1, X2a = base * A2;
        plain = fabs(a1) >= PEM_QA_MIN && fabs(a2) >= PEM_QA_MIN && X1a >= 0.0 && X2a >= 0.0 && j_cex > 0.0 &&
                (fmax(X1a, X2a) >= 1e-280 || (X1a == 0.0 && X2a == 0.0));
        table_tile = __all(plain);
        if (table_tile) inv_mask = a1_nonpos;
    }
    if (!table_tile) {
        // Gaussian recurrences: e_k = exp(-k^2 s), s = (h/a)^2; chunk starts at k = c*CH
        const double s1 = (GRID_H * GRID_H) * u1, s2 = (GRID_H * GRID_H) * u2;
        // a1 == 0: exp(-(0/0)^2) is NaN in the reference; the amplitudes carry it (A1 is NaN there)
        params[0 * WAVE + lane] = base * A1;
        params[1 * WAVE + lane] = base * A2;
        params[2 * WAVE + lane] = j_cex;
        params[3 * WAVE + lane] = exp_nonpos(-s1);
        params[4 * WAVE + lane] = exp_nonpos(-(2.0 * CH) * s1);
        params[5 * WAVE + lane] = exp_nonpos(-(double)(CH * CH) * s1);
        params[6 * WAVE + lane] = exp_nonpos(-s2);
        params[7 * WAVE + lane] = exp_nonpos(-(2.0 * CH) * s2);
        params[8 * WAVE + lane] = exp_nonpos(-(double)(CH * CH) * s2);
        wave_lds_sync();

        // ------------------------------ ROUNDS: L lanes per sample ------------------------------
        // (the counting modes keep the rounds rolled: unrolled four times with count_round inside, the kernel took 340 registers
        // and one wave per SIMD)
        // (the reduced-QoI mode takes this loop for the tiles that hold a sample the tables do not cover -- none under the priors:
        // rolled, with a rolled angle loop, it fits the three waves per SIMD that mode is compiled for; unrolled it spilled there)
#if defined(PEM_COUNT_EXP) && PEM_COUNT_EXP == 3
        constexpr int ROUND_UNROLL = L;
#else
        constexpr int ROUND_UNROLL = (NQ > 0 || JMODE == 0) ? 1 : L;
#endif
        constexpr int ANGLE_UNROLL = JMODE == 0 ? 1 : CH;
#pragma unroll ROUND_UNROLL
        for (int round = 0; round < L; ++round) {
            const int smp = round * S + s;
            double X1 = params[0 * WAVE + smp], X2 = params[1 * WAVE + smp];
            const double jcex = params[2 * WAVE + smp];
            const double r01 = params[3 * WAVE + smp], G1 = params[4 * WAVE + smp], E1 = params[5 * WAVE + smp];
            const double r02 = params[6 * WAVE + smp], G2 = params[7 * WAVE + smp], E2 = params[8 * WAVE + smp];
            // coarse recurrence to this lane's first angle k0 = c*CH:
            //   e_{k0} = E^(c^2), r_{k0} = exp(-(2 k0 + 1) s) = r0 * G^c
            double rr1 = r01, rr2 = r02, rho1 = E1, rho2 = E2;
            const double E1sq = E1 * E1, E2sq = E2 * E2;
#pragma unroll
            for (int i = 0; i < L - 1; ++i) {
                if (i < c) {
                    X1 *= rho1;
                    rho1 *= E1sq;
                    rr1 *= G1;
                    X2 *= rho2;
                    rho2 *= E2sq;
                    rr2 *= G2;
                }
            }
            const double q1 = r01 * r01, q2 = r02 * r02;
            double den = 0.0, num = 0.0, lo = __builtin_inf();   // this lane's chunk of the round (shadows the tile sums)
            // The weight reads are issued PF iterations ahead IN SOURCE ORDER: the tile stores in between are
            // LDS stores the compiler must assume may alias the table, so it cannot hoist the reads itself.
            constexpr int PF = 6;
            double2 wq[CH];
#pragma unroll
            for (int j = 0; WRITE_J && j < PF && j < CH; ++j) wq[j] = my_w[j];
#pragma unroll ANGLE_UNROLL
            for (int j = 0; j < CH; ++j) {
                double2 wj;
                if constexpr (!WRITE_J) {
                    wj = my_w[j];                          // no tile stores in between: the compiler schedules the reads
                } else {
                    if (j + PF < CH) wq[j + PF] = my_w[j + PF];
                    wj = wq[j];
                }
                const double f = X1 + X2;     // j_beam + j_scat
                const double ji = f + jcex;   // plume.py:102
                if ((L - 1) * CH + j < NANG) {  // an angle every chunk has (compile-time after unrolling)
                    if constexpr (WRITE_J) tile[s * NANG + k0 + j] = (JT)ji;
                    lo = fmin(lo, WRITE_J ? ji : f);
                } else {                        // past 90 degrees in the last chunk: store to the sink, skip the min
                    const bool in_range = k0 + j < NANG;
                    if constexpr (WRITE_J) tile[in_range ? s * NANG + k0 + j : TILE] = (JT)ji;
                    lo = fmin(lo, in_range ? (WRITE_J ? ji : f) : __builtin_inf());
                }
                den = fma(wj.x, f, den);
                num = fma(wj.y, f, num);
                X1 *= rr1;
                rr1 *= q1;
                X2 *= rr2;
                rr2 *= q2;
            }
            {
                // deep-underflow or non-positive chunk: the literal evaluation decides (see exact_chunk); rare, and the
                // branch is wave-uniform so that the shuffles inside are executed by every lane
                // (an infinite amplitude -- exp(+x) overflow for a negative density -- must turn into NaN where the
                // reference's exp() is exactly zero.  A class test, not `X - X != 0`: hipcc contracts that with the
                // multiply before it into fma(X', rr, -X), the rounding error of the product, which is never zero.)
                const bool uncertain = (WRITE_J ? lo : lo + jcex) < 1e-290 || !__builtin_isfinite(X1) || !__builtin_isfinite(X2);
                if (__ballot(uncertain)) {
                    const double a1s = __shfl(a1, smp), a2s = __shfl(a2, smp);
                    if (uncertain) {
                        const double X1a = params[0 * WAVE + smp], X2a = params[1 * WAVE + smp];
                        const ChunkSums ex = exact_chunk<JT, WRITE_J>(X1a, X2a, jcex, a1s, a2s, k0, CH, my_w, tile + s * NANG + k0);
                        den = ex.den;
                        num = ex.num;
                        lo = ex.lo;
                    }
                }
            }
            // this round has read its nine parameter rows of sample `smp`: rows 2c, 2c+1 now carry the partial sums
            params[(2 * c) * WAVE + smp] = den;
            params[(2 * c + 1) * WAVE + smp] = num;
            // plume.py:105: invalid if alpha1 <= 0 or any j_ion <= 0 (NaN compares false).  Without a stored profile
            // the minimum runs over f and j_cex is added once: rounding is monotonic, min_k fl(f_k + c) = fl(min_k f_k + c).
            unsigned long long bad = __ballot((WRITE_J ? lo : lo + jcex) <= 0.0);
#pragma unroll
            for (int sh = S; sh < WAVE; sh <<= 1) bad |= bad >> sh;   // fold the L chunk lanes of a sample onto bit s
            bad = (bad | (a1_nonpos >> (round * S))) & ((S == 64) ? ~0ull : ((1ull << S) - 1));
            inv_mask |= bad << (round * S);
            if constexpr (WRITE_J) {
                if ((bad >> s) & 1) {  // plume.py:106: the whole profile of an invalid sample becomes 1e-20 (rare)
                    const JT fill = (JT)1e-20;
                    for (int j = 0; j < CH; ++j)
                        if (k0 + j < NANG) tile[s * NANG + k0 + j] = fill;
                }
                wave_lds_sync();
                const long long first = t * WAVE + (long long)round * S;
                if constexpr (JMODE == 3) {
                    static_assert(JMODE != 3 || 2 * L < param_rows<L>(), "row 2L of `params` carries the likelihood sum");
                    // measured current densities against the staged profile: lane (s, c) takes measurements c, c+L, ...
                    // of its sample's condition (sample index mod n_cond); the sample's sum goes to row 2L of `params`
                    const unsigned cond = ((unsigned)((t * WAVE) % io.n_cond) + (unsigned)(round * S + s)) % (unsigned)io.n_cond;
                    const double4* mt = reinterpret_cast<const double4*>(m.meas) + cond * (io.n_ang | 1);   // {weight, y, 1/std, k}
                    const JT* row = tile + s * NANG;
                    double acc = 0.0;
                    constexpr int MU = PEM_LOGLIK_MU;   // records in flight per lane: the k -> row[k] chain is two LDS latencies deep
                    for (int a0 = c; a0 < io.n_ang; a0 += MU * L) {
                        double4 e[MU];
                        double lo_v[MU], hi_v[MU];
#pragma unroll
                        for (int u = 0; u < MU; ++u) e[u] = mt[a0 + u * L < io.n_ang ? a0 + u * L : a0];
#pragma unroll
                        for (int u = 0; u < MU; ++u) {
                            const int k = __double_as_longlong(e[u].w) & 0x7f;
                            lo_v[u] = row[k];
                            hi_v[u] = row[k + 1];
                        }
#pragma unroll
                        for (int u = 0; u < MU; ++u) {
                            const double model = fma(e[u].x, hi_v[u] - lo_v[u], lo_v[u]);
                            const double z = (e[u].y - model) * e[u].z;
                            if (a0 + u * L < io.n_ang) acc = fma(-0.5 * z, z, acc);
                        }
                    }
#pragma unroll
                    for (int sh = S; sh < WAVE; sh <<= 1) acc += __shfl_xor(acc, sh);   // the L chunk lanes of sample s
                    if (c == 0) params[(2 * L) * WAVE + smp] = acc;
                    wave_lds_sync();
                    continue;
                }
                if constexpr (JMODE == 5) {             // counted, never stored
                    const long long left = io.n - first;
                    qc.cnt = count_round<NQ, S, FULL, PM>(qc.ctx_off, m.tile_off, qc.cnt, lane,
                                                          FULL ? S : (int)(left < S ? (left < 0 ? 0 : left) : S), first);
                    wave_lds_sync();
                    continue;
                }
                // the round's S*91 values are one contiguous, 16-byte aligned block of j_ion
                JT* jbase;
                if constexpr (JMODE == 2) jbase = io.j_ion_f32; else jbase = io.j_ion;
                if constexpr (FULL) {
                    f64x2* dst2 = reinterpret_cast<f64x2*>(jbase + first * NANG);
                    const f64x2* srcv = reinterpret_cast<const f64x2*>(tile);
#pragma unroll
                    for (int it = 0; it < PAIRS / WAVE; ++it) stream_store(srcv[it * WAVE + lane], &dst2[it * WAVE + lane]);
                    if (PAIRS % WAVE != 0 && lane < PAIRS % WAVE)
                        stream_store(srcv[(PAIRS / WAVE) * WAVE + lane], &dst2[(PAIRS / WAVE) * WAVE + lane]);
                } else {
                    long long valid = (io.n - first) * NANG;   // values of this round that exist
                    if (valid > TILE) valid = TILE;
                    if (valid > 0) {
                        JT* dst = jbase + first * NANG;
                        f64x2* dst2 = reinterpret_cast<f64x2*>(dst);
                        const f64x2* srcv = reinterpret_cast<const f64x2*>(tile);
                        const int pieces = (int)(valid / PER16);
                        for (int i = lane; i < pieces; i += WAVE) dst2[i] = srcv[i];
                        const int rest = (int)(valid - (long long)pieces * PER16);
                        if (lane < rest) dst[pieces * PER16 + lane] = tile[pieces * PER16 + lane];
                    }
                }
                if constexpr (JMODE == 4) {             // the stores are on their way (their LDS reads are done): count the tile
                    const long long left = io.n - first;
                    qc.cnt = count_round<NQ, S, FULL, PM>(qc.ctx_off, m.tile_off, qc.cnt, lane,
                                                          FULL ? S : (int)(left < S ? (left < 0 ? 0 : left) : S), first);
                }
                wave_lds_sync();
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < L; ++i) {
            den += params[(2 * i) * WAVE + lane];
            num += params[(2 * i + 1) * WAVE + lane];
        }
    }   // !table_tile
    if constexpr (JMODE == 0) {   // after the loop, so that nothing of it stays live across the loop's 213 registers
        double d1, n1, d2, n2;
        simpson_functionals(m.qpoly, fabs(a1), u1, d1, n1);
        simpson_functionals(m.qpoly, fabs(a2), u2, d2, n2);
        den = plain ? fma(base * A1, d1, (base * A2) * d2) : den;
        num = plain ? fma(base * A1, n1, (base * A2) * n2) : num;
    }

    // ------------------------------ EPILOGUE: one lane per sample ------------------------------
    double cos_div = num / den;  // plume.py:124-127
    if (cos_div == __builtin_inf()) cos_div = __builtin_nan("");
    if constexpr (JMODE == 4 || JMODE == 5) {
        // A NaN in a column makes its percentiles NaN, and the counting above never looks for one: any non-finite f_k leaves the
        // Simpson sum non-finite (w_k f_k is NaN or infinite, and stays), a non-finite j_cex is seen directly -- such a sample
        // (never under the priors) is reported and the selection falls back to the passes that examine every value.
        if (live && (!__builtin_isfinite(den) || !__builtin_isfinite(j_cex))) atomicOr(io.q.flags + 1, 1);
    }
    if constexpr (JMODE == 3) {
        if (live) io.loglik[g] = params[(2 * L) * WAVE + lane];
    }
    if (live) {
        stream_store1(acos(cos_div), io.div + g);
        if (have_T) stream_store1(thrust * cos_div, io.Tc + g);
        if (io.invalid) io.invalid[g] = (uint8_t)((inv_mask >> lane) & 1);
        if constexpr (COUPLED) {
            stream_store1(V_cc, cio.V_cc + g);
            if (cio.I_B0) stream_store1(I_B0, cio.I_B0 + g);
            if (cio.T) stream_store1(thrust, cio.T + g);
        }
    }
    wave_lds_sync();  // params are rewritten by the next tile
}

// Minimum waves per SIMD the register allocator must leave room for: 1 (the default for a 4-wave workgroup; every
// instantiation ends up at two anyway) except the fused Monte-Carlo reduced-QoI one, which is bound by Philox's
// quarter-rate 32x32 multiplies and gains a third wave (99 -> 86 us per 1.25e6 samples); unconstrained it takes 171
// registers, three more than three waves allow.
template <int JMODE, bool MC>
constexpr int min_waves_per_simd() { return (MC && JMODE == 0) ? 3 : 1; }

// Only the fused Monte-Carlo instantiations carry the ~340-byte design in their kernel arguments.
struct NoDesign {};
template <bool MC>
using DesignArg = typename std::conditional<MC, McDesign, NoDesign>::type;

// bytes of LDS the counting modes add per workgroup: brackets' {loh, words} [91][NQ] | below counters [NQ][91] | premask thresholds [91] x 16
template <int NQ, bool PM>
constexpr int count_lds_bytes() { return NANG * NQ * 8 + NQ * NANG * 4 + 8 + (PM ? NANG * 16 + 8 : 0) + WPB * (int)sizeof(CountCtx); }

template <int L, bool COUPLED, int JMODE, bool MC = false, int NQ = 0, bool PM = false>
__global__ __launch_bounds__(WAVE * WPB) __attribute__((amdgpu_waves_per_eu(min_waves_per_simd<JMODE, MC>())))
void plume_r1_kernel(PlumeIO io, CoupledIO cio, long long ntiles, DesignArg<MC> mc) {
    static_assert(!MC || COUPLED, "the fused Monte-Carlo mode generates the coupled inputs");
    static_assert((JMODE == 4 || JMODE == 5) == (NQ > 0), "the counting modes, and only they, know their number of brackets");
    static_assert(NQ == 0 || L == 4, "count_round keeps one bit per sample of a 16-sample round");
    static_assert(L == 2 || L == 4 || L == 8, "lanes per sample");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    double2* tab_simpson = reinterpret_cast<double2*>(lds);          // [96] {cden, cnum}, zero past angle 90
    double* tab_poly = lds + 2 * NSIMP;                               // [32*12]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    WaveLds m;
    m.simpson = tab_simpson;
    m.poly = tab_poly;
    m.params = lds + TABLE_DOUBLES + wave * wave_lds_doubles<L, JMODE>();   // [rows][64], private to this wave
    m.tile = m.params + param_rows<L>() * WAVE;                               // [S*91] + sink
    m.tile_off = (unsigned)(reinterpret_cast<unsigned char*>(m.tile) - smem_raw);
    m.qpoly = nullptr;
    if constexpr (JMODE == 0) {   // Simpson-functional tables behind the per-wave regions
        double* q = lds + TABLE_DOUBLES + WPB * wave_lds_doubles<L, JMODE>();
        for (int i = tid; i < QPOLY_DOUBLES; i += WAVE * WPB) q[i] = PEM_QPOLY[i];
        m.qpoly = reinterpret_cast<const double2*>(q);
    }
    double* design = nullptr;
    if constexpr (MC) {   // the Monte-Carlo design behind everything else
        static_assert(!MC || JMODE == 0 || JMODE == 1 || JMODE == 4 || JMODE == 5, "the fused Monte-Carlo mode writes an fp64 profile or none");
        design = lds + TABLE_DOUBLES + WPB * wave_lds_doubles<L, JMODE>() + (JMODE == 0 ? QPOLY_DOUBLES : 0);
        if (tid < 15) {
            design[tid] = mc.a[tid];
            design[15 + tid] = mc.b[tid];
            reinterpret_cast<int*>(design + 32)[tid] = mc.kind[tid];
        }
    }
    m.meas = nullptr;
    if constexpr (JMODE == 3) {   // measurement tables behind the per-wave regions (vmcnt is in order: never global)
        double* meas = lds + TABLE_DOUBLES + WPB * wave_lds_doubles<L, JMODE>();
        const int nent = io.n_cond * io.n_ang;
        // one 32-byte record per measurement; an odd record stride per condition spreads the conditions over the banks
        for (int i = tid; i < nent; i += WAVE * WPB) {
            const int r = 4 * ((i / io.n_ang) * (io.n_ang | 1) + i % io.n_ang);
            meas[r] = io.m_wgt[i];
            meas[r + 1] = io.m_y[i];
            meas[r + 2] = io.m_inv_std[i];
            meas[r + 3] = __longlong_as_double((long long)io.m_kidx[i]);
        }
        m.meas = meas;
    }

    using QC = typename std::conditional<(NQ > 0), QCount, NoCount>::type;
    QC qc;
    if constexpr (NQ > 0) {   // the brackets and the below counters behind everything else
        double* qbase = lds + TABLE_DOUBLES + WPB * wave_lds_doubles<L, JMODE>() + (MC ? MC_LDS_DOUBLES : 0);
        uint2* qtab = reinterpret_cast<uint2*>(qbase);
        unsigned* qbelow = reinterpret_cast<unsigned*>(qtab + NANG * NQ);
        for (int i = tid; i < NANG * NQ; i += WAVE * WPB) {
            // (a selection of fewer than NQ quantiles leaves the other brackets empty: loh = 2^32 - 1, no words)
            const int c = i / NQ, q = i - c * NQ;
            qtab[i] = q < io.q.nq ? make_uint2(io.q.br[c * io.q.nq + q].loh, io.q.br[c * io.q.nq + q].words) : make_uint2(0xffffffffu, 0u);
            qbelow[i] = 0;
        }
        const unsigned gw = blockIdx.x * WPB + wave;
        unsigned char* after = reinterpret_cast<unsigned char*>(qbelow + NQ * NANG);
        after = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(after) + 15) & ~uintptr_t(15));
        unsigned pm_off = 0;
        if constexpr (PM) {
            uint4* pmt = reinterpret_cast<uint4*>(after);
            for (int i = tid; i < NANG; i += WAVE * WPB) pmt[i] = io.q.premask[i];
            pm_off = (unsigned)(after - smem_raw);
            after += NANG * sizeof(uint4);
        }
        CountCtx* ctx = reinterpret_cast<CountCtx*>(after) + wave;
        if (lane == 0) {
            ctx->rec = io.q.rec + (size_t)gw * io.q.cap;
            ctx->row_certain = io.q.row_certain;
            ctx->row_uncertain = io.q.row_uncertain;
            ctx->cap = io.q.cap;
            ctx->tab_off = (unsigned)(reinterpret_cast<unsigned char*>(qtab) - smem_raw);
            ctx->below_off = (unsigned)(reinterpret_cast<unsigned char*>(qbelow) - smem_raw);
            ctx->pm_off = pm_off;
        }
        qc.ctx_off = (unsigned)(reinterpret_cast<unsigned char*>(ctx) - smem_raw);
        qc.below_off = (unsigned)(reinterpret_cast<unsigned char*>(qbelow) - smem_raw);
        qc.cap = io.q.cap;
        qc.cnt = 0;
    }
    for (int i = tid; i < NSIMP; i += WAVE * WPB)
        tab_simpson[i] = i < NANG ? make_double2(PEM_SIMPSON_CDEN[i], PEM_SIMPSON_CNUM[i]) : make_double2(0.0, 0.0);
    for (int i = tid; i < PEM_NDI * PEM_NDC; i += WAVE * WPB) tab_poly[i] = PEM_DPOLY[i];
    __syncthreads();   // the only workgroup barrier of the evaluation: from here on the waves are independent

    const double rad = io.radius;
    const double inv_r2 = 1.0 / (rad * rad);
    const double inv_2pi_r2 = 1.0 / (2.0 * PEM_PI * (rad * rad));

    // persistent loop over the tiles whose 64 samples all exist; inputs are prefetched one tile ahead
    const long long nfull = io.n / WAVE;
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one), so with tile = workgroup
    // index every XCD writes every eighth 186-KB piece of j_ion; with the bijective remap below (cdna_hip_programming.md, "XCD
    // swizzle must be bijective") the workgroups that share an XCD take one CONTIGUOUS eighth of the tiles, and each XCD's L2 hands
    // the memory system one sequential stream instead of a comb.  There is no inter-workgroup reuse here -- the gain is in how the
    // writes arrive at HBM: 210 -> 204 us per 1.25e6-sample launch, 204 -> 195 us per step on two streams (interleaved A/B,
    // the same on four leases; a first box showed 202 -> 184: profiles/grid_modes_r03.txt).  The remap itself: pem_common.h.
    // Where it pays: the modes that write a profile (without one -- reduced QoIs, VALU-bound -- it measured 2 % slower: 48.4 against
    // 47.2 us), and one-shot grids only -- a persistent grid of a few rounds (312 512-sample pieces) ran at 249 against 215 us per
    // 1.25e6 samples with it.
    const bool one_tile_per_wave = (long long)gridDim.x * WPB >= ntiles;
    const long long vblock = ((JMODE == 1 || JMODE == 2) && one_tile_per_wave) ? (long long)pem::xcd_contiguous_block() : (long long)blockIdx.x;
    const long long me = vblock * WPB + wave, nwaves = (long long)gridDim.x * WPB;
    long long t = me;
    if constexpr (MC) {
        for (; t < nfull; t += nwaves) {
            const SampleIn<COUPLED> in = generate_sample(mc, design, t * WAVE + lane);
            process_tile<L, COUPLED, JMODE, true, NQ, PM>(io, cio, m, in, t, lane, rad, inv_r2, inv_2pi_r2, qc);
        }
        if (nfull < ntiles && (nfull % nwaves) == me) {
            const long long g = nfull * WAVE + lane;
            McDesign quiet = mc;
            if (g >= io.n) quiet.x_out = nullptr;       // dead lanes recompute the last sample and store nothing
            const SampleIn<COUPLED> in = generate_sample(quiet, design, g < io.n ? g : io.n - 1);
            process_tile<L, COUPLED, JMODE, false, NQ, PM>(io, cio, m, in, nfull, lane, rad, inv_r2, inv_2pi_r2, qc);
        }
    } else {
        if (t < nfull) {
            SampleIn<COUPLED> nxt = load_sample<COUPLED>(io, cio, t * WAVE + lane);
            for (; t < nfull; t += nwaves) {
                const SampleIn<COUPLED> in = nxt;
                if (t + nwaves < nfull) nxt = load_sample<COUPLED>(io, cio, (t + nwaves) * WAVE + lane);
                process_tile<L, COUPLED, JMODE, true, NQ, PM>(io, cio, m, in, t, lane, rad, inv_r2, inv_2pi_r2, qc);
            }
        }
        // the ragged last tile (n % 64 samples) goes to the wave that would have been next in line for it
        if (nfull < ntiles && (nfull % nwaves) == me) {
            const long long g = nfull * WAVE + lane;
            const SampleIn<COUPLED> in = load_sample<COUPLED>(io, cio, g < io.n ? g : io.n - 1);
            process_tile<L, COUPLED, JMODE, false, NQ, PM>(io, cio, m, in, nfull, lane, rad, inv_r2, inv_2pi_r2, qc);
        }
    }
    if constexpr (NQ > 0) {
        // every wave reaches this barrier (no path above returns): the workgroup's counters are complete, one atomic each
        if (lane == 0) {
            io.q.rec_count[blockIdx.x * WPB + wave] = qc.cnt;
            if (qc.cnt > qc.cap) atomicOr(io.q.flags, 1);
        }
        __syncthreads();
        for (int i = tid; i < NQ * NANG; i += WAVE * WPB) {
            const int q = i / NANG, c = i - q * NANG;
            const unsigned v = reinterpret_cast<const unsigned*>(smem_raw + qc.below_off)[i];
            if (v && q < io.q.nq) atomicAdd(&io.q.below[c * io.q.nq + q], (unsigned long long)v);
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
    const double c0 = io.c0[g], c1 = io.c1[g];
    const PlumeSetup ps = plume_setup(io.P_b[g], c1, io.c2[g], io.c3[g], io.c4[g], io.c5[g], io.torr2pa);
    const double n_neutral = ps.n_neutral, a1 = ps.a1, a2 = ps.a2;
    const double sigma = io.sigma[g], I_B0 = io.I_B0[g];
    const double u1 = 1.0 / (a1 * a1), u2 = 1.0 / (a2 * a2);
    const double A1 = (1.0 - c0) / normaliser(a1, u1, PEM_DPOLY);
    const double A2 = c0 / normaliser(a2, u2, PEM_DPOLY);
    constexpr double H = HALF_PI / 90.0;
    const double s1 = (H * H) * u1, s2 = (H * H) * u2;
    const double r10 = exp(-s1), r20 = exp(-s2), q1 = exp(-2.0 * s1), q2 = exp(-2.0 * s2);
    const bool have_T = io.T != nullptr;
    const double thrust = have_T ? io.T[g] : 0.0;

    int invalid = (a1 <= 0.0) ? 1 : 0;
    // `literal`: the Gaussians by direct exp() instead of the recurrence -- the sample is redone that way when pass 0
    // finds a value below 1e-290 / non-positive or a non-finite amplitude (the deep tail, see exact_chunk above)
    bool literal = false, uncertain = false;
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: integrals + invalid flag; pass 1: profile stores
        for (int r = 0; r < R; ++r) {
            const double rad = radii[r];
            const double decay = exp(-rad * n_neutral * sigma);
            const double j_cex = I_B0 * (1.0 - decay) / (2.0 * PEM_PI * (rad * rad));
            const double base = I_B0 * decay / (rad * rad);
            const double B1 = base * A1, B2 = base * A2;
            if (!__builtin_isfinite(B1) || !__builtin_isfinite(B2)) uncertain = true;
            double e1 = (a1 == 0.0) ? __builtin_nan("") : 1.0, e2 = e1, r1 = r10, r2 = r20, den = 0.0, num = 0.0;
            for (int k = 0; k < NANG; ++k) {
                if (literal) {
                    const double alpha = k == NANG - 1 ? HALF_PI : (double)k * H;
                    const double t1 = alpha / a1, t2 = alpha / a2;
                    e1 = exp(-(t1 * t1));
                    e2 = exp(-(t2 * t2));
                }
                const double f = B1 * e1 + B2 * e2;
                const double ji = f + j_cex;
                if (pass == 0) {
                    invalid |= (ji <= 0.0) ? 1 : 0;
                    if (ji < 1e-290) uncertain = true;
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
        if (pass == 0 && uncertain && !literal) {   // redo pass 0 literally; pass 1 then stores the literal values
            literal = true;
            invalid = (a1 <= 0.0) ? 1 : 0;
            pass = -1;
        }
    }
    if (io.invalid) io.invalid[g] = (uint8_t)invalid;
}

// ---------------------------------------------------------------------------------------------
// sweep_radius arrays, 2 <= R <= RADII_MAX: one WAVE per sample.  For a sample the (91, R) block of j_ion is the outer
// product  e1[k] B1[r] + e2[k] B2[r] + j_cex[r]  and is contiguous in memory: the wave computes the two Gaussians once
// (91 direct exp() each -- literally the reference's expression, so its deep tail comes for free), the per-radius
// amplitudes with lane = radius, and then streams the block with lane = linear index, 512 contiguous bytes per store,
// deciding plume.py:105 on the way.  The divergence integrals are linear in the amplitudes: four Simpson sums of the two
// Gaussians per sample, combined per radius.  The lane-per-sample
// kernel above writes the same block with a stride of 91 R doubles between lanes: 251 GB/s at R = 25 against
// this kernel's several TB/s (tools/radii_probe.py).
// ---------------------------------------------------------------------------------------------
constexpr int RADII_MAX = 256;
struct RadiiArg {   // the sweep radii travel in the kernel arguments: no device allocation, no copy to wait for
    double r[RADII_MAX];
};
__global__ __launch_bounds__(BLOCK) void plume_radii_kernel(PlumeIO io, RadiiArg radii_arg, int R, int ts) {
#pragma clang fp contract(off)
    __shared__ double lds_all[BLOCK / WAVE][2 * 96 + 3 * RADII_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* e1 = lds_all[wave];
    double* e2 = e1 + 96;
    double* B1 = e2 + 96;
    double* B2 = B1 + RADII_MAX;
    double* JC = B2 + RADII_MAX;
    const long long nwaves = (long long)gridDim.x * (BLOCK / WAVE);
    const bool have_T = io.T != nullptr;
    const int step_r = WAVE % R, step_k = WAVE / R;
    // A wave takes `ts` (<= 64) consecutive samples at a time: their parameters are computed once, one lane per sample
    // (coalesced input loads), and handed to the whole wave by shuffles as it walks through the blocks.  The host picks
    // ts = 64 for large batches and smaller tiles when there would otherwise be too few of them to fill the chip.
    const long long ntiles = (io.n + ts - 1) / ts;
    for (long long t = (long long)blockIdx.x * (BLOCK / WAVE) + wave; t < ntiles; t += nwaves) {
    const long long gl = (lane < ts && t * ts + lane < io.n) ? t * ts + lane : io.n - 1;    // idle lanes repeat the last sample
    const double c0_l = io.c0[gl], c1_l = io.c1[gl];
    const PlumeSetup ps_l = plume_setup(io.P_b[gl], c1_l, io.c2[gl], io.c3[gl], io.c4[gl], io.c5[gl], io.torr2pa);
    const double nn_l = ps_l.n_neutral, sigma_l = io.sigma[gl];
    const double IB0_l = io.I_B0[gl];
    const double a1_l = ps_l.a1, a2_l = ps_l.a2;
    const double A1_l = (1.0 - c0_l) / normaliser(a1_l, 1.0 / (a1_l * a1_l), PEM_DPOLY);
    const double A2_l = c0_l / normaliser(a2_l, 1.0 / (a2_l * a2_l), PEM_DPOLY);
    const double thrust_l = have_T ? io.T[gl] : 0.0;
    const int in_tile = (int)(io.n - t * ts < ts ? io.n - t * ts : ts);
    for (int smp = 0; smp < in_tile; ++smp) {
        const long long g = t * ts + smp;
        const double a1 = __shfl(a1_l, smp), a2 = __shfl(a2_l, smp), A1 = __shfl(A1_l, smp), A2 = __shfl(A2_l, smp);
        const double n_neutral = __shfl(nn_l, smp), sigma = __shfl(sigma_l, smp);
        const double I_B0 = __shfl(IB0_l, smp), thrust = __shfl(thrust_l, smp);
        // the two Gaussians of plume.py:99-100 on the 91-point grid, and their four Simpson functionals: the sums of
        // plume.py:117-123 are linear in the amplitudes, den[r] = B1[r] sum_k w_k e1[k] + B2[r] sum_k w_k e2[k]
        double s1d = 0.0, s1n = 0.0, s2d = 0.0, s2n = 0.0;
        for (int k = lane; k < NANG; k += WAVE) {
            const double alpha = k == NANG - 1 ? HALF_PI : (double)k * GRID_H;
            const double t1 = alpha / a1, t2 = alpha / a2;
            const double g1 = exp(-(t1 * t1)), g2 = exp(-(t2 * t2));
            e1[k] = g1;
            e2[k] = g2;
            s1d = __builtin_fma(PEM_SIMPSON_CDEN[k], g1, s1d);
            s1n = __builtin_fma(PEM_SIMPSON_CNUM[k], g1, s1n);
            s2d = __builtin_fma(PEM_SIMPSON_CDEN[k], g2, s2d);
            s2n = __builtin_fma(PEM_SIMPSON_CNUM[k], g2, s2n);
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
            s1d += __shfl_xor(s1d, sh);
            s1n += __shfl_xor(s1n, sh);
            s2d += __shfl_xor(s2d, sh);
            s2n += __shfl_xor(s2n, sh);
        }
        // per radius (lane = radius): amplitudes and the divergence angle
        for (int r0 = 0; r0 < R; r0 += WAVE) {
            const int r = r0 + lane;
            if (r < R) {
                const double rad = radii_arg.r[r];
                const double decay = exp(-rad * n_neutral * sigma);
                const double j_cex = I_B0 * (1.0 - decay) / (2.0 * PEM_PI * (rad * rad));
                const double base = I_B0 * decay / (rad * rad);
                const double b1 = base * A1, b2 = base * A2;
                B1[r] = b1;
                B2[r] = b2;
                JC[r] = j_cex;
                double num = b1 * s1n + b2 * s2n, den = b1 * s1d + b2 * s2d;
                if (!(fabs(b1) + fabs(b2) < 1e300) || ((b1 < 0.0) != (b2 < 0.0) && b1 != 0.0 && b2 != 0.0)) {
                    // Sum as the reference does (rare) when the amplitudes are near the overflow threshold (exp(+x) of a
                    // negative density: its f_k = b1 e1[k] + b2 e2[k] overflows where the factored sums do not), or of
                    // opposite sign (c0 outside [0, 1]): the reference cancels angle by angle, the factored form would
                    // cancel two large sums at the end
                    num = 0.0;
                    den = 0.0;
                    for (int k = 0; k < NANG; ++k) {
                        const double f = b1 * e1[k] + b2 * e2[k];
                        den = __builtin_fma(PEM_SIMPSON_CDEN[k], f, den);
                        num = __builtin_fma(PEM_SIMPSON_CNUM[k], f, num);
                    }
                }
                double cos_div = num / den;
                if (cos_div == __builtin_inf()) cos_div = __builtin_nan("");
                io.div[(size_t)g * R + r] = acos(cos_div);
                if (have_T) io.Tc[(size_t)g * R + r] = thrust * cos_div;
            }
        }
        wave_lds_sync();
        // the (91, R) block, contiguous: lane = linear index k R + r; plume.py:105 is decided on the way
        double* dst = io.j_ion + (size_t)g * NANG * R;
        bool bad = a1 <= 0.0;
        {
            int k = lane / R, r = lane - k * R;
            for (int idx = lane; idx < NANG * R; idx += WAVE) {
                const double ji = (B1[r] * e1[k] + B2[r] * e2[k]) + JC[r];
                bad |= ji <= 0.0;
                __builtin_nontemporal_store(ji, dst + idx);
                r += step_r;
                k += step_k;
                if (r >= R) {
                    r -= R;
                    ++k;
                }
            }
        }
        const bool invalid = __ballot(bad) != 0;
        if (invalid)   // plume.py:106: the whole block becomes 1e-20 (rare: a second pass over it)
            for (int idx = lane; idx < NANG * R; idx += WAVE) dst[idx] = 1e-20;
        if (io.invalid && lane == 0) io.invalid[g] = (uint8_t)invalid;
        wave_lds_sync();   // the staged rows are rewritten for the next sample
    }
    }
}

// ---------------------------------------------------------------------------------------------
// sweep_radius arrays, RADII_SMALL < R <= RMID_MAX (tests/test_plume.py:31 uses 25): the recipe of the few-radii kernel below
// applied to the wave-per-sample kernel above -- several samples in flight per wave, the block staged in LDS in final order,
// 16-byte stores of whole contiguous runs.  G = 64 / R samples share a wave (7 at 9 radii ... 1 above 32): lane (grp, r) owns radius
// r of sample grp, keeps its amplitudes in registers and walks the 91 angles; the two Gaussians of a sample (by recurrence from
// three exp per beam and lane, literal exp() where the reference's own has left the normal range) are read from LDS as one
// broadcast 16-byte word per angle.
// What the lane produces -- b1[r] e1[k] + b2[r] e2[k] + j_cex[r], the Simpson sums of plume.py:117-123 taken angle by angle as the
// reference takes them -- goes to an LDS tile laid out as j_ion is, `kc` rows of every sample at a time (8 KB per wave), and
// leaves as runs of kc R contiguous doubles: one leading 8-byte store where a run starts on an odd double (the LDS copy is
// placed with the same parity), then 1 KiB per instruction.  Against the kernel above this halves the LDS reads per value
// (2.5 instead of 5), removes the per-value index arithmetic and never assembles a cache line from 8-byte pieces.
// ---------------------------------------------------------------------------------------------
// A run of `len` doubles from LDS to `dst`, the whole wave on it.  `from` has the 16-byte parity of `dst` (the LDS copy is placed
// so).  Store instructions that cover WHOLE 128-byte lines are what the memory system wants: with every instruction straddling
// a line boundary the same kernels run a quarter slower (block sizes 91 R x 8 bytes: 4.65 TB/s at R = 32, 3.39 at R = 33;
// profiles/radii_mid_r03.txt).  So: the doubles up to the next line boundary as one partial instruction of 8-byte stores, then
// 16 bytes per lane, 1 KiB per instruction, line-aligned; an odd double left at the end goes out alone.
__device__ __forceinline__ void stream_run(const double* from, double* dst, int len, int lane) {
#if defined(PEM_RMID_EXP) && PEM_RMID_EXP == 1     // experiment: the phases without their stores (tools/rmid_ab_probe.py)
    if (len >= 0) return;
#endif
    int head = (int)((0 - (reinterpret_cast<uintptr_t>(dst) >> 3)) & 15);
    head = head < len ? head : len;
    if (lane < head) __builtin_nontemporal_store(from[lane], dst + lane);
    const int body = (len - head) >> 1;
    const f64x2* s2 = reinterpret_cast<const f64x2*>(from + head);
    f64x2* d2 = reinterpret_cast<f64x2*>(dst + head);
    for (int i = lane; i < body; i += WAVE) stream_store(s2[i], &d2[i]);
    if (((len - head) & 1) && lane == 0) __builtin_nontemporal_store(from[len - 1], dst + (len - 1));
}

constexpr int RADII_SMALL = 8;                  // up to here: the recurrence kernel with the radii in registers (plume_rfew_kernel)
constexpr int RMID_MAX = 64;
constexpr int RMID_G_MAX = 5;                   // samples in flight per wave the staged kernel is instantiated for (R >= 11)
// doubles of staged rows per wave: 10 KB -- with the Gaussians' 1.5 KB per sample what three workgroups per CU leave each other.
// (Round 4, profiles/radii_mid_r04.txt: 512 / 768 / 1024 / 1280 doubles give 3.08 / 3.38 / 3.58 / 3.82 TB/s at 17 radii, 3.36 / 3.55 /
// 3.73 / 3.85 at 25; radius counts whose rows are whole lines -- 32, 64 -- do not care.)
#ifndef PEM_RMID_TILE_DOUBLES
#define PEM_RMID_TILE_DOUBLES 1280
#endif
constexpr int RMID_TILE = PEM_RMID_TILE_DOUBLES;
struct RadiiMidArg {
    double r[RMID_MAX];
};
constexpr int RMID_ES = 97;                     // 16-byte words of E per sample: an odd stride, so that the G broadcast reads of an
                                                // instruction fall on different banks (96: all on the same ones, G-way conflict)
template <int G>
constexpr int rmid_wave_doubles() { return ((G * RMID_ES * 2 + 1) & ~1) + RMID_TILE + 4; }   // E | tile | the samples' invalid flags (G <= 7 ints)

#ifndef PEM_RMID_WAVES
#define PEM_RMID_WAVES 3
#endif
// (round 4) S samples share a wave in P passes: the S R (sample, radius) pairs are dealt over the lanes pass by pass -- pair
// f = 64 p + lane is radius f % R of sample f / R -- instead of 64 / R whole samples side by side with the lanes past their radii
// idle.  With one pass (S = 64 / R) that is the round-3 kernel: 25 radii used 50 lanes of 64 and ran at 0.78 of what 32 radii
// reach, 33 radii 33 lanes; five samples in two passes use 125 of 128 lane slots, three samples of 33 radii 99 of 128.
// (four samples' Gaussians and rows, or a second pass' amplitudes, leave LDS / registers for two waves per SIMD only)
template <int S, int P>
constexpr int rmid_waves_per_simd() { return (S >= 4 || P >= 2) ? 2 : PEM_RMID_WAVES; }
template <int S, int P>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(rmid_waves_per_simd<S, P>())))
void plume_rmid_kernel(PlumeIO io, RadiiMidArg radii_arg, int R, int ts) {
#pragma clang fp contract(off)
    constexpr int RS = (RMID_TILE / S) & ~1;    // doubles of the tile per sample (even)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* mine = reinterpret_cast<double*>(smem_raw) + (size_t)wave * rmid_wave_doubles<S>();
    double2* E = reinterpret_cast<double2*>(mine);   // [S][RMID_ES] {e1[k], e2[k]}
    double* tile = mine + ((S * RMID_ES * 2 + 1) & ~1);   // [S][RS] staged rows (16-byte aligned)
    int* badflag = reinterpret_cast<int*>(tile + RMID_TILE);   // [S] (the two spare doubles of the tile hold up to four; S <= 7: see rmid_wave_doubles)
    // this lane's pairs: (sample of the group, radius) per pass; a lane past the S R pairs repeats the last pair and keeps nothing
    int grp[P], rr[P];
    bool on[P];
    double rad[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int f = 64 * p + lane;
        on[p] = f < S * R;
        const int ff = on[p] ? f : S * R - 1;
        grp[p] = ff / R;
        rr[p] = ff - grp[p] * R;
        rad[p] = radii_arg.r[rr[p]];
    }
    const int kc = (RS - 2) / R;                      // rows per chunk: kc R + 1 <= RS - 1
    const long long nwaves = (long long)gridDim.x * (BLOCK / WAVE);
    const bool have_T = io.T != nullptr;
    const long long ntiles = (io.n + ts - 1) / ts;
    for (long long t = (long long)blockIdx.x * (BLOCK / WAVE) + wave; t < ntiles; t += nwaves) {
        // parameters of the tile's samples, one lane per sample (as plume_radii_kernel)
        const long long gl = (lane < ts && t * ts + lane < io.n) ? t * ts + lane : io.n - 1;
        const double c0_l = io.c0[gl], c1_l = io.c1[gl];
        const PlumeSetup ps_l = plume_setup(io.P_b[gl], c1_l, io.c2[gl], io.c3[gl], io.c4[gl], io.c5[gl], io.torr2pa);
        // (sigma, I_B0 and T of a sample are read again by its own lanes when its group comes up -- three loads that hit the cache --
        // instead of being carried in registers across the tile: with them the kernel was eight registers over three waves per SIMD)
        const double nn_l = ps_l.n_neutral;
        const double a1_l = ps_l.a1, a2_l = ps_l.a2;
        const double A1_l = (1.0 - c0_l) / normaliser(a1_l, 1.0 / (a1_l * a1_l), PEM_DPOLY);
        const double A2_l = c0_l / normaliser(a2_l, 1.0 / (a2_l * a2_l), PEM_DPOLY);
        const int in_tile = (int)(io.n - t * ts < ts ? io.n - t * ts : ts);
        for (int s0 = 0; s0 < in_tile; s0 += S) {
            // The Gaussians of the group's samples: the R lanes of a (sample, pass) take CHK consecutive angles each and advance
            // e_k = exp(-(k h / a)^2) by the two-term recurrence of the R = 1 kernel (e_{k+1} = e_k r_k, r_{k+1} = r_k q) from
            // three branch-free exp per beam.  A chunk in which the reference's own exp() has left the normal range (a value
            // below 1e-290), or whose widths are not finite numbers, is evaluated literally as the reference does
            // (plume.py:99-100), deep tail included.  (Straight into LDS: kept in a register array first the kernel spilled.)
            if (lane < S) badflag[lane] = 0;
            const int chk = (NANG + R - 1) / R;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int k0 = rr[p] * chk;
                const int sm = s0 + grp[p] < in_tile ? s0 + grp[p] : in_tile - 1;
                const double a1g = __shfl(a1_l, sm), a2g = __shfl(a2_l, sm);
                const double s1 = (GRID_H * GRID_H) * (1.0 / (a1g * a1g)), s2 = (GRID_H * GRID_H) * (1.0 / (a2g * a2g));
                double e1 = exp_nonpos(-(double)(k0 * k0) * s1), r1 = exp_nonpos(-(double)(2 * k0 + 1) * s1);
                double e2 = exp_nonpos(-(double)(k0 * k0) * s2), r2 = exp_nonpos(-(double)(2 * k0 + 1) * s2);
                const double q1 = exp_nonpos(-2.0 * s1), q2 = exp_nonpos(-2.0 * s2);
                double lo = __builtin_inf();
                for (int i = 0; i < chk; ++i) {
                    if (on[p] && k0 + i < NANG) E[grp[p] * RMID_ES + k0 + i] = make_double2(e1, e2);
                    lo = fmin(lo, fmin(e1, e2));
                    e1 *= r1;
                    r1 *= q1;
                    e2 *= r2;
                    r2 *= q2;
                }
                if (!(lo >= 1e-290) || !__builtin_isfinite(s1) || !__builtin_isfinite(s2)) {
                    for (int i = 0; i < chk; ++i) {
                        const int k = k0 + i;
                        const double alpha = k >= NANG - 1 ? HALF_PI : (double)k * GRID_H;
                        const double t1 = alpha / a1g, t2 = alpha / a2g;
                        if (on[p] && k < NANG) E[grp[p] * RMID_ES + k] = make_double2(exp(-(t1 * t1)), exp(-(t2 * t2)));
                    }
                }
            }
            // this lane's (sample, radius) pairs: amplitudes of plume.py:95-100
            bool smp_on[P];
            long long g[P];
            double b1[P], b2[P], jcx[P], den[P], num[P], thrust[P];
            bool bad[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                smp_on[p] = on[p] && s0 + grp[p] < in_tile;
                const int src = s0 + grp[p] < in_tile ? s0 + grp[p] : in_tile - 1;      // an idle pair repeats the last sample and stores nothing
                g[p] = t * ts + src;
                const double A1 = __shfl(A1_l, src), A2 = __shfl(A2_l, src);
                const double n_neutral = __shfl(nn_l, src), sigma = io.sigma[g[p]];
                const double I_B0 = io.I_B0[g[p]];
                thrust[p] = have_T ? io.T[g[p]] : 0.0;
                const double decay = exp(-rad[p] * n_neutral * sigma);
                jcx[p] = I_B0 * (1.0 - decay) / (2.0 * PEM_PI * (rad[p] * rad[p]));
                const double base = I_B0 * decay / (rad[p] * rad[p]);
                b1[p] = base * A1;
                b2[p] = base * A2;
                den[p] = 0.0;
                num[p] = 0.0;
                bad[p] = false;
            }
            wave_lds_sync();
            for (int k0 = 0; k0 < NANG; k0 += kc) {
                const int rows = NANG - k0 < kc ? NANG - k0 : kc;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    // where this pair's run starts in j_ion: the LDS copy gets the same parity
                    const double* gdst = io.j_ion + ((size_t)g[p] * NANG + k0) * R;
                    double* run = tile + grp[p] * RS + (int)((reinterpret_cast<uintptr_t>(gdst) >> 3) & 1);
#if defined(PEM_RMID_EXP) && PEM_RMID_EXP == 2     // experiment: the stores without the rows' arithmetic (one row computed per chunk)
                    if (on[p]) {
                        for (int kk = 0; kk < (rows > 1 ? 1 : rows); ++kk) {
#else
                    if (on[p]) {
#pragma unroll 4
                        for (int kk = 0; kk < rows; ++kk) {
#endif
                            const int k = k0 + kk;
                            const double2 ee = E[grp[p] * RMID_ES + k];
                            const double f = b1[p] * ee.x + b2[p] * ee.y;      // j_beam + j_scat
                            const double ji = f + jcx[p];                      // plume.py:102
                            run[kk * R + rr[p]] = ji;
                            den[p] = __builtin_fma(PEM_SIMPSON_CDEN[k], f, den[p]);
                            num[p] = __builtin_fma(PEM_SIMPSON_CNUM[k], f, num[p]);
                            bad[p] |= ji <= 0.0;
                        }
                    }
                }
                wave_lds_sync();
                // the runs leave one after the other, the whole wave on each
                for (int gi = 0; gi < S; ++gi) {
                    if (s0 + gi >= in_tile) break;
                    double* dst = io.j_ion + ((size_t)(t * ts + s0 + gi) * NANG + k0) * R;
                    stream_run(tile + gi * RS + (int)((reinterpret_cast<uintptr_t>(dst) >> 3) & 1), dst, rows * R, lane);
                }
                wave_lds_sync();
            }
            // plume.py:105: a sample is invalid if alpha1 <= 0 or any of its values is <= 0 -- its pairs sit on several lanes and passes
#pragma unroll
            for (int p = 0; p < P; ++p)
                if (smp_on[p] && bad[p]) badflag[grp[p]] = 1;
            wave_lds_sync();
#pragma unroll
            for (int p = 0; p < P; ++p) {
                double cos_div = num[p] / den[p];   // plume.py:124-127
                if (cos_div == __builtin_inf()) cos_div = __builtin_nan("");
                const int src = s0 + grp[p] < in_tile ? s0 + grp[p] : in_tile - 1;
                const bool invalid = __shfl(a1_l, src) <= 0.0 || badflag[grp[p]] != 0;
                if (smp_on[p]) {
                    io.div[(size_t)g[p] * R + rr[p]] = acos(cos_div);
                    if (have_T) io.Tc[(size_t)g[p] * R + rr[p]] = thrust[p] * cos_div;
                    if (io.invalid && rr[p] == 0) io.invalid[g[p]] = (uint8_t)invalid;
                }
            }
            // plume.py:106: the whole block of an invalid sample becomes 1e-20 (rare: a second pass over it, the whole wave on each)
            for (int gi = 0; gi < S; ++gi) {
                if (s0 + gi >= in_tile) break;
                const bool invalid = __shfl(a1_l, s0 + gi) <= 0.0 || badflag[gi] != 0;
                if (invalid) {
                    double* blk = io.j_ion + (size_t)(t * ts + s0 + gi) * NANG * R;
                    for (int idx = lane; idx < NANG * R; idx += WAVE) blk[idx] = 1e-20;
                }
            }
            wave_lds_sync();
        }
        wave_lds_sync();
    }
}

struct RadiiSmallArg {
    double r[RADII_SMALL];
};

// ---------------------------------------------------------------------------------------------
// FEW radii by recurrence: the R = 1 fast path generalised (2 <= R <= RADII_SMALL).  The wave-per-sample kernel above is
// bound by LATENCY, not by issue or HBM: a wave has one sample in flight, and every sample is a chain Gaussians -> LDS ->
// wave reduction -> LDS -> block stream (5.4k cycles per sample measured at R = 2 where the instruction count says 1.3k;
// moving the per-radius work out of that chain gained 10-19 %: profiles/radii_probe_r02.txt).  Here a wave works on 8 samples at a time as plume_r1_kernel does: lane (s, c) walks
// angles k = 12 c .. 12 c + 11 of sample s, advancing the two Gaussians by the two-term recurrence (4 multiplies per
// angle) from chunk starts that come from the same recurrence at stride 12; per angle it forms the R values
// b1[r] e1 + b2[r] e2 + j_cex[r] from amplitudes it holds in registers and puts them -- R consecutive doubles -- into an LDS
// tile laid out as j_ion is, which leaves as 1-KiB-per-instruction 16-byte stores when the round is done.  The Simpson functionals of the two Gaussians ride along (4 FMAs per angle) and are folded over
// the 8 chunk lanes; cos_div / arccos / T_c of all (sample, radius) pairs of the 64-sample tile follow, one lane per pair.
// "Equal to the reference" in the deep tail is kept as in the R = 1 path: a chunk with a value below 1e-290 (or <= 0, or
// a non-finite amplitude) is re-evaluated literally with direct exp(); amplitudes of opposite sign or near overflow send
// the pair's divergence integrals through the literal angle-by-angle sum.
// LDS (doubles): shared: simpson[96][2] | dpoly[384];  per wave: params[8][64] | PB[64][R][3] | tile[8][91][R]
// ---------------------------------------------------------------------------------------------
constexpr int RF_L = 8, RF_S = WAVE / RF_L, RF_CH = 12;
static_assert(RF_L * RF_CH >= NANG && RF_L * RF_CH <= NSIMP, "8 chunks of 12 angles cover the 91-point grid inside the padded table");
// per wave: params[8][64] | PB[64][R][3] | tile[8][91][R] + 2; the workgroup has as many waves as fit 160 KB beside the tables
template <int R>
constexpr int rfew_wave_doubles() { return 8 * WAVE + 3 * WAVE * R + RF_S * NANG * R + 2; }
template <int R>
constexpr int rfew_waves() { return R <= 4 ? 4 : (R <= 6 ? 3 : 2); }

template <int R>   // the number of radii is a compile-time constant: amplitudes and the values of two angles live in registers
__global__ __launch_bounds__(WAVE * rfew_waves<R>()) void plume_rfew_kernel(PlumeIO io, RadiiSmallArg radii_arg) {
    constexpr int RM = R;
    // A round's 8 x 91 x R values go to an LDS tile in final order and leave as 1-KiB-per-instruction 16-byte stores, as in
    // the R = 1 path.  (Stored straight from the angle loop instead -- 16 bytes per lane, 64 separate pieces per instruction
    // -- the kernel ran at 2.1-2.4 TB/s; staged 3.7-4.7: profiles/radii_probe_r02.txt.)  The tile grows with R, so the
    // workgroup shrinks: 4 waves up to R = 4, 3 up to 6, 2 for 7 and 8.
    constexpr int NW = rfew_waves<R>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    double2* tab_simpson = reinterpret_cast<double2*>(lds);
    double* tab_poly = lds + 2 * NSIMP;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int per_wave = rfew_wave_doubles<R>();
    double* params = lds + TABLE_DOUBLES + wave * per_wave;     // rows: a1 a2 | r0 G E (beam 1) | r0 G E (beam 2)
    // a sample's rows 2..5 are free once its round has read them: they then carry its {s1d, s1n, s2d, s2n}
    double* PB = params + 8 * WAVE;                             // [64][R][3] {b1, b2, j_cex}
    double* tile = PB + 3 * WAVE * R;                           // [8][91][R] + 2
    for (int i = tid; i < NSIMP; i += WAVE * NW)
        tab_simpson[i] = i < NANG ? make_double2(PEM_SIMPSON_CDEN[i], PEM_SIMPSON_CNUM[i]) : make_double2(0.0, 0.0);
    for (int i = tid; i < PEM_NDI * PEM_NDC; i += WAVE * NW) tab_poly[i] = PEM_DPOLY[i];
    __syncthreads();

    const int s = lane % RF_S, c = lane / RF_S, k0 = c * RF_CH;
    const double2* my_w = tab_simpson + k0;
    const bool have_T = io.T != nullptr;
    const long long ntiles = (io.n + WAVE - 1) / WAVE;
    const long long nwaves = (long long)gridDim.x * NW;
    const size_t blk = (size_t)NANG * R;
    for (long long t = (long long)blockIdx.x * NW + wave; t < ntiles; t += nwaves) {
        const int in_tile = (int)(io.n - t * WAVE < WAVE ? io.n - t * WAVE : WAVE);
        const long long gl = lane < in_tile ? t * WAVE + lane : io.n - 1;     // idle lanes repeat the last sample
        // ------------------------------ prelude: one lane per sample ------------------------------
        unsigned literal_l = 0;
        {
            const double c0_l = io.c0[gl];
            const PlumeSetup ps = plume_setup(io.P_b[gl], io.c1[gl], io.c2[gl], io.c3[gl], io.c4[gl], io.c5[gl], io.torr2pa);
            const double sigma_l = io.sigma[gl], IB0_l = io.I_B0[gl];
            const double u1 = 1.0 / (ps.a1 * ps.a1), u2 = 1.0 / (ps.a2 * ps.a2);
            const double A1 = (1.0 - c0_l) / normaliser(ps.a1, u1, tab_poly);
            const double A2 = c0_l / normaliser(ps.a2, u2, tab_poly);
            const double s1 = (GRID_H * GRID_H) * u1, s2 = (GRID_H * GRID_H) * u2;
            params[0 * WAVE + lane] = ps.a1;
            params[1 * WAVE + lane] = ps.a2;
            params[2 * WAVE + lane] = exp_nonpos(-s1);
            params[3 * WAVE + lane] = exp_nonpos(-(2.0 * RF_CH) * s1);
            params[4 * WAVE + lane] = exp_nonpos(-(double)(RF_CH * RF_CH) * s1);
            params[5 * WAVE + lane] = exp_nonpos(-s2);
            params[6 * WAVE + lane] = exp_nonpos(-(2.0 * RF_CH) * s2);
            params[7 * WAVE + lane] = exp_nonpos(-(double)(RF_CH * RF_CH) * s2);
            for (int r = 0; r < R; ++r) {
#pragma clang fp contract(off)
                const double rad = radii_arg.r[r];
                const double decay = exp(-rad * ps.n_neutral * sigma_l);
                const double j_cex = IB0_l * (1.0 - decay) / (2.0 * PEM_PI * (rad * rad));
                const double base = IB0_l * decay / (rad * rad);
                const double b1 = base * A1, b2 = base * A2;
                double* pb = PB + (lane * R + r) * 3;
                pb[0] = b1;
                pb[1] = b2;
                pb[2] = j_cex;
                if (!(fabs(b1) + fabs(b2) < 1e300) || ((b1 < 0.0) != (b2 < 0.0) && b1 != 0.0 && b2 != 0.0)) literal_l |= 1u << r;
            }
        }
        const unsigned long long a1_nonpos = __ballot(params[0 * WAVE + lane] <= 0.0);
        wave_lds_sync();
        unsigned long long inv_mask = 0;
        // ------------------------------ rounds: 8 samples, 8 chunk lanes each ------------------------------
        for (int round = 0; round < RF_L; ++round) {
            const int smp = round * RF_S + s;
            const double r01 = params[2 * WAVE + smp], G1 = params[3 * WAVE + smp], E1 = params[4 * WAVE + smp];
            const double r02 = params[5 * WAVE + smp], G2 = params[6 * WAVE + smp], E2 = params[7 * WAVE + smp];
            double b1[RM], b2[RM], jc[RM];
            bool finite = true;
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                const double* pb = PB + (smp * R + (r < R ? r : 0)) * 3;
                b1[r] = pb[0];
                b2[r] = pb[1];
                jc[r] = pb[2];
                finite = finite && __builtin_isfinite(b1[r]) && __builtin_isfinite(b2[r]);
            }
            // chunk start k0 = 12 c by the coarse recurrence: e_{k0} = E^(c^2), r_{k0} = r0 G^c
            double e1 = 1.0, e2 = 1.0, rr1 = r01, rr2 = r02, rho1 = E1, rho2 = E2;
            if (params[0 * WAVE + smp] == 0.0) e1 = e2 = __builtin_nan("");   // alpha1 = 0: exp(-(0/0)^2) is NaN in the reference
            const double E1sq = E1 * E1, E2sq = E2 * E2;
#pragma unroll
            for (int i = 0; i < RF_L - 1; ++i) {
                if (i < c) {
                    e1 *= rho1;
                    rho1 *= E1sq;
                    rr1 *= G1;
                    e2 *= rho2;
                    rho2 *= E2sq;
                    rr2 *= G2;
                }
            }
            const double q1 = r01 * r01, q2 = r02 * r02;
            double part[4] = {0.0, 0.0, 0.0, 0.0}, lo = __builtin_inf();
            double* dst = tile + (size_t)s * blk + (size_t)k0 * R;
            // The chunk's values are one run of 12 R consecutive doubles of the tile (laid out as j_ion is).  Two angles = 2 R doubles per iteration,
            // stored as R 16-byte pieces.  For an odd R the run of an odd sample starts at an odd double (the start is
            // (g 91 + 12 c) R doubles into a 16-byte aligned array): such a lane stores its first double on its own, then
            // pieces shifted by one element (the last element of an iteration is carried into the next), and the last
            // double on its own again -- selected per lane, so that every 16-byte store has all 64 lanes in it.
            const bool mis = (R & 1) && (smp & 1);
            double carry = 0.0;
#pragma unroll 1
            for (int jj = 0; jj < RF_CH; jj += 2) {
                double ev[2 * RM];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double2 w = my_w[jj + h];
#pragma unroll
                    for (int r = 0; r < RM; ++r) ev[h * RM + r] = fma(b1[r], e1, b2[r] * e2) + jc[r];
                    part[0] = fma(w.x, e1, part[0]);
                    part[1] = fma(w.y, e1, part[1]);
                    part[2] = fma(w.x, e2, part[2]);
                    part[3] = fma(w.y, e2, part[3]);
                    e1 *= rr1;
                    rr1 *= q1;
                    e2 *= rr2;
                    rr2 *= q2;
                }
                double* d = dst + (size_t)jj * R;
                if (k0 + jj + 1 < NANG) {                  // both angles exist (all but the last iterations of the last chunk)
#pragma unroll
                    for (int i = 0; i < 2 * RM; ++i) lo = fmin(lo, ev[i]);
                    {
                        if constexpr ((R & 1) == 0) {
#pragma unroll
                            for (int q = 0; q < RM; ++q) *reinterpret_cast<f64x2*>(d + 2 * q) = f64x2{ev[2 * q], ev[2 * q + 1]};
                        } else {
                            if (mis && jj == 0) d[0] = ev[0];
#pragma unroll
                            for (int q = 0; q < RM; ++q) {
                                f64x2 pr;
                                pr.x = mis ? (q == 0 ? carry : ev[2 * q - 1]) : ev[2 * q];
                                pr.y = mis ? ev[2 * q] : ev[2 * q + 1];
                                if (!(mis && jj == 0 && q == 0)) *reinterpret_cast<f64x2*>(d + 2 * q - (mis ? 1 : 0)) = pr;
                            }
                            carry = ev[2 * RM - 1];
                            if (mis && jj + 2 >= RF_CH) d[2 * RM - 1] = carry;      // the run ends here: its last double
                        }
                    }
                } else {                                   // past 90 degrees: at most the first angle of the pair exists
                    if (k0 + jj < NANG) {
#pragma unroll
                        for (int r = 0; r < RM; ++r) lo = fmin(lo, ev[r]);
                        {
                            if ((R & 1) && mis && jj > 0) d[-1] = carry;           // the carried double of the iteration before
#pragma unroll
                            for (int r = 0; r < RM; ++r) d[r] = ev[r];
                        }
                    } else if ((R & 1) && mis && jj > 0 && k0 + jj - 1 < NANG) {
                        d[-1] = carry;
                    }
                    carry = 0.0;
                    // (nothing further of this chunk exists; the recurrence runs on harmlessly)
                }
            }
            // deep tail: where the reference's own exp() has left the normal range the chunk is evaluated literally
            const bool uncertain = lo < 1e-290 || !finite;
            if (__ballot(uncertain)) {
                if (uncertain) {
#pragma clang fp contract(off)
                    const double a1s = params[0 * WAVE + smp], a2s = params[1 * WAVE + smp];
                    part[0] = part[1] = part[2] = part[3] = 0.0;
                    lo = __builtin_inf();
                    for (int j = 0; j < RF_CH; ++j) {
                        const int k = k0 + j;
                        if (k >= NANG) break;
                        const double alpha = k == NANG - 1 ? HALF_PI : (double)k * GRID_H;
                        const double t1 = alpha / a1s, t2 = alpha / a2s;
                        const double g1 = exp(-(t1 * t1)), g2 = exp(-(t2 * t2));
                        for (int r = 0; r < R; ++r) {
                            const double* pb = PB + (smp * R + r) * 3;
                            const double ji = (pb[0] * g1 + pb[1] * g2) + pb[2];
                            lo = fmin(lo, ji);
                            dst[(size_t)j * R + r] = ji;
                        }
                        part[0] = __builtin_fma(my_w[j].x, g1, part[0]);
                        part[1] = __builtin_fma(my_w[j].y, g1, part[1]);
                        part[2] = __builtin_fma(my_w[j].x, g2, part[2]);
                        part[3] = __builtin_fma(my_w[j].y, g2, part[3]);
                    }
                }
            }
            // fold the 8 chunk lanes of a sample: Simpson functionals and plume.py:105
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int sh = RF_S; sh < WAVE; sh <<= 1) part[q] += __shfl_xor(part[q], sh);
            }
            if (c == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) params[(2 + q) * WAVE + smp] = part[q];
            }
            unsigned long long bad = __ballot(lo <= 0.0);
#pragma unroll
            for (int sh = RF_S; sh < WAVE; sh <<= 1) bad |= bad >> sh;
            bad = (bad | (a1_nonpos >> (round * RF_S))) & ((1ull << RF_S) - 1);
            inv_mask |= bad << (round * RF_S);
            if ((bad >> s) & 1) {   // plume.py:106: the whole block of an invalid sample becomes 1e-20 (rare)
                for (int j = 0; j < RF_CH; ++j)
                    if (k0 + j < NANG)
                        for (int r = 0; r < R; ++r) dst[(size_t)j * R + r] = 1e-20;
            }
            {
                // the round's samples are one contiguous, 16-byte aligned piece of j_ion (round * 8 is even)
                wave_lds_sync();
                const long long first = t * WAVE + (long long)round * RF_S;
                long long valid = (io.n - first) * (long long)blk;          // doubles of this round that exist
                if (valid > (long long)RF_S * (long long)blk) valid = (long long)RF_S * (long long)blk;
                // (for an odd R every other round starts 64 bytes into a 128-byte line: stream_run brings the body back onto
                // line boundaries with one leading partial instruction)
                if (valid > 0) stream_run(tile, io.j_ion + (size_t)first * blk, (int)valid, lane);
                wave_lds_sync();   // the tile is rewritten by the next round
            }
        }
        wave_lds_sync();
        if (io.invalid && lane < in_tile) io.invalid[t * WAVE + lane] = (uint8_t)((inv_mask >> lane) & 1);
        // ------------------------------ postlude: one lane per (sample, radius) pair ------------------------------
        const int pairs = in_tile * R;
        for (int idx = lane; idx < pairs; idx += WAVE) {
            const int smp = idx / R, r = idx - smp * R;
            const unsigned literal = (unsigned)__shfl((int)literal_l, smp);
            const double* pb = PB + idx * 3;
            const double s1d = params[2 * WAVE + smp], s1n = params[3 * WAVE + smp], s2d = params[4 * WAVE + smp], s2n = params[5 * WAVE + smp];
            double num, den;
            {
#pragma clang fp contract(off)
                num = pb[0] * s1n + pb[1] * s2n;
                den = pb[0] * s1d + pb[1] * s2d;
            }
            if ((literal >> r) & 1) {   // the reference's own summation order (amplitudes of opposite sign / near overflow)
#pragma clang fp contract(off)
                const double a1s = params[0 * WAVE + smp], a2s = params[1 * WAVE + smp];
                num = 0.0;
                den = 0.0;
                for (int k = 0; k < NANG; ++k) {
                    const double alpha = k == NANG - 1 ? HALF_PI : (double)k * GRID_H;
                    const double t1 = alpha / a1s, t2 = alpha / a2s;
                    const double f = pb[0] * exp(-(t1 * t1)) + pb[1] * exp(-(t2 * t2));
                    den = __builtin_fma(tab_simpson[k].x, f, den);
                    num = __builtin_fma(tab_simpson[k].y, f, num);
                }
            }
            double cos_div = num / den;
            if (cos_div == __builtin_inf()) cos_div = __builtin_nan("");
            const long long g = t * WAVE + smp;
            io.div[(size_t)g * R + r] = acos(cos_div);
            if (have_T) io.Tc[(size_t)g * R + r] = io.T[g] * cos_div;
        }
        wave_lds_sync();   // params / PB are rewritten by the next tile
    }
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

// u_ion(z) of sim_hallthruster.jl:46-47 on z = range(z0, z1, length = ncells): one row per sample, lanes along z
__global__ __launch_bounds__(BLOCK) void thruster_uion_kernel(long long n, const double* __restrict__ v_exh, double z0,
                                                              double z1, int ncells, double* __restrict__ z_out,
                                                              double* __restrict__ u_ion) {
    const long long total = n * ncells;
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long idx = (long long)blockIdx.x * BLOCK + threadIdx.x; idx < total; idx += stride) {
        const long long i = idx / ncells;
        const int c = (int)(idx - i * ncells);
        const double z = z0 + (z1 - z0) * ((double)c / (double)(ncells - 1));
        if (i == 0 && z_out) z_out[c] = z;
        u_ion[idx] = v_exh[i] / (1.0 + exp(-100.0 * (z - 0.04)));
    }
}

// The two filters hallthruster_jl applies to a finished run, batched (thruster.py:490-502):
//   bit 0: thrust < 0 or beam current < 0 (non-physical);  bit 1: the ion velocity peaks before `threshold`
// One wave per sample row: strided argmax (first maximum wins, as np.argmax) + wave reduction.
__global__ __launch_bounds__(BLOCK) void thruster_filter_kernel(long long n, int ncells, const double* __restrict__ u_ion,
                                                                const double* __restrict__ z, double threshold,
                                                                int use_shock, const double* __restrict__ T,
                                                                const double* __restrict__ I_B0,
                                                                uint8_t* __restrict__ flags) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * BLOCK) >> 6;
    for (long long i = wave; i < n; i += nwaves) {
        int flag = 0;
        if (lane == 0) {
            const double t = T ? T[i] : 0.0, b = I_B0 ? I_B0[i] : 0.0;
            flag = (t < 0.0 || b < 0.0) ? 1 : 0;
        }
        if (use_shock) {
            double best = -__builtin_inf();
            int where = 0x7fffffff;
            bool any_nan = false;
            for (int c = lane; c < ncells; c += 64) {
                const double u = u_ion[i * ncells + c];
                any_nan |= (u != u);
                if (u > best) {
                    best = u;
                    where = c;
                }
            }
            // np.argmax returns the first NaN if there is one; otherwise the first maximum
            int nan_at = 0x7fffffff;
            if (any_nan)
                for (int c = lane; c < ncells; c += 64)
                    if (u_ion[i * ncells + c] != u_ion[i * ncells + c]) {
                        nan_at = c;
                        break;
                    }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const double ob = __shfl_xor(best, m);
                const int ow = __shfl_xor(where, m);
                const int on = __shfl_xor(nan_at, m);
                if (ob > best || (ob == best && ow < where)) {
                    best = ob;
                    where = ow;
                }
                nan_at = on < nan_at ? on : nan_at;
            }
            if (lane == 0) {
                const int arg = nan_at != 0x7fffffff ? nan_at : (where == 0x7fffffff ? 0 : where);
                if (z[arg] < threshold) flag |= 2;
            }
        }
        if (lane == 0) flags[i] = (uint8_t)flag;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
std::atomic<int> g_lanes{4};
std::atomic<int> g_device{-1};   // process default of the host-pointer entry points (pem_init); -1: the calling thread's

// Host-pointer entry points run on the device given to pem_init, whichever thread calls them: a new thread's current
// HIP device is 0, which is the wrong card for the worker threads of a one-process-per-GPU rank (gen_data.py:448-456
// evaluates models on Thread pools).
int use_default_device() {
    const int d = g_device.load(std::memory_order_relaxed);
    if (d >= 0) HIP_TRY(hipSetDevice(d));
    return PEM_OK;
}
double g_angle_grid[NANG];
std::once_flag g_grid_once;
using pem::check_device;
using pem::fail;

// persistent grid of the fast kernel: workgroups of WPB waves, `per_cu` of them resident on every CU
// Balanced rounds: a persistent wave takes tiles me, me + nwaves, ...; with the largest grid that fits, a 1.25e6-sample shard is
// 9.54 tiles per wave -- ten rounds, the last one 54 % full and as long as a full one (a tile's duration is latency, not
// bandwidth: tools/tail_probe.py).  The smallest grid with the same number of rounds fills every round instead.
// Only where that costs little occupancy (>= 90 % of the slots stay in use): with two or three rounds the balanced grid is much
// smaller than the full one and the kernel loses more to the missing parallelism than it gains at the tail (plume_radii_kernel,
// 1e5 samples x 25 radii, 1.2 rounds: 560 us with the full grid, 655 us balanced; profiles/reconstruct_balanced_r02z.txt).
// PEM_BALANCED_GRID=0 restores the full grid everywhere.
size_t balanced_grid(size_t need, size_t cap) {
    static const bool balanced = getenv("PEM_BALANCED_GRID") ? atoi(getenv("PEM_BALANCED_GRID")) != 0 : true;
    if (need <= cap) return need;
    if (!balanced || cap == 0) return cap;
    const size_t rounds = (need + cap - 1) / cap;
    const size_t g = (need + rounds - 1) / rounds;
    return 10 * g >= 9 * cap ? g : cap;
}

// The grid of the R = 1 kernel, apart from the device query so that the host side can plan range launches with it
// (pem_persistent_grid).  Two regimes for the HBM-bound modes (profile written), measured interleaved on one box in the
// streaming regime (tools/grid_mode_ab.py, tools/launch_size_probe.py --walk; profiles/grid_modes_r03.txt):
//   * more than ONE_SHOT_ROUNDS rounds of work: a ONE-SHOT grid, one tile per wave, every workgroup handed to whichever CU has
//     room.  With the static walk (tile me, me + nwaves, ...) the launch ends when the slowest wave has done its share while
//     the others idle -- a tile's duration depends on where its wave sits; the dispatcher deals tiles out as slots free up
//     instead: 206-208 against 214-219 us for the 1.25e6-sample shard (9.5 rounds), 213-215 against 222-226 for half of it.
//     (A tile counter drawn from with atomics inside a persistent loop cost 36 registers, one wave per SIMD, 265 us; removed.)
//   * fewer rounds: the persistent loop with balanced rounds -- the next tile's inputs are in flight while a tile is worked on
//     and the tables are loaded once per wave, which is worth more than the dealing when a wave sees two or three tiles
//     (312 512 samples, 2.4 rounds: 214-217 against 224-225 us per 1.25e6).
constexpr long long ONE_SHOT_ROUNDS = 3;
long long persistent_grid(long long ntiles, long long cus, long long per_cu, bool memory_bound) {
    if (per_cu < 1) per_cu = 1;
    long long g = cus * per_cu;
    const long long need = (ntiles + WPB - 1) / WPB;
    if (g > need) g = need;
    if (!memory_bound) return g;
    // (the modes bound by instruction issue want every slot: balanced, the reduced-QoI launch takes 45.3 instead of 44.4 us and
    // the fused Monte-Carlo one 88 instead of 82 us; tools/grid_ab_probe.py)
    static const bool one_shot = getenv("PEM_ONE_SHOT") ? atoi(getenv("PEM_ONE_SHOT")) != 0 : true;
    if (one_shot && need > ONE_SHOT_ROUNDS * g) return need;
    return (long long)balanced_grid((size_t)need, (size_t)g);
}

int fast_grid(long long per_cu, long long ntiles, bool memory_bound, unsigned* grid) {
    int cus = 0;
    HIP_TRY(pem::device_cus(&cus));
    if (const char* e = getenv("PEM_WAVES_PER_CU")) {          // tuning/experiments only
        const long long v = atoll(e) / WPB;
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    *grid = (unsigned)persistent_grid(ntiles, cus, per_cu, memory_bound);
    if (const char* e = getenv("PEM_GRID_MULT")) {             // experiments: a grid of m x the resident slots (0 = one tile per wave)
        const long long m = atoll(e), need = (ntiles + WPB - 1) / WPB, cap = (long long)cus * (per_cu < 1 ? 1 : per_cu);
        *grid = (unsigned)((m <= 0 || m * cap > need) ? need : m * cap);
    }
    return PEM_OK;
}

template <int L, int JMODE, bool MC, int NQ = 0, bool PM = false>
size_t r1_lds_bytes(const PlumeIO& io) {
    size_t lds = (size_t)fast_lds_doubles<L, JMODE>() * 8;
    if (NQ > 0) lds += (size_t)count_lds_bytes<(NQ > 0 ? NQ : 1), PM>();
    if (JMODE == 3) lds += (size_t)io.n_cond * (io.n_ang | 1) * 32;
    if (JMODE == 0) lds += (size_t)QPOLY_DOUBLES * 8;
    if (MC) lds += (size_t)MC_LDS_DOUBLES * 8;
    return lds;
}

// Workgroups resident per CU: bounded by the LDS (160 KB) and by the registers this instantiation was compiled to
// (hipFuncGetAttributes; 512 per SIMD lane) -- the persistent loop must launch exactly as many as fit, a workgroup that
// waits for a slot turns the tile split into a two-pass schedule -- and capped at two waves per SIMD, which measured
// best for the HBM-bound modes, three for the profile-less ones: the fused Monte-Carlo kernel uses a third wave to
// hide Philox's quarter-rate multiplies whenever its register count allows one (<= 168).
template <int L, bool COUPLED, int JMODE, bool MC, int NQ = 0, bool PM = false>
int r1_per_cu(size_t lds, long long* per_cu) {
    auto kern = plume_r1_kernel<L, COUPLED, JMODE, MC, NQ, PM>;
    // the register count belongs to the code object (one architecture): once per process
    static std::once_flag once;
    static int by_regs = 0;
    static hipError_t err = hipSuccess;
    std::call_once(once, [&] {
        hipFuncAttributes fa;
        err = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
        if (err != hipSuccess) return;
        const int regs = fa.numRegs > 0 ? ((fa.numRegs + 7) & ~7) : 256;
        by_regs = (512 / regs) * 4 / WPB;
        if (getenv("PEM_DEBUG_OCCUPANCY"))
            fprintf(stderr, "pem: plume_r1_kernel<%d,%d,%d,%d>: %d registers -> %d workgroups per CU\n", L, (int)COUPLED, JMODE,
                    (int)MC, fa.numRegs, by_regs);
    });
    HIP_TRY(err);
    long long v = (long long)(160 * 1024 / lds);
    const int cap = (JMODE == 0 ? 12 : 8) / WPB;
    if (v > by_regs) v = by_regs;
    if (v > cap) v = cap;
    *per_cu = v;
    return PEM_OK;
}

// `grid_only`: report the grid the launch would use (the counting modes size their record buffer by it) and launch nothing
template <int L, bool COUPLED, int JMODE, bool MC = false, int NQ = 0, bool PM = false>
int launch_r1(const PlumeIO& io, const CoupledIO& cio, hipStream_t st, const McDesign& mc = McDesign{}, unsigned* grid_only = nullptr) {
    const size_t lds = r1_lds_bytes<L, JMODE, MC, NQ, PM>(io);
    const long long ntiles = (io.n + WAVE - 1) / WAVE;
    unsigned grid = 0;
    auto kern = plume_r1_kernel<L, COUPLED, JMODE, MC, NQ, PM>;
    if (lds > 64 * 1024) {
        static pem::LdsAttrOnce attr;
        HIP_TRY(attr.ensure(reinterpret_cast<const void*>(kern)));
    }
    long long per_cu = 0;
    if (int rc = r1_per_cu<L, COUPLED, JMODE, MC, NQ, PM>(lds, &per_cu)) return rc;
    // (the counting modes: a persistent grid of resident waves, each with its own region of the record buffer)
    if (int rc = fast_grid(per_cu, ntiles, JMODE == 1 || JMODE == 2, &grid)) return rc;
    if (grid_only) {
        *grid_only = grid;
        return PEM_OK;
    }
    if constexpr (MC) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * WPB), lds, st, io, cio, ntiles, mc);
    else hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * WPB), lds, st, io, cio, ntiles, NoDesign{});
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

template <bool COUPLED, int JMODE>
int dispatch_lanes(const PlumeIO& io, const CoupledIO& cio, hipStream_t st) {
    switch (g_lanes.load(std::memory_order_relaxed)) {
        case 2: return launch_r1<2, COUPLED, JMODE>(io, cio, st);
        case 8: return launch_r1<8, COUPLED, JMODE>(io, cio, st);
        default: return launch_r1<4, COUPLED, JMODE>(io, cio, st);
    }
}

template <int R>
int launch_rfew(size_t n, hipStream_t st, const PlumeIO& io, const RadiiSmallArg& ra) {
    constexpr int NW = rfew_waves<R>();
    constexpr size_t lds = (size_t)(TABLE_DOUBLES + NW * rfew_wave_doubles<R>()) * 8;
    static_assert(lds <= 160 * 1024, "the few-radii kernel's workgroup must fit the LDS");
    if (lds > 64 * 1024) {
        static pem::LdsAttrOnce attr;
        HIP_TRY(attr.ensure(reinterpret_cast<const void*>(plume_rfew_kernel<R>)));
    }
    int cus = 256;
    HIP_TRY(pem::device_cus(&cus));
    const size_t per_cu = (160 * 1024) / lds < 2 ? 1 : 2;          // persistent: workgroups resident per CU
    size_t grid = ((n + WAVE - 1) / WAVE + NW - 1) / NW;
    grid = balanced_grid(grid, (size_t)cus * per_cu);     // (grids of 2 / 4 x the slots or one tile per wave: within 2 %, r03o)
    hipLaunchKernelGGL(plume_rfew_kernel<R>, dim3((unsigned)grid), dim3(WAVE * NW), lds, st, io, ra);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
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

// Host <-> device movement of one chunk of a host-pointer entry point.  A chunk whose whole workspace footprint fits
// the pinned staging buffer is moved with ONE host-to-device and ONE device-to-host copy through a pinned mirror of the
// workspace layout (inputs are carved first, outputs after them, so each side is one contiguous range): a call with
// 15 input and 6 output arrays otherwise pays ~20 pageable-copy latencies (coupled, n = 1: 186 -> 61 us per call,
// tools/latency_probe.py).  The two sides decide separately: inputs are staged when they fit, outputs when the whole
// footprint does; what does not fit is copied array by array, where bandwidth is what matters.
constexpr size_t STAGE_BYTES = size_t(2) << 20;
struct Stage {
    unsigned char* pin = nullptr;
    bool tried = false;
    unsigned char* get() {
        if (!tried) {
            tried = true;
            void* p = nullptr;
            if (hipHostMalloc(&p, STAGE_BYTES, hipHostMallocPortable) == hipSuccess) pin = static_cast<unsigned char*>(p);
            else (void)hipGetLastError();
        }
        return pin;
    }
} g_stage;   // guarded by g_ws.mu

// Small calls skip the copies altogether: the kernels read their inputs from the pinned staging buffer and write their results
// to it over PCIe (hipHostMalloc memory is device-accessible at its host address and coherent), so a call is memcpy in, ONE launch,
// a stream synchronisation, memcpy out -- without the two copy-engine round trips: the calls amisc makes while it trains (a few to a
// few hundred samples) go from 33-35 to 27-29 us (cathode_coupling) and from 60-63 to 54-59 us (pem_v0_coupled), n = 1000: 113-148 ->
// 95-108 us (tools/latency_probe.py, interleaved; profiles/latency_r04.txt).  At 560 KB of footprint (BASELINE configs[0]: 1e4 cathode
// samples) the kernels' reads over the link cost what the copies saved: ZC_BYTES stays below that.  PEM_ZERO_COPY=0 switches it off.
constexpr size_t ZC_BYTES = size_t(256) << 10;
unsigned char* host_call_base(size_t footprint) {          // where a host-pointer call carves its arrays: g_ws.mu held, g_ws reserved
    static const bool on = !(getenv("PEM_ZERO_COPY") && atoi(getenv("PEM_ZERO_COPY")) == 0);
    if (on && footprint <= ZC_BYTES)
        if (unsigned char* pin = g_stage.get()) return pin;
    return static_cast<unsigned char*>(static_cast<void*>(g_ws.buf));
}

struct Mover {
    unsigned char* ws;
    unsigned char* pin;       // staging for the inputs, or nullptr: array-by-array copies
    unsigned char* pin_out;   // staging for the outputs (needs the whole footprint to fit), or nullptr
    size_t in_lo = ~size_t(0), in_hi = 0, out_lo = ~size_t(0), out_hi = 0;
    struct Out {
        void* host;
        size_t off, bytes;
    } outs[8];
    int nout = 0;
    // inputs are carved first: they end at `in_end`; the outputs end at `footprint`
    Mover(void* workspace, size_t in_end, size_t footprint)
        : ws(static_cast<unsigned char*>(workspace)),
          pin(in_end <= STAGE_BYTES ? g_stage.get() : nullptr),
          pin_out(footprint <= STAGE_BYTES ? pin : nullptr) {}
    int in(const void* host, void* dev, size_t bytes) {
        if (!pin) {
            HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, nullptr));
            return PEM_OK;
        }
        const size_t off = static_cast<unsigned char*>(dev) - ws;
        memcpy(pin + off, host, bytes);
        if (off < in_lo) in_lo = off;
        if (off + bytes > in_hi) in_hi = off + bytes;
        return PEM_OK;
    }
    int flush_in() {
        if (pin && pin != ws && in_hi > in_lo) HIP_TRY(hipMemcpyAsync(ws + in_lo, pin + in_lo, in_hi - in_lo, hipMemcpyHostToDevice, nullptr));
        return PEM_OK;
    }
    int out(void* host, const void* dev, size_t bytes) {
        if (!pin_out || nout == 8) {
            HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, nullptr));
            return PEM_OK;
        }
        const size_t off = static_cast<const unsigned char*>(dev) - ws;
        outs[nout++] = Out{host, off, bytes};
        if (off < out_lo) out_lo = off;
        if (off + bytes > out_hi) out_hi = off + bytes;
        return PEM_OK;
    }
    int finish() {
        if (pin_out && pin_out != ws && out_hi > out_lo)       // (ws == pin: the zero-copy form -- the kernels wrote there)
            HIP_TRY(hipMemcpyAsync(pin_out + out_lo, ws + out_lo, out_hi - out_lo, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        for (int i = 0; i < nout; ++i) memcpy(outs[i].host, pin_out + outs[i].off, outs[i].bytes);
        return PEM_OK;
    }
};
#define PEM_TRY(expr)              \
    do {                           \
        if (int rc_ = (expr)) return rc_; \
    } while (0)

}  // namespace

namespace pem {
char* error_buffer() {
    thread_local char buf[512] = "";
    return buf;
}
}  // namespace pem

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* pem_version(void) { return "hallthrusterpem_amd libpem_hip 0.1.0 (gfx950)"; }

const char* pem_last_error(void) { return pem::error_buffer(); }

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
    g_device.store(device, std::memory_order_relaxed);
    return PEM_OK;
}

int pem_synchronize(pem_stream_t stream) {
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PEM_OK;
}

int pem_set_lanes_per_sample(int lanes) {
    if (lanes == 0) lanes = 4;
    if (lanes == 2 || lanes == 4 || lanes == 8) g_lanes = lanes;
    return g_lanes;
}

int pem_persistent_grid(size_t n, int cus, int wg_per_cu, int memory_bound, size_t* workgroups, size_t* samples_per_round) {
    if (!workgroups || !samples_per_round) return fail(PEM_ERR_INVALID_ARG, "pem_persistent_grid: NULL result pointer");
    if (cus < 1 || wg_per_cu < 1) return fail(PEM_ERR_INVALID_ARG, "pem_persistent_grid: cus and wg_per_cu must be positive");
    const long long ntiles = (long long)((n + WAVE - 1) / WAVE);
    const long long g = persistent_grid(ntiles, cus, wg_per_cu, memory_bound != 0);
    const long long resident = g < (long long)cus * wg_per_cu ? g : (long long)cus * wg_per_cu;   // a one-shot grid is longer than that
    *workgroups = (size_t)g;
    *samples_per_round = (size_t)resident * WPB * WAVE;
    return PEM_OK;
}

int pem_coupled_occupancy(int profile_mode, int* cus, int* wg_per_cu) {
    if (!cus || !wg_per_cu) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_occupancy: NULL result pointer");
    if (int rc = check_device()) return rc;
    HIP_TRY(pem::device_cus(cus));
    PlumeIO io{};
    long long v = 0;
    int rc;
    // the instantiations pem_coupled_f64_dev / pem_coupled_mixed_dev launch at the current lanes-per-sample setting
#define PEM_OCC(L_)                                                                                      \
    (profile_mode == 0   ? r1_per_cu<L_, true, 0, false>(r1_lds_bytes<L_, 0, false>(io), &v)               \
     : profile_mode == 1 ? r1_per_cu<L_, true, 1, false>(r1_lds_bytes<L_, 1, false>(io), &v)               \
                         : r1_per_cu<L_, true, 2, false>(r1_lds_bytes<L_, 2, false>(io), &v))
    if (profile_mode < 0 || profile_mode > 2) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_occupancy: profile_mode must be 0, 1 or 2");
    switch (g_lanes.load(std::memory_order_relaxed)) {
        case 2: rc = PEM_OCC(2); break;
        case 8: rc = PEM_OCC(8); break;
        default: rc = PEM_OCC(4); break;
    }
#undef PEM_OCC
    if (rc) return rc;
    if (const char* e = getenv("PEM_WAVES_PER_CU")) {          // as fast_grid
        const long long w = atoll(e) / WPB;
        if (w >= 1 && w < v) v = w;
    }
    *wg_per_cu = (int)v;
    return PEM_OK;
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
    if (n_radii == 1 && aligned16(j_ion)) return dispatch_lanes<false, 1>(io, CoupledIO{}, st);

    if (n_radii >= 2 && n_radii <= RADII_SMALL && aligned16(j_ion) && !getenv("PEM_RADII_GENERAL")) {
        // few radii: eight samples per wave in flight, Gaussians by recurrence, profile stored from the angle loop
        RadiiSmallArg ra;
        for (int r = 0; r < RADII_SMALL; ++r) ra.r[r] = r < n_radii ? radii[r] : 1.0;
        int rc = PEM_OK;
        switch (n_radii) {
            case 2: rc = launch_rfew<2>(n, st, io, ra); break;
            case 3: rc = launch_rfew<3>(n, st, io, ra); break;
            case 4: rc = launch_rfew<4>(n, st, io, ra); break;
            case 5: rc = launch_rfew<5>(n, st, io, ra); break;
            case 6: rc = launch_rfew<6>(n, st, io, ra); break;
            case 7: rc = launch_rfew<7>(n, st, io, ra); break;
            default: rc = launch_rfew<8>(n, st, io, ra); break;
        }
        if (rc) return rc;
        HIP_TRY(hipGetLastError());
        return PEM_OK;
    }
    static const bool use_rmid = getenv("PEM_RADII_MID") ? atoi(getenv("PEM_RADII_MID")) != 0 : true;
    // (read per call: tests walk through the instantiations)
    int rmid_min = getenv("PEM_RMID_MIN") ? atoi(getenv("PEM_RMID_MIN")) : 13;
    if (rmid_min < WAVE / RMID_G_MAX + 1) rmid_min = WAVE / RMID_G_MAX + 1;
    if (use_rmid && n_radii >= rmid_min && n_radii > RADII_SMALL && n_radii <= RMID_MAX) {
        // S samples in flight per wave in P passes, rows staged in LDS, line-aligned 16-byte stores (plume_rmid_kernel).  The pair
        // (S, P) of the instantiated ones that fills the most lane slots, S R / (64 P): 25 radii -> five samples in two passes (125
        // of 128; round 3: two samples in one, 50 of 64), 33 -> three in two (99 of 128; one sample before: 33 of 64).
        // From 13 radii on (round 4, with the 10-KB tile: 13 / 14 / 15 / 16 radii 3.02 -> 3.35, 3.30 -> 3.73, 3.35 -> 3.57, 4.26 -> 4.38 TB/s
        // against the wave-per-sample kernel below, interleaved; profiles/radii_mid_r04.txt); at 11 and 12 radii (PEM_RMID_MIN=11) it
        // works and gains nothing.
        // Instantiated: one pass.  Two- and three-pass packings (-DPEM_RMID_MULTIPASS=1) fill 86-98 % of the lane slots where one pass
        // fills 52-80 %, and measured SLOWER at every radius count (25 radii: 3.25 against 3.69 TB/s, 33: 3.59 against 4.12, 44: 3.58
        // against 4.33; profiles/radii_mid_r04.txt): more samples share the 8 KB of staged rows, so a sample's runs get shorter (200
        // doubles instead of 500 at 25 radii) and the head / body / tail of a run and the two syncs around it are paid 12 times per
        // sample instead of 5 -- the kernel's bound is that phase structure, not idle lanes.  What separates 32 / 40 / 48 / 64 radii
        // (4.6-4.9 TB/s) from their neighbours (3.7-4.3) is the alignment of a sample's rows to 128-byte lines, not the lane count.
#if defined(PEM_RMID_MULTIPASS) && PEM_RMID_MULTIPASS
        static const int combos[][2] = {{1, 1}, {2, 1}, {3, 1}, {4, 1}, {5, 1}, {3, 2}, {5, 2}, {6, 2}, {7, 2}, {4, 3}, {6, 3}, {7, 3}};
#else
        static const int combos[][2] = {{1, 1}, {2, 1}, {3, 1}, {4, 1}, {5, 1}};
#endif
        int S = 1, P = 1;
        double best = 0.0;
        for (const auto& c : combos) {
            if (c[0] * n_radii > WAVE * c[1] || (c[1] > 1 && c[0] * n_radii <= WAVE * (c[1] - 1))) continue;   // the pairs fill 64 (P - 1) + 1 .. 64 P slots
            if ((RMID_TILE / c[0] - 2) / n_radii < 1) continue;                                               // a staged row per sample must fit
            const double eff = (double)(c[0] * n_radii) / (WAVE * c[1]) - 0.02 * (c[1] - 1);                  // (a pass more has to pay for itself)
            if (eff > best) {
                best = eff;
                S = c[0];
                P = c[1];
            }
        }
        if (const char* e = getenv("PEM_RMID_SP")) {                                   // tests / experiments: "S,P" of an instantiated pair
            int es = 0, ep = 0;
            if (sscanf(e, "%d,%d", &es, &ep) == 2 && es * n_radii <= WAVE * ep && (RMID_TILE / es - 2) / n_radii >= 1)
                for (const auto& c : combos)
                    if (c[0] == es && c[1] == ep) {
                        S = es;
                        P = ep;
                    }
        }
        RadiiMidArg ra;
        for (int r = 0; r < RMID_MAX; ++r) ra.r[r] = r < n_radii ? radii[r] : 1.0;
        int ts = WAVE;                             // samples per wave tile: fewer when the batch is small
        while (ts > 8 && (n + ts - 1) / ts < 256 * 32) ts >>= 1;
        ts = ts / S * S;                           // whole groups only
        if (ts < 2 * S) ts = 2 * S <= WAVE ? 2 * S : S;
        if (const char* e = getenv("PEM_RMID_TS")) ts = atoi(e);                      // experiments
        if (ts < 1 || ts > WAVE) return fail(PEM_ERR_INVALID_ARG, "pem_plume: PEM_RMID_TS must be 1..64");
        const size_t ntiles = (n + ts - 1) / ts;
        int cus = 256;
        HIP_TRY(pem::device_cus(&cus));
        size_t blocks = (ntiles + BLOCK / WAVE - 1) / (BLOCK / WAVE);
#define PEM_RMID_LAUNCH(S_, P_)                                                                                     \
    do {                                                                                                            \
        const size_t lds = (size_t)(BLOCK / WAVE) * rmid_wave_doubles<S_>() * 8;                                    \
        size_t per_cu = (160 * 1024) / lds;                                                                         \
        if (per_cu > (size_t)rmid_waves_per_simd<S_, P_>()) per_cu = rmid_waves_per_simd<S_, P_>();                 \
        static pem::LdsAttrOnce attr;                          /* (four and five samples per wave: more than 64 KB) */ \
        HIP_TRY(attr.ensure(reinterpret_cast<const void*>(plume_rmid_kernel<S_, P_>)));                             \
        blocks = balanced_grid(blocks, (size_t)cus * per_cu);                                                       \
        hipLaunchKernelGGL((plume_rmid_kernel<S_, P_>), dim3((unsigned)blocks), dim3(BLOCK), lds, st, io, ra, n_radii, ts); \
    } while (0)
        switch (S * 10 + P) {
            case 11: PEM_RMID_LAUNCH(1, 1); break;
            case 21: PEM_RMID_LAUNCH(2, 1); break;
            case 31: PEM_RMID_LAUNCH(3, 1); break;
            case 41: PEM_RMID_LAUNCH(4, 1); break;
#if defined(PEM_RMID_MULTIPASS) && PEM_RMID_MULTIPASS
            case 32: PEM_RMID_LAUNCH(3, 2); break;
            case 52: PEM_RMID_LAUNCH(5, 2); break;
            case 62: PEM_RMID_LAUNCH(6, 2); break;
            case 72: PEM_RMID_LAUNCH(7, 2); break;
            case 43: PEM_RMID_LAUNCH(4, 3); break;
            case 63: PEM_RMID_LAUNCH(6, 3); break;
            case 73: PEM_RMID_LAUNCH(7, 3); break;
#endif
            default: PEM_RMID_LAUNCH(5, 1); break;
        }
#undef PEM_RMID_LAUNCH
        HIP_TRY(hipGetLastError());
        return PEM_OK;
    }
    if (n_radii >= 2 && n_radii <= RADII_MAX) {
        // wave per sample, coalesced (91, R) blocks, literal Gaussians (per 1e5..1e6 samples, tools/radii_probe.py: R = 25:
        // 7415 -> 614 us, R = 5: 2089 -> 795 us, R = 3: 1262 -> 940 us, R = 2: 1183 -> 1314 us)
        RadiiArg ra;
        for (int r = 0; r < RADII_MAX; ++r) ra.r[r] = r < n_radii ? radii[r] : 1.0;
        int ts = WAVE;                             // samples per wave tile: fewer when the batch is small
        while (ts > 4 && (n + ts - 1) / ts < 256 * 20) ts >>= 1;
        const size_t ntiles = (n + ts - 1) / ts;
        size_t blocks = (ntiles + BLOCK / WAVE - 1) / (BLOCK / WAVE);
        blocks = balanced_grid(blocks, 256 * 5);   // persistent: 31 KB of LDS per workgroup, five per CU
        hipLaunchKernelGGL(plume_radii_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, st, io, ra, n_radii, ts);
        HIP_TRY(hipGetLastError());
        return PEM_OK;
    }
    // lane-per-sample kernel (more than RADII_MAX radii; one radius with an unaligned j_ion): the radii go to the device
    // through a small stream-ordered allocation, and -- `radii` being the caller's host memory -- this one path waits
    // for the stream before it returns
    double* d_radii = nullptr;
    HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&d_radii), sizeof(double) * n_radii, st));
    HIP_TRY(hipMemcpyAsync(d_radii, radii, sizeof(double) * n_radii, hipMemcpyHostToDevice, st));
    const size_t blocks = (n + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(plume_generic_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, st, io, d_radii, n_radii);
    hipError_t le = hipGetLastError();
    HIP_TRY(hipFreeAsync(d_radii, st));
    HIP_TRY(le);
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
    return j_ion ? dispatch_lanes<true, 1>(io, cio, st) : dispatch_lanes<true, 0>(io, cio, st);
}

// ---- coupled, inputs tile-interleaved: [ceil(n / 64)][15][64], one contiguous 7.5 KB block per 64-sample tile ----
int pem_coupled_tiled_f64_dev(size_t n, double torr2pa, double radius, const double* x_tiled, double* V_cc, double* I_B0,
                              double* T, double* j_ion, double* div_angle, double* T_c, uint8_t* invalid, pem_stream_t stream) {
    if (n == 0) return PEM_OK;
    if (!x_tiled || !V_cc || !div_angle || !T_c) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_tiled: NULL array");
    if (j_ion && !aligned16(j_ion)) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_tiled: j_ion must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double* x = x_tiled;   // rows in the order of pem_coupled_f64_dev's arguments: P_b V_a T_e V_vac Pstar P_T mdot_a a_1 c0..c5 sigma_cex
    PlumeIO io{(long long)n, torr2pa, radius, x, x + 8 * WAVE, x + 9 * WAVE, x + 10 * WAVE, x + 11 * WAVE, x + 12 * WAVE, x + 13 * WAVE,
               x + 14 * WAVE, nullptr, nullptr, j_ion, div_angle, T_c, invalid};
    io.in_tile_stride = 15 * WAVE;
    CoupledIO cio{x + 1 * WAVE, x + 2 * WAVE, x + 3 * WAVE, x + 4 * WAVE, x + 5 * WAVE, x + 6 * WAVE, x + 7 * WAVE, V_cc, I_B0, T};
    return j_ion ? dispatch_lanes<true, 1>(io, cio, st) : dispatch_lanes<true, 0>(io, cio, st);
}

// ---- coupled, fused Monte-Carlo: inputs generated from the counter-based design inside the kernel ------------
int pem_coupled_mc_f64_dev(size_t n, uint64_t first_index, uint64_t seed, uint32_t stream_id, int swap_dim,
                           const int32_t* kind, const double* a, const double* b, double torr2pa, double radius,
                           double* x_out, size_t ld,
                           double* V_cc, double* I_B0, double* T, double* j_ion, double* div_angle, double* T_c,
                           uint8_t* invalid, pem_stream_t stream) {
    if (n == 0) return PEM_OK;
    if (!kind || !a || !b || !V_cc || !div_angle || !T_c) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc: NULL array");
    if (j_ion && !aligned16(j_ion)) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc: j_ion must be 16-byte aligned");
    if (x_out && ld < n) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc: leading dimension smaller than n");
    if (swap_dim < -2 || swap_dim >= 15) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc: swap_dim out of range");
    if (int rc = check_device()) return rc;
    McDesign mc{};
    mc.seed = seed;
    mc.first = first_index;
    mc.stream = stream_id;
    mc.swap_dim = swap_dim;
    for (int d = 0; d < 15; ++d) {
        if (kind[d] < PEM_DIST_UNIFORM || kind[d] > PEM_DIST_NORMAL)
            return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc: unknown distribution kind %d for input %d", kind[d], d);
        mc.kind[d] = kind[d];
        mc.a[d] = a[d];
        mc.b[d] = b[d];
    }
    mc.x_out = x_out;
    mc.ld = ld;
    PlumeIO io{(long long)n, torr2pa, radius, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, j_ion, div_angle, T_c, invalid, nullptr};
    CoupledIO cio{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, V_cc, I_B0, T};
    hipStream_t st = static_cast<hipStream_t>(stream);
    return j_ion ? launch_r1<4, true, 1, true>(io, cio, st, mc) : launch_r1<4, true, 0, true>(io, cio, st, mc);
}

// ---- coupled, fused Monte-Carlo + campaign statistics: see csrc/pem_qfused.h, pem_coupled_mc_stats_f64_dev below --------------
}  // extern "C"

namespace {

int mc_design_of(const pem::McLaunch& a, McDesign* mc) {
    mc->seed = a.seed;
    mc->first = a.first_index;
    mc->stream = a.stream_id;
    mc->swap_dim = -1;
    for (int d = 0; d < 15; ++d) {
        if (a.kind[d] < PEM_DIST_UNIFORM || a.kind[d] > PEM_DIST_NORMAL)
            return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc: unknown distribution kind %d for input %d", a.kind[d], d);
        mc->kind[d] = a.kind[d];
        mc->a[d] = a.a[d];
        mc->b[d] = a.b[d];
    }
    mc->x_out = a.x_out;
    mc->ld = a.ld;
    return PEM_OK;
}

// the counting launch for nq quantiles: instantiated for 3, 5 and 6 brackets per angle (fewer are padded with empty ones)
template <int JMODE>
int launch_count(const PlumeIO& io, const CoupledIO& cio, const McDesign& mc, int nq, hipStream_t st, unsigned* grid_only) {
    // (the premask rides with up to five brackets: six leave no LDS for its thresholds beside two workgroups per CU)
    if (io.q.premask && nq <= 5) return launch_r1<4, true, JMODE, true, 5, true>(io, cio, st, mc, grid_only);
    if (nq <= 3) return launch_r1<4, true, JMODE, true, 3>(io, cio, st, mc, grid_only);
    if (nq <= 5) return launch_r1<4, true, JMODE, true, 5>(io, cio, st, mc, grid_only);
    return launch_r1<4, true, JMODE, true, 6>(io, cio, st, mc, grid_only);
}

}  // namespace

namespace pem {

int launch_coupled_mc(const McLaunch& a, hipStream_t st) {
    if (a.n == 0) return PEM_OK;
    McDesign mc{};
    if (int rc = mc_design_of(a, &mc)) return rc;
    PlumeIO io{(long long)a.n, a.torr2pa, a.radius, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, a.j_ion, a.div_angle, a.T_c, a.invalid, nullptr};
    CoupledIO cio{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, a.V_cc, a.I_B0, a.T};
    return a.j_ion ? launch_r1<4, true, 1, true>(io, cio, st, mc) : launch_r1<4, true, 0, true>(io, cio, st, mc);
}

static int count_launch(const McLaunch& a, const CountIO& c, bool store_profile, hipStream_t st, unsigned* grid_only) {
    McDesign mc{};
    if (int rc = mc_design_of(a, &mc)) return rc;
    PlumeIO io{(long long)a.n, a.torr2pa, a.radius, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, a.j_ion, a.div_angle, a.T_c, a.invalid, nullptr};
    io.q = c;
    CoupledIO cio{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, a.V_cc, a.I_B0, a.T};
    return store_profile ? launch_count<4>(io, cio, mc, c.nq, st, grid_only) : launch_count<5>(io, cio, mc, c.nq, st, grid_only);
}

int coupled_count_waves(size_t n, int nq, bool store_profile, unsigned* waves) {
    McLaunch a{};
    a.n = n;
    for (int d = 0; d < 15; ++d) a.kind[d] = PEM_DIST_UNIFORM;
    CountIO c{};
    c.nq = nq;
    unsigned grid = 0;
    if (int rc = count_launch(a, c, store_profile, nullptr, &grid)) return rc;
    *waves = grid * WPB;
    return PEM_OK;
}

int launch_coupled_mc_count(const McLaunch& a, const CountIO& c, bool store_profile, hipStream_t st) {
    if (a.n == 0) return PEM_OK;
    if (c.nq < 1 || c.nq > 6 || !c.br || !c.below || !c.rec || !c.rec_count || !c.flags || c.cap == 0)
        return fail(PEM_ERR_INVALID_ARG, "counting launch: incomplete arguments");
    if (store_profile && !a.j_ion) return fail(PEM_ERR_INVALID_ARG, "counting launch: no profile array to store into");
    return count_launch(a, c, store_profile, st, nullptr);
}

}  // namespace pem

extern "C" {

// ---- coupled, fused Monte-Carlo with the percentiles of the profile counted on the way (round 4) --------------------------------
namespace {
// The scalar QoIs' percentiles of a campaign, selected on a second host thread and stream while the calling thread takes the
// profile through its pilot, its counting launch and the passes over its records (csrc/pem_quantile.hip keeps a second set of
// buffers for it).  Two stages, each released on the host (a promise) and ordered on the device (an event): the scalars of the
// pilot's samples exist once the pilot evaluation is under way -- the worker brackets the wanted ranks from them while the
// counting launch runs -- and all of them once the counting launch (or, after a decline, the plain launch) is.
struct ScalarJob {
    struct Stage {
        std::promise<int> go;                  // 1: `ev` marks the launch; 0: give up (an error on the calling thread)
        std::future<int> gone;
        bool signalled = false;
        hipEvent_t ev = nullptr;
        Stage() : gone(go.get_future()) {}
        void signal(int v) {
            if (!signalled) {
                signalled = true;
                go.set_value(v);
            }
        }
        int launched(hipStream_t st) {         // after the launch has been enqueued on `st`
            if (signalled) return PEM_OK;
            HIP_TRY(hipEventRecord(ev, st));
            signal(1);
            return PEM_OK;
        }
        // the worker: block until the launch is under way, then make `side` wait for it
        int await(hipStream_t side) {
            if (gone.get() != 1) return fail(PEM_ERR_HIP, "pem_coupled_mc_stats (scalar selection): given up");
            HIP_TRY(hipStreamWaitEvent(side, ev, 0));
            return PEM_OK;
        }
    };
    Stage pilot, full;
    // The other direction: the counting launch must not START while the side selection's subsample passes are still running -- their
    // histograms take most of a CU's LDS, a workgroup of the persistent counting grid that finds no room waits for a whole pass of its
    // neighbours, and the launch takes 4 ms instead of 2.3 (seen in two calls of seven under the profiler, whose host threads are
    // slow).  The worker marks the end of those passes (`side`: released by the worker, awaited by the calling thread).
    Stage side;
    std::thread worker;
    std::function<void()> body;                // what the worker runs
    bool started = false;
    int rc = PEM_OK;
    std::string error;
    // Events and thread are made AFTER the pilot evaluation has been enqueued (McProducer::pilot): their 50-100 us of host time then
    // pass while the GPU works instead of in front of the call's first kernel.
    int start() {
        if (started) return PEM_OK;
        started = true;
        HIP_TRY(hipEventCreateWithFlags(&pilot.ev, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&full.ev, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&side.ev, hipEventDisableTiming));
        try {
            worker = std::thread(body);
        } catch (const std::exception& e) {    // (no thread to be had: nothing may leave a C entry point but its return code)
            return fail(PEM_ERR_HIP, "pem_coupled_mc_stats: could not start the scalar selection's thread: %s", e.what());
        }
        return PEM_OK;
    }
    int join() {
        pilot.signal(0);
        full.signal(0);
        if (worker.joinable()) worker.join();
        for (Stage* s : {&pilot, &full, &side}) {
            if (s->ev) (void)hipEventDestroy(s->ev);
            s->ev = nullptr;
        }
        return rc;
    }
    ~ScalarJob() { (void)join(); }
};

// a stream of the library's own per device (created once)
int side_stream(hipStream_t* out) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(PEM_ERR_INVALID_ARG, "device index out of range");
    std::lock_guard<std::mutex> lock(mu);
    // (a higher stream priority for the side work was measured and changes nothing: 3.85-3.92 ms per 1e7-sample campaign either way)
    if (!streams[dev]) HIP_TRY(hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking));
    *out = streams[dev];
    return PEM_OK;
}

struct McProducer : pem::FusedProducer {
    pem::McLaunch a;
    bool store_profile;
    bool counted = false;                          // the counting launch is under way: every output but the percentiles gets written
    ScalarJob* job = nullptr;
    int pilot(size_t rows, double* dst, hipStream_t st) override {
        pem::McLaunch p = a;                       // samples 0 .. rows-1 of the same design; their profile rows go to dst
        p.n = rows;
        p.j_ion = dst;
        if (int rc = pem::launch_coupled_mc(p, st)) return rc;
        if (!job) return PEM_OK;
        if (int rc = job->start()) return rc;
        return job->pilot.launched(st);            // (their scalars too: the side selection's subsample)
    }
    int waves(int nq, unsigned* w) override { return pem::coupled_count_waves(a.n, nq, store_profile, w); }
    int count(const pem::CountIO& io, hipStream_t st) override {
        if (job && job->worker.joinable() && job->side.gone.get() == 1) HIP_TRY(hipStreamWaitEvent(st, job->side.ev, 0));   // (see ScalarJob::side)
        if (int rc = pem::launch_coupled_mc_count(a, io, store_profile, st)) return rc;
        counted = true;
        return job ? job->full.launched(st) : PEM_OK;    // (the side selection's passes over all samples may follow this launch)
    }
};
}  // namespace

int pem_coupled_mc_stats_f64_dev(size_t n, uint64_t first_index, uint64_t seed, uint32_t stream_id, const int32_t* kind, const double* a,
                                 const double* b, double torr2pa, double radius, double* x_out, size_t ld, double* V_cc, double* I_B0,
                                 double* T, double* j_ion, double* pilot_rows, double* div_angle, double* T_c, uint8_t* invalid, int nq,
                                 const uint64_t* rank_prev, const uint64_t* rank_next, const double* gamma, double* q_out, double* q_scalars,
                                 int* fused_ok, int q25, int q75, double iqr_factor, uint8_t* row_certain, uint8_t* row_uncertain,
                                 int* premask_ok, pem_stream_t stream) {
    if (!kind || !a || !b || !V_cc || !div_angle || !T_c || !rank_prev || !rank_next || !gamma || !q_out || !fused_ok)
        return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc_stats: NULL array");
    if (!j_ion && !pilot_rows) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc_stats: without a profile array, room for the pilot rows is needed");
    if (j_ion && !aligned16(j_ion)) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc_stats: j_ion must be 16-byte aligned");
    if (pilot_rows && !aligned16(pilot_rows)) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc_stats: pilot_rows must be 16-byte aligned");
    if (x_out && ld < n) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc_stats: leading dimension smaller than n");
    if (nq < 1 || nq > PEM_QUANTILE_MAX_Q) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc_stats: 1 <= nq <= %d", PEM_QUANTILE_MAX_Q);
    if (n < PEM_MC_STATS_MIN_N) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc_stats: at least %d samples", PEM_MC_STATS_MIN_N);
    // q_scalars: V_cc, div_angle, T_c must then be rows 0, 1, 2 of one [3][row stride >= n] array (the reduced-QoI tensor)
    const ptrdiff_t qstride = div_angle - V_cc;
    if (q_scalars && (qstride < (ptrdiff_t)n || T_c - div_angle != qstride))
        return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mc_stats: q_scalars needs V_cc, div_angle, T_c as equally spaced rows of one array");
    if (int rc = check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ScalarJob job;
    McProducer prod;
    if (q_scalars) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        hipStream_t side = nullptr;
        if (int rc = side_stream(&side)) return rc;
        job.body = [&job, dev, side, n, nq, V_cc, qstride, rank_prev, rank_next, gamma, q_scalars]() {
            struct Release {                                             // (whatever happens: the calling thread is not left waiting)
                ScalarJob& j;
                ~Release() { j.side.signal(0); }
            } release{job};
            auto note = [&job](int rc) {
                job.rc = rc;
                if (rc != PEM_OK) job.error = pem_last_error();          // (the message lives in this thread's buffer)
                return rc;
            };
            if (hipSetDevice(dev) != hipSuccess) {
                (void)note(fail(PEM_ERR_HIP, "pem_coupled_mc_stats (scalar selection): hipSetDevice failed"));
                return;
            }
            if (note(job.pilot.await(side))) return;
            pem::SidePlan plan;
            plan.ctx = &job;
            plan.before_full = [](void* ctx, hipStream_t st) {
                ScalarJob* j = static_cast<ScalarJob*>(ctx);
                if (j->side.launched(st)) j->side.signal(0);             // the subsample's passes end here (on `st`, the side stream)
                return j->full.await(st);
            };
            (void)note(pem::quantiles_side(n, 3, V_cc, 1, (size_t)qstride, nq, rank_prev, rank_next, gamma, q_scalars, side, &plan));
        };
        prod.job = &job;
    }
    prod.a.n = n;
    prod.a.first_index = first_index;
    prod.a.seed = seed;
    prod.a.stream_id = stream_id;
    for (int d = 0; d < 15; ++d) {
        prod.a.kind[d] = kind[d];
        prod.a.a[d] = a[d];
        prod.a.b[d] = b[d];
    }
    prod.a.torr2pa = torr2pa;
    prod.a.radius = radius;
    prod.a.x_out = x_out;
    prod.a.ld = ld;
    prod.a.V_cc = V_cc;
    prod.a.I_B0 = I_B0;
    prod.a.T = T;
    prod.a.j_ion = j_ion;
    prod.a.div_angle = div_angle;
    prod.a.T_c = T_c;
    prod.a.invalid = invalid;
    prod.store_profile = j_ion != nullptr;
    if (row_certain && row_uncertain && premask_ok) {
        prod.pm_q25 = q25;
        prod.pm_q75 = q75;
        prod.pm_factor = iqr_factor;
        prod.pm_certain = row_certain;
        prod.pm_uncertain = row_uncertain;
    }
    if (premask_ok) *premask_ok = 0;
    *fused_ok = 0;
    // with a profile array the pilot rows are its own first rows (the counting launch writes the same values there again)
    if (int rc = pem::quantiles_fused(n, NANG, nq, rank_prev, rank_next, gamma, j_ion ? j_ion : pilot_rows, prod, q_out, fused_ok, st))
        return rc;                                 // (~ScalarJob tells the worker to give up and joins it)
    if (premask_ok) *premask_ok = (*fused_ok && prod.pm_done) ? 1 : 0;
    if (!*fused_ok && !prod.counted) {
        // declined before the counting launch (the fused form refused the call's shape), with nothing but the pilot's samples evaluated:
        // the plain launch makes every output complete.  (Declined AFTER it -- unfit brackets, a rank outside its bracket, record
        // overflow, a non-finite value -- the counting launch has written every output already; only the percentiles are missing.)
        if (int rc = pem::launch_coupled_mc(prod.a, st)) return rc;
        if (q_scalars) {
            if (int rc = job.start()) return rc;
            if (int rc = job.pilot.launched(st)) return rc;               // (no-ops for a stage that has been released)
            if (int rc = job.full.launched(st)) return rc;
        }
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (q_scalars) {                               // (both stages released by now on every path that comes here)
        if (int rc = job.join()) return fail(rc, "%s", job.error.c_str());
    }
    return PEM_OK;
}

// ---- coupled + likelihood fused: the profile never leaves the chip ------------------------------------------------
int pem_coupled_loglik_f64_dev(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a,
                               const double* T_e, const double* V_vac, const double* Pstar, const double* P_T,
                               const double* mdot_a, const double* a_1, const double* c0, const double* c1, const double* c2,
                               const double* c3, const double* c4, const double* c5, const double* sigma_cex, int n_cond,
                               int n_ang, const int32_t* kidx, const double* weight, const double* y, const double* inv_std,
                               double* V_cc, double* div_angle, double* T_c, double* loglik, uint8_t* invalid,
                               pem_stream_t stream) {
    if (n_cond < 1 || n_ang < 1 || (long long)n_cond * (n_ang | 1) > PEM_FUSED_LOGLIK_MAX_MEASUREMENTS)
        return fail(PEM_ERR_INVALID_ARG, "pem_coupled_loglik: 1 <= n_cond * (n_ang | 1) <= %d", PEM_FUSED_LOGLIK_MAX_MEASUREMENTS);
    if (n == 0) return PEM_OK;
    if (!P_b || !V_a || !T_e || !V_vac || !Pstar || !P_T || !mdot_a || !a_1 || !c0 || !c1 || !c2 || !c3 || !c4 || !c5 ||
        !sigma_cex || !kidx || !weight || !y || !inv_std || !V_cc || !div_angle || !T_c || !loglik)
        return fail(PEM_ERR_INVALID_ARG, "pem_coupled_loglik: NULL array");
    if (int rc = check_device()) return rc;
    PlumeIO io{(long long)n, torr2pa, radius, P_b, c0, c1, c2, c3, c4, c5, sigma_cex, nullptr, nullptr, nullptr, div_angle, T_c, invalid, nullptr,
               kidx, weight, y, inv_std, loglik, n_cond, n_ang};
    CoupledIO cio{V_a, T_e, V_vac, Pstar, P_T, mdot_a, a_1, V_cc, nullptr, nullptr};
    return launch_r1<4, true, 3>(io, cio, static_cast<hipStream_t>(stream));
}

// ---- coupled, mixed precision: fp64 arithmetic, the 91-point profile stored as fp32 -----------------
int pem_coupled_mixed_dev(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a, const double* T_e,
                          const double* V_vac, const double* Pstar, const double* P_T, const double* mdot_a,
                          const double* a_1, const double* c0, const double* c1, const double* c2, const double* c3,
                          const double* c4, const double* c5, const double* sigma_cex, double* V_cc, double* I_B0,
                          double* T, float* j_ion_f32, double* div_angle, double* T_c, uint8_t* invalid,
                          pem_stream_t stream) {
    if (n == 0) return PEM_OK;
    if (!P_b || !V_a || !T_e || !V_vac || !Pstar || !P_T || !mdot_a || !a_1 || !c0 || !c1 || !c2 || !c3 || !c4 || !c5 ||
        !sigma_cex || !V_cc || !div_angle || !T_c || !j_ion_f32)
        return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mixed: NULL array");
    if (!aligned16(j_ion_f32)) return fail(PEM_ERR_INVALID_ARG, "pem_coupled_mixed: j_ion_f32 must be 16-byte aligned");
    if (int rc = check_device()) return rc;
    PlumeIO io{(long long)n, torr2pa, radius, P_b, c0, c1, c2, c3, c4, c5, sigma_cex, nullptr, nullptr, nullptr, div_angle, T_c, invalid, j_ion_f32};
    CoupledIO cio{V_a, T_e, V_vac, Pstar, P_T, mdot_a, a_1, V_cc, I_B0, T};
    return dispatch_lanes<true, 2>(io, cio, static_cast<hipStream_t>(stream));
}

// ---- thruster profile + filters ----------------------------------------------------------------------
int pem_thruster_uion_f64_dev(size_t n, const double* v_exh, double z0, double z1, int ncells, double* z, double* u_ion,
                              pem_stream_t stream) {
    if (ncells < 2) return fail(PEM_ERR_INVALID_ARG, "pem_thruster_uion: need at least 2 grid points");
    if (n == 0) return PEM_OK;
    if (!v_exh || !u_ion) return fail(PEM_ERR_INVALID_ARG, "pem_thruster_uion: NULL array");
    if (int rc = check_device()) return rc;
    size_t blocks = (n * (size_t)ncells + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(thruster_uion_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                       (long long)n, v_exh, z0, z1, ncells, z, u_ion);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

int pem_thruster_filter_f64_dev(size_t n, int ncells, const double* u_ion, const double* z, double shock_threshold,
                                int use_shock, const double* T, const double* I_B0, uint8_t* flags, pem_stream_t stream) {
    if (n == 0) return PEM_OK;
    if (!flags) return fail(PEM_ERR_INVALID_ARG, "pem_thruster_filter: NULL flags");
    if (use_shock && (!u_ion || !z || ncells < 1)) return fail(PEM_ERR_INVALID_ARG, "pem_thruster_filter: shock filter needs u_ion and z");
    if (int rc = check_device()) return rc;
    size_t blocks = (n * 64 + BLOCK - 1) / BLOCK;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(thruster_filter_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, static_cast<hipStream_t>(stream),
                       (long long)n, ncells, u_ion, z, shock_threshold, use_shock, T, I_B0, flags);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

// =============================================================================================
// host-pointer entry points: stage through the device workspace in chunks
// =============================================================================================
int pem_cathode_f64(size_t n, const double* P_b, const double* V_a, const double* T_e, const double* V_vac,
                    const double* Pstar, const double* P_T, double torr2pa, double* V_cc) {
    if (n == 0) return PEM_OK;
    if (!P_b || !V_a || !T_e || !V_vac || !Pstar || !P_T || !V_cc) return fail(PEM_ERR_INVALID_ARG, "pem_cathode: NULL array");
    if (int rc = check_device()) return rc;
    if (int rc = use_default_device()) return rc;
    std::lock_guard<std::mutex> lock(g_ws.mu);
    const size_t chunk = n < (size_t(1) << 24) ? n : (size_t(1) << 24);
    if (int rc = g_ws.reserve(7 * padded(chunk * 8))) return rc;
    unsigned char* const base = host_call_base(7 * padded(chunk * 8));
    const double* in[6] = {P_b, V_a, T_e, V_vac, Pstar, P_T};
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = (n - off < chunk) ? n - off : chunk;
        Carver cv(base);
        double* d[7];
        for (int i = 0; i < 6; ++i) d[i] = cv.take<double>(chunk);
        const size_t in_end = cv.off;
        d[6] = cv.take<double>(chunk);
        Mover mv(base, in_end, cv.off);
        for (int i = 0; i < 6; ++i) PEM_TRY(mv.in(in[i] + off, d[i], m * 8));
        PEM_TRY(mv.flush_in());
        if (int rc = pem_cathode_f64_dev(m, d[0], d[1], d[2], d[3], d[4], d[5], torr2pa, d[6], nullptr)) return rc;
        PEM_TRY(mv.out(V_cc + off, d[6], m * 8));
        PEM_TRY(mv.finish());
    }
    return PEM_OK;
}

int pem_thruster_f64(size_t n, const double* V_a, const double* V_cc, const double* mdot_a, const double* a_1,
                     double* I_B0, double* I_d, double* T, double* eta_c, double* eta_m, double* eta_v, double* eta_a,
                     double* v_exh) {
    if (n == 0) return PEM_OK;
    if (!V_a || !V_cc || !mdot_a || !a_1) return fail(PEM_ERR_INVALID_ARG, "pem_thruster: NULL input array");
    if (int rc = check_device()) return rc;
    if (int rc = use_default_device()) return rc;
    std::lock_guard<std::mutex> lock(g_ws.mu);
    const size_t chunk = n < (size_t(1) << 24) ? n : (size_t(1) << 24);
    if (int rc = g_ws.reserve(12 * padded(chunk * 8))) return rc;
    unsigned char* const base = host_call_base(12 * padded(chunk * 8));
    const double* in[4] = {V_a, V_cc, mdot_a, a_1};
    double* out[8] = {I_B0, I_d, T, eta_c, eta_m, eta_v, eta_a, v_exh};
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = (n - off < chunk) ? n - off : chunk;
        Carver cv(base);
        double *di[4], *dout[8];
        for (auto& p : di) p = cv.take<double>(chunk);
        const size_t in_end = cv.off;
        for (int i = 0; i < 8; ++i) dout[i] = out[i] ? cv.take<double>(chunk) : nullptr;
        Mover mv(base, in_end, cv.off);
        for (int i = 0; i < 4; ++i) PEM_TRY(mv.in(in[i] + off, di[i], m * 8));
        PEM_TRY(mv.flush_in());
        if (int rc = pem_thruster_f64_dev(m, di[0], di[1], di[2], di[3], dout[0], dout[1], dout[2], dout[3], dout[4],
                                          dout[5], dout[6], dout[7], nullptr))
            return rc;
        for (int i = 0; i < 8; ++i)
            if (out[i]) PEM_TRY(mv.out(out[i] + off, dout[i], m * 8));
        PEM_TRY(mv.finish());
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
    if (int rc = use_default_device()) return rc;
    std::lock_guard<std::mutex> lock(g_ws.mu);
    const size_t R = (size_t)n_radii;
    // bound the profile chunk to ~256 MiB of device memory
    size_t chunk = (size_t(1) << 28) / (NANG * R * 8);
    if (chunk < 1024) chunk = 1024;
    if (chunk > n) chunk = n;
    chunk = (chunk + 63) & ~size_t(63);
    const size_t need = 10 * padded(chunk * 8) + padded(chunk * NANG * R * 8) + 2 * padded(chunk * R * 8) + padded(chunk);
    if (int rc = g_ws.reserve(need)) return rc;
    unsigned char* const base = host_call_base(need);
    const double* in[10] = {P_b, c0, c1, c2, c3, c4, c5, sigma_cex, I_B0, T};
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = (n - off < chunk) ? n - off : chunk;
        Carver cv(base);
        double* d[10];
        for (auto& p : d) p = cv.take<double>(chunk);
        const size_t in_end = cv.off;
        double* dj = cv.take<double>(chunk * NANG * R);
        double* ddiv = cv.take<double>(chunk * R);
        double* dtc = cv.take<double>(chunk * R);
        uint8_t* dinv = cv.take<uint8_t>(chunk);
        Mover mv(base, in_end, cv.off);
        for (int i = 0; i < 10; ++i)
            if (in[i]) PEM_TRY(mv.in(in[i] + off, d[i], m * 8));
        PEM_TRY(mv.flush_in());
        if (int rc = pem_plume_f64_dev(m, n_radii, radii, torr2pa, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8],
                                       T ? d[9] : nullptr, dj, ddiv, T ? dtc : nullptr, invalid ? dinv : nullptr, nullptr))
            return rc;
        PEM_TRY(mv.out(j_ion + off * NANG * R, dj, m * NANG * R * 8));
        PEM_TRY(mv.out(div_angle + off * R, ddiv, m * R * 8));
        if (T) PEM_TRY(mv.out(T_c + off * R, dtc, m * R * 8));
        if (invalid) PEM_TRY(mv.out(invalid + off, dinv, m));
        PEM_TRY(mv.finish());
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
    if (int rc = use_default_device()) return rc;
    std::lock_guard<std::mutex> lock(g_ws.mu);
    size_t chunk = (size_t(1) << 28) / (NANG * 8);
    if (chunk > n) chunk = n;
    chunk = (chunk + 63) & ~size_t(63);
    const size_t need = 20 * padded(chunk * 8) + padded(chunk * NANG * 8) + padded(chunk);
    if (int rc = g_ws.reserve(need)) return rc;
    unsigned char* const base = host_call_base(need);
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = (n - off < chunk) ? n - off : chunk;
        Carver cv(base);
        double* d[15];
        for (auto& p : d) p = cv.take<double>(chunk);
        const size_t in_end = cv.off;
        double* dvcc = cv.take<double>(chunk);
        double* dib0 = cv.take<double>(chunk);
        double* dT = cv.take<double>(chunk);
        double* ddiv = cv.take<double>(chunk);
        double* dtc = cv.take<double>(chunk);
        uint8_t* dinv = cv.take<uint8_t>(chunk);
        double* dj = cv.take<double>(chunk * NANG);   // last: without a profile the staged copy-back stops before it
        Mover mv(base, in_end, cv.off);
        for (int i = 0; i < 15; ++i) PEM_TRY(mv.in(in[i] + off, d[i], m * 8));
        PEM_TRY(mv.flush_in());
        if (int rc = pem_coupled_f64_dev(m, torr2pa, radius, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9],
                                         d[10], d[11], d[12], d[13], d[14], dvcc, I_B0 ? dib0 : nullptr, T ? dT : nullptr,
                                         j_ion ? dj : nullptr, ddiv, dtc, invalid ? dinv : nullptr, nullptr))
            return rc;
        PEM_TRY(mv.out(V_cc + off, dvcc, m * 8));
        if (I_B0) PEM_TRY(mv.out(I_B0 + off, dib0, m * 8));
        if (T) PEM_TRY(mv.out(T + off, dT, m * 8));
        if (j_ion) PEM_TRY(mv.out(j_ion + off * NANG, dj, m * NANG * 8));
        PEM_TRY(mv.out(div_angle + off, ddiv, m * 8));
        PEM_TRY(mv.out(T_c + off, dtc, m * 8));
        if (invalid) PEM_TRY(mv.out(invalid + off, dinv, m));
        PEM_TRY(mv.finish());
    }
    return PEM_OK;
}

}  // extern "C"
