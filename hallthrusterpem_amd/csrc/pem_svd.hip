// pem_svd.hip -- SVD compression / reconstruction of field QoIs with fp64 MFMA (gfx950).
//
// What it stands in for: amisc's `Compression(method='svd')` applied to the field outputs of PEM v0
// (`j_ion`, norm log10, and `u_ion`, norm linear(1e-3): scripts/pem_v0/pem_v0_SPT-100.yml:207-214,273-280;
// call sites scripts/gen_data.py:261-294).  amisc is third-party and absent from the reference tree, so parity
// is UNPINNED: the formulas are this library's own --
//     compress:     latent[n][r] = norm(field[n][dof]) @ basis[dof][r]
//     reconstruct:  field[n][dof] = denorm(latent[n][r] @ basis[dof][r]^T)
// with norm/denorm = identity, log10 / 10^x, or scale s / 1/s -- and are checked against numpy matmul in fp64.
//
// These are the only GEMM-shaped pieces of the path: tall-skinny, K or N = dof <= 208,
// the other dimension r <= 16, so they stream `field` once and are HBM-bound.
//   reconstruct: v_mfma_f64_16x16x4_f64 (D[16x16] += A[16x4] B[4x16]; lane l holds A[l&15][l>>4],
//                B[l>>4][l&15] and D[(l>>4) + 4 i][l&15], i = 0..3 -- the f64 map differs from the f32/bf16 one).
//   compress:    fp64 VALU.  Measured on MI355X (tools/microbench/mfma_f64_rate.hip): v_mfma_f64_16x16x4_f64 issues
//                every 147 cycles per SIMD = 14 flop/clk/SIMD (~33 TFLOP/s chip), v_fma_f64 every 7.5 cycles
//                with one wave per SIMD = 17 flop/clk/SIMD and twice that with two.  The MFMA form also pads r to
//                16 columns; with r = 6 it needs 2.7x the flops at a lower rate and ran at 2.3 TB/s, so the
//                compress product uses FMAs (lanes = 16 samples x 4 interleaved k-slices, shuffle-reduced).
// A wave owns 16 consecutive samples per tile; their dof values are one contiguous block of `field`, moved
// between HBM and LDS with 16-byte-per-lane accesses, exactly like the profile stores of pem_kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "pem_common.h"
#include "pem_math.h"
#include "pem_hip.h"

namespace {

constexpr int WAVES = 4;
constexpr int BLOCK = 64 * WAVES;
constexpr int RMAX = 16;
constexpr int DOF_MAX = PEM_SVD_MAX_DOF;

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// MODE is a template parameter: as a run-time value hipcc evaluates log10()/exp10() for every element and selects
// afterwards (the "none" mode then ran no faster than it would with the transcendental in it).
// `logtab`: the LDS copy of the log10 table (pem_math.h), only read in the log10 mode
template <int MODE>
__device__ __forceinline__ double norm_fwd(double scale, double x, const double* logtab) {
    if constexpr (MODE == PEM_NORM_LOG10) return pem::pem_log10_tab(x, logtab);
    else if constexpr (MODE == PEM_NORM_LINEAR) return x * scale;
    else return x;
}
template <int MODE>
__device__ __forceinline__ double norm_inv(double scale, double y) {
    if constexpr (MODE == PEM_NORM_LOG10) return pem::pem_exp10(y);
    else if constexpr (MODE == PEM_NORM_LINEAR) return y / scale;
    else return y;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A wave owns ROWS consecutive samples per tile (16 for dof <= 96; 8 above, so that the tile, the two prefetched tiles in
// registers and the occupancy stay what they are for the 91-point profile: with 16 x 202 values the kernel needed ~400
// registers and 26 KB of LDS per wave, one wave per SIMD, and ran at 1.8 TB/s); lane = (row, k-group), KG = 64 / ROWS
// interleaved k-slices.
// LDS: basis_p[ksteps*KG][16] (zero padded) | per wave: tile[ROWS*dof] + KG zeroed slack doubles | log10 mode: table[2048]
// UN = 16-byte pieces per lane of one tile (ceil(ROWS*dof/2/64)); the pieces of the NEXT tile are loaded into
// registers while the current tile is multiplied.  RT = latent columns computed (>= rank, zero-padded basis).
// Index arithmetic is kept 32-bit and out of the inner loops: the first version spent 1740 VALU instructions per
// 16-sample tile, most of them 64-bit address math (rocprofv3 SQ_INSTS_VALU), and ran at 2.3 TB/s.
// MFMA = true (ROWS = 16 only): the tile's 16 x dof by dof x 16 product on v_mfma_f64_16x16x4_f64 instead of VALU FMAs.
// Measured rates (profiles/microbench_rates_r02c.txt): the fp64 matrix pipe sustains 78 TFLOP/s from one wave per SIMD
// and runs beside the VALU; v_fma_f64 reaches 39-60.  Without a norm the kernel is HBM-bound and the VALU form (no padding
// of the rank to 16 columns, no dependent MFMA chain) is as fast; with the log10 norm the VALU is what the kernel waits
// for (profiles/compress_pmc_r02d.txt: 55 VALU instructions per value, VALU active 2/3 of the time at two waves per
// SIMD), and moving the 8 FMAs + 4 basis reads per value to the matrix pipe takes a quarter of them away.
template <int ROWS, int UN, int RT, int MODE, bool MFMA = false>
__global__ __launch_bounds__(BLOCK) void svd_compress_kernel(long long n, int dof, int r, double scale,
                                                             const double* __restrict__ field,
                                                             const double* __restrict__ basis,
                                                             double* __restrict__ latent) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    constexpr int KG = 64 / ROWS;                            // k-groups: lane (row, grp) takes k = grp, grp + KG, ...
    const int ksteps = (dof + KG - 1) / KG;
    double* basis_p = lds;                                   // [ksteps*KG][16]
    const int tile_doubles = (ROWS * dof + KG + 1) & ~1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* tile = lds + ksteps * KG * 16 + wave * tile_doubles;  // [ROWS][dof] + slack
    double* logtab = lds + ksteps * KG * 16 + WAVES * tile_doubles;  // log10 mode: 16 KB table behind the tiles
    if constexpr (MODE == PEM_NORM_LOG10) pem::load_log_table(logtab, tid, BLOCK);
    for (int i = tid; i < ksteps * KG * 16; i += BLOCK) {
        const int k = i >> 4, c = i & 15;
        basis_p[i] = (k < dof && c < r) ? basis[(size_t)k * r + c] : 0.0;
    }
    if (lane < KG) tile[ROWS * dof + lane] = 0.0;            // read (and masked) by the last k-step
    __syncthreads();

    const int row = lane & (ROWS - 1), quad = lane / ROWS;    // `quad` = k-group index (0..KG-1)
    const int tile_len = ROWS * dof;                          // doubles per full tile
    const long long ntiles = (n + ROWS - 1) / ROWS;
    const long long stride = (long long)gridDim.x * WAVES;
    const bool vec_ok = (((uintptr_t)field) & 15) == 0;       // tiles start at multiples of ROWS*dof*8 bytes

    // `len` = doubles of the tile that exist (tile_len except for the last tile of the batch)
    auto fetch = [&](const double* tp, int len, f64x2 (&v)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = 2 * (u * 64 + lane);
            if (vec_ok && i + 1 < len) {
                v[u] = *reinterpret_cast<const f64x2*>(tp + i);
            } else {
                v[u].x = i < len ? tp[i] : 1.0;
                v[u].y = i + 1 < len ? tp[i + 1] : 1.0;
            }
        }
    };
    auto tile_length = [&](long long t) {
        const long long rest = (n - t * ROWS) * dof;
        return rest < tile_len ? (int)rest : tile_len;
    };

    // Two tiles are kept in flight in registers beyond the one in LDS: with one, the bytes in flight per CU are
    // bounded by the LDS tiles (8 waves x 12 KB) and the read stream ran at 2.6 TB/s; with two 4.2 TB/s (232 us per
    // 1.25e6 x 91 batch); with three the 232 VGPRs cost more than the extra bytes in flight bring (265 us).
    long long t = (long long)blockIdx.x * WAVES + wave;
    f64x2 nxt[UN], nxt2[UN];
    if (t < ntiles) fetch(field + t * tile_len, tile_length(t), nxt);
    if (t + stride < ntiles) fetch(field + (t + stride) * tile_len, tile_length(t + stride), nxt2);
    const double* a_base = tile + row * dof + quad;
    const f64x2* b_base = reinterpret_cast<const f64x2*>(basis_p + quad * 16);
    for (; t < ntiles; t += stride) {
        const int len = tile_length(t);
        // registers -> LDS with the variable's norm applied; entries past the end of the batch become 0
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = 2 * (u * 64 + lane);
            if (i < tile_len) {
                f64x2 w;
                w.x = i < len ? norm_fwd<MODE>(scale, nxt[u].x, logtab) : 0.0;
                w.y = i + 1 < len ? norm_fwd<MODE>(scale, nxt[u].y, logtab) : 0.0;
                *reinterpret_cast<f64x2*>(tile + i) = w;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) nxt[u] = nxt2[u];
        if (t + 2 * stride < ntiles) fetch(field + (t + 2 * stride) * tile_len, tile_length(t + 2 * stride), nxt2);
        wave_lds_sync();

        if constexpr (MFMA) {
            static_assert(!MFMA || ROWS == 16, "the 16x16x4 tile is 16 samples by 16 latent columns");
            // D[16 samples][16 columns] += A[16][4] B[4][16] per k-step: lane (row, quad) feeds A[row][4 step + quad] =
            // tile[row][k] and B[4 step + quad][row] = basis_p[k][column = row]; two accumulators halve the dependent chain
            f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            const double* ap = a_base;
            const double* bp = basis_p + quad * 16 + row;
            int step = 0;
            for (; step + 2 < ksteps; step += 2) {
                const double a0 = ap[0], a1 = ap[KG], b0 = bp[0], b1 = bp[KG * 16];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
                ap += 2 * KG;
                bp += 2 * KG * 16;
            }
            for (; step < ksteps; ++step) {   // the last one or two steps: k >= dof is masked (0 * inf = NaN otherwise)
                const double a = quad + KG * step < dof ? *ap : 0.0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, *bp, acc0, 0, 0, 0);
                ap += KG;
                bp += KG * 16;
            }
            // lane (row, quad) holds D[quad + 4 i][column = row], i = 0..3
            wave_lds_sync();
            if (row < r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[(quad + 4 * i) * r + row] = acc0[i] + acc1[i];
            }
        } else {
            // lane (row = sample, quad): partial dot products over k = quad, quad + 4, ... for RT latent columns
            double acc[RT];
#pragma unroll
            for (int j = 0; j < RT; ++j) acc[j] = 0.0;
            const double* ap = a_base;
            const f64x2* bp = b_base;
#pragma unroll 4
            for (int step = 0; step < ksteps - 1; ++step) {
                const double a = *ap;
                ap += KG;
#pragma unroll
                for (int j = 0; j < RT; j += 2) {
                    const f64x2 b = bp[j >> 1];
                    acc[j] = fma(a, b.x, acc[j]);
                    if (j + 1 < RT) acc[j + 1] = fma(a, b.y, acc[j + 1]);
                }
                bp += KG * 8;                     // KG basis rows of 16 doubles
            }
            {   // last step: k >= dof would read the next sample's first values (or slack) against a zero basis row --
                // masked, because 0 * inf = NaN would let one sample's non-finite value poison its neighbour's latents
                const double a = quad + KG * (ksteps - 1) < dof ? *ap : 0.0;
#pragma unroll
                for (int j = 0; j < RT; j += 2) {
                    const f64x2 b = bp[j >> 1];
                    acc[j] = fma(a, b.x, acc[j]);
                    if (j + 1 < RT) acc[j + 1] = fma(a, b.y, acc[j + 1]);
                }
            }
#pragma unroll
            for (int j = 0; j < RT; ++j) {
#pragma unroll
                for (int sh = ROWS; sh < 64; sh <<= 1) acc[j] += __shfl_xor(acc[j], sh);
            }
            // every k-group now holds the full sums.  The tile's ROWS x r latents are one contiguous block of `latent`:
            // gather them in LDS (the tile is consumed) and store them with consecutive lanes.
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < RT; ++j)
                if ((j & (KG - 1)) == quad && j < r) tile[row * r + j] = acc[j];
        }
        wave_lds_sync();
        {
            double* out = latent + t * ROWS * r;
            const int count = (len / dof) * r;
            for (int e = lane; e < count; e += 64) out[e] = tile[e];
        }
        wave_lds_sync();
    }
}

// compress for dof <= 96 (the 91-point profile): no LDS tile at all.  A lane loads the values it will feed to the matrix pipe
// straight from `field` in the operand layout of v_mfma_f64_16x16x4_f64 -- lane (row, quad) takes field[sample row][4 step
// + quad], steps 0..23: per instruction 16 rows x 32 contiguous bytes, every byte of the tile loaded exactly once -- takes
// their log10 in registers (table version, the only LDS traffic left) and issues the 16 x 4 by 4 x 16 products against
// basis rows it keeps in registers for the whole launch.  Against the tiled kernel above that removes the transposition
// through LDS (12 16-byte writes and 23 reads per lane and tile) and with it the 47 KB of tiles per workgroup that held the
// kernel at two waves per SIMD: this one is limited by its ~140 registers (three waves per SIMD), so one wave's loads are
// in flight while the others take logs.
constexpr int DIRECT_STEPS = 24;     // dof <= 96
// Which k the matrix pipe's slot (step, quad) carries.  Any one-to-one map will do as long as A and B agree; this one gives a
// lane two CONSECUTIVE k per pair of steps -- k = 8 (step / 2) + 2 quad + (step & 1) -- so that it loads them with one 16-byte
// instruction: 12 loads of 16 rows x 64 contiguous bytes per tile instead of 24 of 16 x 32.
__device__ __forceinline__ constexpr int direct_k(int step, int quad) { return 8 * (step >> 1) + 2 * quad + (step & 1); }
// BREG: the lane's 24 basis values live in registers for the whole launch (160 VGPRs, three waves per SIMD); otherwise they
// are re-read from LDS per tile (conflict-free 8-byte reads; PEM_SVD_BREG=0, an A/B switch).  Both are compiled for three waves
// per SIMD: at four the LDS form spilled 2-12 registers to scratch (round 4: no kernel of the library uses scratch).
template <int MODE, bool BREG>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(3)))
void svd_compress_direct_kernel(long long n, int dof, int r, double scale, const double* __restrict__ field,
                                const double* __restrict__ basis, double* __restrict__ latent) {
    __shared__ __attribute__((aligned(16))) double logtab[MODE == PEM_NORM_LOG10 ? pem::LOG_TABLE_DOUBLES : 2];
    __shared__ double bas[BREG ? 1 : 4 * DIRECT_STEPS * 16];        // [k][column], zero padded
    if constexpr (MODE == PEM_NORM_LOG10) pem::load_log_table(logtab, threadIdx.x, BLOCK);
    if constexpr (!BREG) {
        for (int i = threadIdx.x; i < 4 * DIRECT_STEPS * 16; i += BLOCK) {
            const int k = i >> 4, c = i & 15;
            bas[i] = (k < dof && c < r) ? basis[(size_t)k * r + c] : 0.0;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane & 15, quad = lane >> 4;
    // B[4 step + quad][column = row] = basis[k][row], zero past the rank and past dof
    double b[BREG ? DIRECT_STEPS : 1];
    if constexpr (BREG) {
#pragma unroll
        for (int step = 0; step < DIRECT_STEPS; ++step) {
            const int k = direct_k(step, quad);
            b[step] = (k < dof && row < r) ? basis[(size_t)k * r + row] : 0.0;
        }
    }
    const double* bl = bas + row;
    const long long ntiles = (n + 15) / 16;
    const long long stride = (long long)gridDim.x * WAVES;
    for (long long t = (long long)blockIdx.x * WAVES + wave; t < ntiles; t += stride) {
        const long long s0 = t * 16;
        const long long smp = s0 + row < n ? s0 + row : n - 1;         // rows past the batch repeat its last sample
        const double* src = field + smp * dof;
        double a[DIRECT_STEPS];
        constexpr double PAD = MODE == PEM_NORM_LOG10 ? 1.0 : 0.0;
#pragma unroll
        for (int step = 0; step < DIRECT_STEPS; step += 2) {       // two consecutive k per lane: one 16-byte load
            const int k = direct_k(step, quad);
            if (k + 1 < dof) {
                typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));   // rows of an odd dof start 8 bytes off
                const f64x2_a8 v = *reinterpret_cast<const f64x2_a8*>(src + k);
                a[step] = v.x;
                a[step + 1] = v.y;
            } else {
                a[step] = k < dof ? src[k] : PAD;
                a[step + 1] = PAD;
            }
        }
        f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int step = 0; step < DIRECT_STEPS; step += 2) {
            // k >= dof: the value is 0 AFTER the norm (log10(1) = 0 exactly), and it meets a zero basis row
            const double l0 = norm_fwd<MODE>(scale, a[step], logtab), l1 = norm_fwd<MODE>(scale, a[step + 1], logtab);
            const double b0 = BREG ? b[BREG ? step : 0] : bl[16 * direct_k(step, quad)], b1 = BREG ? b[BREG ? step + 1 : 0] : bl[16 * direct_k(step + 1, quad)];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(l0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(l1, b1, acc1, 0, 0, 0);
        }
        // lane (row, quad) holds D[sample quad + 4 i][column = row]
        if (row < r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long sg = s0 + quad + 4 * i;
                if (sg < n) latent[sg * r + row] = acc0[i] + acc1[i];
            }
        }
    }
}

// LDS: basis_t[16][dof_p] with dof_p = 16*ceil(dof/16) (zero padded) | per wave: tile[16*dof] (+2)
// BREG (dof <= 96): the lane's B operands -- basis[angle 16 g + row][4 step + quad], 6 groups x 4 steps -- stay in registers for
// the whole launch instead of being re-read from LDS per tile: the workgroup's LDS is then the four tiles only (46.6 instead
// of 58.7 KB) and three workgroups fit a CU instead of two.
template <int MODE, bool BREG>
__global__ __launch_bounds__(BLOCK) void svd_reconstruct_kernel(long long n, int dof, int r, double scale,
                                                                const double* __restrict__ latent,
                                                                const double* __restrict__ basis,
                                                                double* __restrict__ field) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int groups = (dof + 15) / 16, dof_p = groups * 16;
    double* basis_t = lds;                                   // [16][dof_p]: basis_t[j][k] = basis[k][j] (not with BREG)
    const int tile_doubles = (16 * dof + 3) & ~1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* tile = lds + (BREG ? 0 : 16 * dof_p) + wave * tile_doubles;
    const int row = lane & 15, quad = lane >> 4;
    constexpr int RGROUPS = 6;                               // dof <= 96
    double breg[BREG ? RGROUPS * 4 : 1];
    if constexpr (BREG) {
#pragma unroll
        for (int g = 0; g < RGROUPS; ++g) {
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const int k = 16 * g + row, j = 4 * step + quad;
                breg[g * 4 + step] = (k < dof && j < r) ? basis[(size_t)k * r + j] : 0.0;
            }
        }
    } else {
        for (int i = tid; i < 16 * dof_p; i += BLOCK) {
            const int j = i / dof_p, k = i - j * dof_p;
            basis_t[i] = (k < dof && j < r) ? basis[(size_t)k * r + j] : 0.0;
        }
        __syncthreads();
    }

    const long long ntiles = (n + 15) / 16;
    const long long stride = (long long)gridDim.x * WAVES;
    const int ksteps = (r + 3) >> 2;      // k-steps of four latent columns that hold any: 2 of 4 for rank 5..8 (the rest multiplies zeros)
    for (long long t = (long long)blockIdx.x * WAVES + wave; t < ntiles; t += stride) {
        const long long s0 = t * 16;
        // A fragments: latent[s0 + row][4 step + quad], 4 k-steps cover the 16 padded latent columns
        double a[4];
        const long long smp_a = s0 + row;
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int j = 4 * step + quad;
            a[step] = (j < r && smp_a < n) ? latent[smp_a * r + j] : 0.0;
        }
        auto one_group = [&](int g, const double (&b)[4]) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int step = 0; step < 4; ++step)
                if (step < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[step], b[step], acc, 0, 0, 0);
            const int k = 16 * g + row;
            if (k < dof) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[(quad + 4 * i) * dof + k] = acc[i];
            }
        };
        if constexpr (BREG) {
#pragma unroll
            for (int g = 0; g < RGROUPS; ++g) {
                if (g < groups) {
                    const double b[4] = {breg[BREG ? g * 4 : 0], breg[BREG ? g * 4 + 1 : 0], breg[BREG ? g * 4 + 2 : 0], breg[BREG ? g * 4 + 3 : 0]};
                    one_group(g, b);
                }
            }
        } else {
            for (int g = 0; g < groups; ++g) {
                double b[4];
#pragma unroll
                for (int step = 0; step < 4; ++step) b[step] = basis_t[(4 * step + quad) * dof_p + 16 * g + row];   // B[j][angle 16 g + (lane & 15)]
                one_group(g, b);
            }
        }
        wave_lds_sync();
        long long valid = (n - s0) * dof;
        if (valid > 16LL * dof) valid = 16LL * dof;
        double* dst = field + s0 * dof;
        if ((((uintptr_t)dst) & 15) == 0) {
            const int pieces = (int)(valid >> 1);
            f64x2* dst2 = reinterpret_cast<f64x2*>(dst);
            for (int i = lane; i < pieces; i += 64) {
                f64x2 v;
                v.x = norm_inv<MODE>(scale, tile[2 * i]);
                v.y = norm_inv<MODE>(scale, tile[2 * i + 1]);
                __builtin_nontemporal_store(v, dst2 + i);
            }
            if ((valid & 1) && lane == 0) dst[valid - 1] = norm_inv<MODE>(scale, tile[valid - 1]);
        } else {
            for (int i = lane; i < (int)valid; i += 64) dst[i] = norm_inv<MODE>(scale, tile[i]);
        }
        wave_lds_sync();
    }
}

int check_args(const char* who, size_t n, int dof, int r, int mode, const void* a, const void* b, const void* c) {
    if (dof < 1 || dof > DOF_MAX) return pem::fail(PEM_ERR_INVALID_ARG, "%s: dof must be in [1, %d]", who, DOF_MAX);
    if (r < 1 || r > RMAX) return pem::fail(PEM_ERR_INVALID_ARG, "%s: rank must be in [1, %d]", who, RMAX);
    if (mode < PEM_NORM_NONE || mode > PEM_NORM_LINEAR) return pem::fail(PEM_ERR_INVALID_ARG, "%s: unknown norm %d", who, mode);
    if (n && (!a || !b || !c)) return pem::fail(PEM_ERR_INVALID_ARG, "%s: NULL array", who);
    return PEM_OK;
}

// The smallest persistent grid that needs as many rounds as `cap` workgroups would: every round full (the last round of a
// grid of `cap` costs a full round however few of its waves have a tile; pem_kernels.hip fast_grid)
// -- where that keeps >= 90 % of the slots in use (with two or three rounds the missing parallelism costs more).
size_t balanced_blocks(size_t need, size_t cap) {
    if (need <= cap) return need;
    if (const char* e = getenv("PEM_SVD_GRID_MULT")) {     // experiments: m x the resident slots (0: one tile per wave)
        const long long m = atoll(e);
        return (m <= 0 || (size_t)m * cap > need) ? need : (size_t)m * cap;
    }
    // more than three rounds of work: four times the resident workgroups, dealt out by the dispatcher as slots free up -- the
    // static walk ends with the slowest wave (pem_kernels.hip persistent_grid); one tile per wave would pay the tables and the
    // register-resident basis once per tile.  1.25e6 profiles, streaming: compress with log10 268 -> 255 us, reconstruct
    // without a norm 183 -> 174 us, the other two unchanged (profiles/svd_grid_r03.txt)
    if (need > 3 * cap) return need < 4 * cap ? need : 4 * cap;
    const size_t rounds = (need + cap - 1) / cap;
    const size_t g = (need + rounds - 1) / rounds;
    return 10 * g >= 9 * cap ? g : cap;
}

unsigned grid_for(size_t n, int per_cu = 2) {
    size_t tiles = (n + 15) / 16, blocks = (tiles + WAVES - 1) / WAVES;
    blocks = balanced_blocks(blocks, 256 * (size_t)per_cu);               // persistent: LDS admits 1-3 workgroups per CU
    return (unsigned)(blocks ? blocks : 1);
}

}  // namespace

extern "C" {

int pem_svd_compress_f64_dev(size_t n, int dof, int rank, int norm, double norm_scale, const double* field,
                             const double* basis, double* latent, pem_stream_t stream) {
    if (int rc = check_args("pem_svd_compress", n, dof, rank, norm, field, basis, latent)) return rc;
    if (n == 0) return PEM_OK;
    if (int rc = pem::check_device()) return rc;
    // ROWS samples per tile: 16 (4 k-groups) for dof <= 96, 8 (8 k-groups) above; UN: 16-byte pieces per lane of a tile
    // (12 covers 16 x 96, 13 covers 8 x 208); RT: latent columns computed (>= rank)
    const int rows = dof <= 96 ? 16 : 8, kg = 64 / rows;
    const size_t lds = ((size_t)((dof + kg - 1) / kg) * kg * 16 + (size_t)WAVES * ((rows * dof + kg + 1) & ~1) +
                        (norm == PEM_NORM_LOG10 ? (size_t)pem::LOG_TABLE_DOUBLES : 0)) * 8;
    const size_t tiles = (n + rows - 1) / rows;
    size_t blocks = balanced_blocks((tiles + WAVES - 1) / WAVES, 256 * 2);
#define PEM_SVD_LAUNCH(ROWS_, UN_, RT_, MODE_)                                                                            \
    do {                                                                                                             \
        static pem::LdsAttrOnce attr;                                                                                \
        HIP_TRY(attr.ensure(reinterpret_cast<const void*>(svd_compress_kernel<ROWS_, UN_, RT_, MODE_>)));             \
        hipLaunchKernelGGL((svd_compress_kernel<ROWS_, UN_, RT_, MODE_>), dim3((unsigned)blocks), dim3(BLOCK), lds, \
                           static_cast<hipStream_t>(stream), (long long)n, dof, rank, norm_scale, field, basis, latent); \
    } while (0)
#define PEM_SVD_LAUNCH_MFMA()                                                                                          \
    do {                                                                                                             \
        static pem::LdsAttrOnce attr;                                                                                \
        HIP_TRY(attr.ensure(reinterpret_cast<const void*>(svd_compress_kernel<16, 12, 16, PEM_NORM_LOG10, true>)));   \
        hipLaunchKernelGGL((svd_compress_kernel<16, 12, 16, PEM_NORM_LOG10, true>), dim3((unsigned)blocks), dim3(BLOCK), lds, \
                           static_cast<hipStream_t>(stream), (long long)n, dof, rank, norm_scale, field, basis, latent); \
    } while (0)
#define PEM_SVD_BY_MODE(ROWS_, UN_, RT_)                                                 \
    do {                                                                            \
        if (norm == PEM_NORM_LOG10 && ROWS_ == 16 && !getenv("PEM_SVD_NO_MFMA")) PEM_SVD_LAUNCH_MFMA();       \
        else if (norm == PEM_NORM_LOG10) PEM_SVD_LAUNCH(ROWS_, UN_, RT_, PEM_NORM_LOG10);       \
        else if (norm == PEM_NORM_LINEAR) PEM_SVD_LAUNCH(ROWS_, UN_, RT_, PEM_NORM_LINEAR); \
        else PEM_SVD_LAUNCH(ROWS_, UN_, RT_, PEM_NORM_NONE);                                \
    } while (0)
    if (dof <= 4 * DIRECT_STEPS && !getenv("PEM_SVD_TILED")) {
        size_t dblocks = ((n + 15) / 16 + WAVES - 1) / WAVES;
        static const bool breg = getenv("PEM_SVD_BREG") ? atoi(getenv("PEM_SVD_BREG")) != 0 : true;
        const size_t per_cu = 3;                      // persistent: workgroups (of four waves) resident per CU, by registers
        dblocks = balanced_blocks(dblocks, 256 * per_cu);
        hipStream_t st = static_cast<hipStream_t>(stream);
#define PEM_SVD_DIRECT(MODE_)                                                                                                   \
    do {                                                                                                                        \
        if (breg) hipLaunchKernelGGL((svd_compress_direct_kernel<MODE_, true>), dim3((unsigned)dblocks), dim3(BLOCK), 0, st,   \
                                     (long long)n, dof, rank, norm_scale, field, basis, latent);                               \
        else hipLaunchKernelGGL((svd_compress_direct_kernel<MODE_, false>), dim3((unsigned)dblocks), dim3(BLOCK), 0, st,       \
                                (long long)n, dof, rank, norm_scale, field, basis, latent);                                    \
    } while (0)
        if (norm == PEM_NORM_LOG10) PEM_SVD_DIRECT(PEM_NORM_LOG10);
        else if (norm == PEM_NORM_LINEAR) PEM_SVD_DIRECT(PEM_NORM_LINEAR);
        else PEM_SVD_DIRECT(PEM_NORM_NONE);
#undef PEM_SVD_DIRECT
        HIP_TRY(hipGetLastError());
        return PEM_OK;
    }
    const int rt = rank <= 4 ? 4 : (rank <= 8 ? 8 : 16);
    if (dof <= 96) {
        if (rt == 4) PEM_SVD_BY_MODE(16, 12, 4); else if (rt == 8) PEM_SVD_BY_MODE(16, 12, 8); else PEM_SVD_BY_MODE(16, 12, 16);
    } else {
        if (rt == 4) PEM_SVD_BY_MODE(8, 13, 4); else if (rt == 8) PEM_SVD_BY_MODE(8, 13, 8); else PEM_SVD_BY_MODE(8, 13, 16);
    }
#undef PEM_SVD_BY_MODE
#undef PEM_SVD_LAUNCH_MFMA
#undef PEM_SVD_LAUNCH
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

int pem_svd_reconstruct_f64_dev(size_t n, int dof, int rank, int norm, double norm_scale, const double* latent,
                                const double* basis, double* field, pem_stream_t stream) {
    if (int rc = check_args("pem_svd_reconstruct", n, dof, rank, norm, latent, basis, field)) return rc;
    if (n == 0) return PEM_OK;
    if (int rc = pem::check_device()) return rc;
    // (A direct form -- denorm and 8-byte stores straight from the MFMA result registers, no LDS tile -- was measured and
    // dropped: 16 lanes write 128 contiguous bytes, but a 728-byte profile row puts those pieces across cache lines, and the
    // kernel ran at 1.4-1.5 TB/s against 3.9 / 5.6 TB/s for the staged form below; profiles/svd_probe_r02k.txt.)
    static const bool rbreg_env = getenv("PEM_SVD_RBREG") ? atoi(getenv("PEM_SVD_RBREG")) != 0 : true;
    const bool rbreg = rbreg_env && dof <= 96;
    const size_t lds = ((rbreg ? 0 : (size_t)16 * (((dof + 15) / 16) * 16)) + (size_t)WAVES * ((16 * dof + 3) & ~1)) * 8;
    const unsigned rgrid = grid_for(n, rbreg ? 3 : 2);       // persistent: workgroups per CU that the LDS admits
#define PEM_SVD_RLAUNCH2(MODE_, BREG_)                                                                               \
    do {                                                                                                             \
        static pem::LdsAttrOnce attr;                                                                                \
        HIP_TRY(attr.ensure(reinterpret_cast<const void*>(svd_reconstruct_kernel<MODE_, BREG_>)));                    \
        hipLaunchKernelGGL((svd_reconstruct_kernel<MODE_, BREG_>), dim3(rgrid), dim3(BLOCK), lds,                    \
                           static_cast<hipStream_t>(stream), (long long)n, dof, rank, norm_scale, latent, basis, field); \
    } while (0)
#define PEM_SVD_RLAUNCH(MODE_)                            \
    do {                                                  \
        if (rbreg) PEM_SVD_RLAUNCH2(MODE_, true);         \
        else PEM_SVD_RLAUNCH2(MODE_, false);              \
    } while (0)
    if (norm == PEM_NORM_LOG10) PEM_SVD_RLAUNCH(PEM_NORM_LOG10);
    else if (norm == PEM_NORM_LINEAR) PEM_SVD_RLAUNCH(PEM_NORM_LINEAR);
    else PEM_SVD_RLAUNCH(PEM_NORM_NONE);
#undef PEM_SVD_RLAUNCH
#undef PEM_SVD_RLAUNCH2
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // extern "C"
