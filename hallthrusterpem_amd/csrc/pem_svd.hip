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
// the other dimension r <= 16, so they stream `field` once and are HBM-bound; the products run on
// v_mfma_f64_16x16x4_f64 (D[16x16] += A[16x4] B[4x16]; lane l holds A[l&15][l>>4], B[l>>4][l&15] and
// D[(l>>4) + 4 i][l&15], i = 0..3 -- MI355X guide section 3, the f64 map differs from the f32/bf16 one).
// A wave owns 16 consecutive samples per tile; their dof values are one contiguous block of `field`, moved
// between HBM and LDS with 16-byte-per-lane accesses, exactly like the profile stores of pem_kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"

namespace {

constexpr int WAVES = 4;
constexpr int BLOCK = 64 * WAVES;
constexpr int RMAX = 16;
constexpr int DOF_MAX = PEM_SVD_MAX_DOF;

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double norm_fwd(int mode, double scale, double x) {
    return mode == PEM_NORM_LOG10 ? log10(x) : (mode == PEM_NORM_LINEAR ? x * scale : x);
}
__device__ __forceinline__ double norm_inv(int mode, double scale, double y) {
    return mode == PEM_NORM_LOG10 ? exp10(y) : (mode == PEM_NORM_LINEAR ? y / scale : y);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS: basis_p[ksteps*4][16] (zero padded) | per wave: tile[16*dof] (+2 slack)
__global__ __launch_bounds__(BLOCK) void svd_compress_kernel(long long n, int dof, int r, int mode, double scale,
                                                             const double* __restrict__ field,
                                                             const double* __restrict__ basis,
                                                             double* __restrict__ latent) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int ksteps = (dof + 3) / 4;
    double* basis_p = lds;                                   // [ksteps*4][16]
    const int tile_doubles = (16 * dof + 3) & ~1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* tile = lds + ksteps * 64 + wave * tile_doubles;  // [16][dof]
    for (int i = tid; i < ksteps * 64; i += BLOCK) {
        const int k = i >> 4, c = i & 15;
        basis_p[i] = (k < dof && c < r) ? basis[(size_t)k * r + c] : 0.0;
    }
    __syncthreads();

    const int row = lane & 15, quad = lane >> 4;
    const long long ntiles = (n + 15) / 16;
    const long long stride = (long long)gridDim.x * WAVES;
    for (long long t = (long long)blockIdx.x * WAVES + wave; t < ntiles; t += stride) {
        const long long s0 = t * 16;
        long long valid = (n - s0) * dof;                     // values of this tile that exist
        if (valid > 16LL * dof) valid = 16LL * dof;
        const double* src = field + s0 * dof;
        // HBM -> LDS, normalising on the way; 16-byte pieces when the tile base is 16-byte aligned
        if ((((uintptr_t)src) & 15) == 0) {
            const int pieces = (int)(valid >> 1);
            const f64x2* src2 = reinterpret_cast<const f64x2*>(src);
            for (int i = lane; i < pieces; i += 64) {
                const f64x2 v = __builtin_nontemporal_load(src2 + i);
                tile[2 * i] = norm_fwd(mode, scale, v.x);
                tile[2 * i + 1] = norm_fwd(mode, scale, v.y);
            }
            if ((valid & 1) && lane == 0) tile[valid - 1] = norm_fwd(mode, scale, src[valid - 1]);
        } else {
            for (int i = lane; i < (int)valid; i += 64) tile[i] = norm_fwd(mode, scale, src[i]);
        }
        for (int i = (int)valid + lane; i < 16 * dof; i += 64) tile[i] = 0.0;   // rows past the end of the batch
        wave_lds_sync();

        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int step = 0; step < ksteps; ++step) {
            const int k = 4 * step + quad;
            const double a = k < dof ? tile[row * dof + k] : 0.0;       // A[sample row][k]
            const double b = basis_p[k * 16 + row];                      // B[k][column = lane & 15]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        // D[(quad + 4 i)][col = lane & 15]
        const int col = row;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long smp = s0 + quad + 4 * i;
            if (col < r && smp < n) latent[smp * r + col] = acc[i];
        }
        wave_lds_sync();
    }
}

// LDS: basis_t[16][dof_p] with dof_p = 16*ceil(dof/16) (zero padded) | per wave: tile[16*dof] (+2)
__global__ __launch_bounds__(BLOCK) void svd_reconstruct_kernel(long long n, int dof, int r, int mode, double scale,
                                                                const double* __restrict__ latent,
                                                                const double* __restrict__ basis,
                                                                double* __restrict__ field) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const int groups = (dof + 15) / 16, dof_p = groups * 16;
    double* basis_t = lds;                                   // [16][dof_p]: basis_t[j][k] = basis[k][j]
    const int tile_doubles = (16 * dof + 3) & ~1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* tile = lds + 16 * dof_p + wave * tile_doubles;
    for (int i = tid; i < 16 * dof_p; i += BLOCK) {
        const int j = i / dof_p, k = i - j * dof_p;
        basis_t[i] = (k < dof && j < r) ? basis[(size_t)k * r + j] : 0.0;
    }
    __syncthreads();

    const int row = lane & 15, quad = lane >> 4;
    const long long ntiles = (n + 15) / 16;
    const long long stride = (long long)gridDim.x * WAVES;
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
        for (int g = 0; g < groups; ++g) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                const double b = basis_t[(4 * step + quad) * dof_p + 16 * g + row];   // B[j][angle 16 g + (lane & 15)]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[step], b, acc, 0, 0, 0);
            }
            const int k = 16 * g + row;
            if (k < dof) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[(quad + 4 * i) * dof + k] = acc[i];
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
                v.x = norm_inv(mode, scale, tile[2 * i]);
                v.y = norm_inv(mode, scale, tile[2 * i + 1]);
                __builtin_nontemporal_store(v, dst2 + i);
            }
            if ((valid & 1) && lane == 0) dst[valid - 1] = norm_inv(mode, scale, tile[valid - 1]);
        } else {
            for (int i = lane; i < (int)valid; i += 64) dst[i] = norm_inv(mode, scale, tile[i]);
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

unsigned grid_for(size_t n) {
    size_t tiles = (n + 15) / 16, blocks = (tiles + WAVES - 1) / WAVES;
    if (blocks > 256 * 2) blocks = 256 * 2;     // persistent: LDS admits 1-2 workgroups per CU
    return (unsigned)(blocks ? blocks : 1);
}

}  // namespace

extern "C" {

int pem_svd_compress_f64_dev(size_t n, int dof, int rank, int norm, double norm_scale, const double* field,
                             const double* basis, double* latent, pem_stream_t stream) {
    if (int rc = check_args("pem_svd_compress", n, dof, rank, norm, field, basis, latent)) return rc;
    if (n == 0) return PEM_OK;
    if (int rc = pem::check_device()) return rc;
    const size_t lds = ((size_t)((dof + 3) / 4) * 64 + (size_t)WAVES * ((16 * dof + 3) & ~1)) * 8;
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(svd_compress_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    HIP_TRY(attr);
    hipLaunchKernelGGL(svd_compress_kernel, dim3(grid_for(n)), dim3(BLOCK), lds, static_cast<hipStream_t>(stream),
                       (long long)n, dof, rank, norm, norm_scale, field, basis, latent);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

int pem_svd_reconstruct_f64_dev(size_t n, int dof, int rank, int norm, double norm_scale, const double* latent,
                                const double* basis, double* field, pem_stream_t stream) {
    if (int rc = check_args("pem_svd_reconstruct", n, dof, rank, norm, latent, basis, field)) return rc;
    if (n == 0) return PEM_OK;
    if (int rc = pem::check_device()) return rc;
    const size_t lds = ((size_t)16 * (((dof + 15) / 16) * 16) + (size_t)WAVES * ((16 * dof + 3) & ~1)) * 8;
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(svd_reconstruct_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    HIP_TRY(attr);
    hipLaunchKernelGGL(svd_reconstruct_kernel, dim3(grid_for(n)), dim3(BLOCK), lds, static_cast<hipStream_t>(stream),
                       (long long)n, dof, rank, norm, norm_scale, latent, basis, field);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // extern "C"
