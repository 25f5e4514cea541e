// pem_masks.hip -- the per-sample NaN / outlier masks of scripts/gen_data.py:150-168 (`_filter_outputs`) in one pass (gfx950).
//
// The reference marks, per output variable of shape (num_samples, ...), the samples that hold a NaN and the samples of which MORE
// than int(0.75 * entries) entries lie outside [p25 - f iqr, p75 + f iqr], the percentiles taken per entry over the samples.  The
// percentiles come from csrc/pem_quantile.hip; what is left is `np.any(np.isnan(arr), axis=rest)` and
// `np.sum((arr < lo) | (arr > hi), axis=rest)` -- in torch three elementwise passes over the array, two boolean arrays of its
// shape and two reductions (1e7 x 91: 17 ms), here one coalesced read (1.3 ms): every wave instruction covers whole rows, the
// comparisons are balloted and the bits of a row counted, so a row's result costs a population count and no exchange.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pem_common.h"
#include "pem_hip.h"

namespace {

typedef unsigned long long u64;
constexpr int MBLOCK = 256;
constexpr int MWAVES = MBLOCK / 64;
constexpr int MROWS = 8;        // row groups a wave requests before it consumes any

// m <= 64: a wave instruction covers rpw = 64 / m whole rows, lane = row-in-group * m + column (the layout of the percentile
// passes).  The lane of a row's first column counts the row's bits of the two ballots and writes its results: one store
// instruction per row group, rpw rows wide.
__global__ __launch_bounds__(MBLOCK) void row_masks_narrow_kernel(long long n, int m, size_t ld, const double* __restrict__ data,
                                                                   const double* __restrict__ lo, const double* __restrict__ hi,
                                                                   uint8_t* __restrict__ nan_out, int32_t* __restrict__ outside_out) {
    const int lane = threadIdx.x & 63, rpw = 64 / m;
    const bool active = lane < rpw * m;
    const int c = lane % m, rsub = lane / m;
    const double l = active ? lo[c] : 0.0, h = active ? hi[c] : 0.0;
    const u64 row_bits = m == 64 ? ~0ull : ((1ull << m) - 1);
    const long long wave = (long long)blockIdx.x * MWAVES + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * MWAVES;
    const long long groups = (n + rpw - 1) / rpw;
    for (long long g0 = wave * MROWS; g0 < groups; g0 += nwaves * MROWS) {
        double x[MROWS];
        bool ok[MROWS];
#pragma unroll
        for (int u = 0; u < MROWS; ++u) {
            const long long row = (g0 + u) * rpw + rsub;
            ok[u] = active && row < n;
            x[u] = ok[u] ? data[(size_t)row * ld + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < MROWS; ++u) {
            const u64 out = __ballot(ok[u] && (x[u] < l || x[u] > h)), nan = __ballot(ok[u] && x[u] != x[u]);
            if (ok[u] && c == 0) {
                const long long row = (g0 + u) * rpw + rsub;
                outside_out[row] = __popcll((out >> lane) & row_bits);
                nan_out[row] = ((nan >> lane) & row_bits) ? 1 : 0;
            }
        }
    }
}

// 64 < m <= 64 NC: one row per wave instruction and chunk, column = lane + 64 chunk, bounds in registers.  The results of a
// group of ROWS rows (ROWS x NC = 16 loads in flight per lane) are collected in its first lanes and written by one instruction.
template <int NC, int ROWS = 16 / NC>
__global__ __launch_bounds__(MBLOCK) void row_masks_wide_kernel(long long n, int m, size_t ld, const double* __restrict__ data,
                                                                 const double* __restrict__ lo, const double* __restrict__ hi,
                                                                 uint8_t* __restrict__ nan_out, int32_t* __restrict__ outside_out) {
    const int lane = threadIdx.x & 63;
    double l[NC], h[NC];
    bool on[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = lane + 64 * j;
        on[j] = c < m;
        l[j] = on[j] ? lo[c] : 0.0;
        h[j] = on[j] ? hi[c] : 0.0;
    }
    const long long wave = (long long)blockIdx.x * MWAVES + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * MWAVES;
    for (long long r0 = wave * ROWS; r0 < n; r0 += nwaves * ROWS) {
        double x[ROWS][NC];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
#pragma unroll
            for (int j = 0; j < NC; ++j) x[u][j] = (on[j] && r0 + u < n) ? data[(size_t)(r0 + u) * ld + lane + 64 * j] : 0.0;
        }
        int my_count = 0, my_nan = 0;
#pragma unroll
        for (int u = 0; u < ROWS; ++u) {
            int count = 0;
            u64 nan = 0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                count += __popcll(__ballot(on[j] && (x[u][j] < l[j] || x[u][j] > h[j])));
                nan |= __ballot(on[j] && x[u][j] != x[u][j]);
            }
            if (lane == u) {
                my_count = count;
                my_nan = nan ? 1 : 0;
            }
        }
        if (lane < ROWS && r0 + lane < n) {
            outside_out[r0 + lane] = my_count;
            nan_out[r0 + lane] = (uint8_t)my_nan;
        }
    }
}

}  // namespace

namespace {

// The masks of a campaign's scalar outputs and the verdict of the profile's premask counts, one thread per sample
// (pem_campaign_masks_f64_dev): what drivers.forward_uq_statistics did in some twenty elementwise torch launches over n.
constexpr int CM_MAX_VARS = 8;
struct CampaignVars {
    const double* v[CM_MAX_VARS];
};
// V = 4: four consecutive samples per thread, masks written as 4-byte words (rows of the mask arrays 4-byte aligned: mask_ld a
// multiple of 4) (1e7 samples, three variables + the premask counts: 58 us)
template <int V>
__global__ __launch_bounds__(MBLOCK) void campaign_masks_kernel(long long n, int nvar, CampaignVars vars, const double* __restrict__ q, int q_ld,
                                                                int row25, int row75, double factor, uint8_t* __restrict__ nan_out,
                                                                uint8_t* __restrict__ outl_out, size_t mask_ld, const uint8_t* __restrict__ certain,
                                                                const uint8_t* __restrict__ uncertain, int thresh,
                                                                long long* __restrict__ open_rows, int* __restrict__ open_count, int cap) {
    // gen_data.py:163-166: iqr = p75 - p25; the bounds p25 - f iqr and p75 + f iqr, every operation rounded on its own as numpy does
    double lo[CM_MAX_VARS], hi[CM_MAX_VARS];
#pragma unroll
    for (int k = 0; k < CM_MAX_VARS; ++k) {
        const double p25 = k < nvar ? q[(size_t)row25 * q_ld + k] : 0.0, p75 = k < nvar ? q[(size_t)row75 * q_ld + k] : 0.0;
        const double iqr = __dsub_rn(p75, p25);
        lo[k] = __dsub_rn(p25, __dmul_rn(factor, iqr));
        hi[k] = __dadd_rn(p75, __dmul_rn(factor, iqr));
    }
    const long long stride = (long long)gridDim.x * MBLOCK * V;
    for (long long i = ((long long)blockIdx.x * MBLOCK + threadIdx.x) * V; i < n; i += stride) {
        const int left = n - i < V ? (int)(n - i) : V;
#pragma unroll
        for (int k = 0; k < CM_MAX_VARS; ++k) {
            if (k < nvar) {
                unsigned nanw = 0, outw = 0;
#pragma unroll
                for (int u = 0; u < V; ++u) {
                    const double x = u < left ? vars.v[k][i + u] : 0.0;       // (plain loads: with the non-temporal hint the pass took 85 us instead of 58)
                    nanw |= (unsigned)(x != x) << (8 * u);                        // np.isnan
                    outw |= (unsigned)((x < lo[k]) | (x > hi[k])) << (8 * u);     // one entry per sample: "any entry outside" (a NaN compares false)
                }
                uint8_t* np_ = nan_out + (size_t)k * mask_ld + i;
                uint8_t* op_ = outl_out + (size_t)k * mask_ld + i;
                if (V == 4 && left == V) {
                    *reinterpret_cast<unsigned*>(np_) = nanw;
                    *reinterpret_cast<unsigned*>(op_) = outw;
                } else {
                    for (int u = 0; u < left; ++u) {
                        np_[u] = (uint8_t)(nanw >> (8 * u));
                        op_[u] = (uint8_t)(outw >> (8 * u));
                    }
                }
            }
        }
        if (certain) {
            unsigned outw = 0, cw = 0, uw = 0;
            if (V == 4 && left == V) {                                            // (the counts of four samples as one word each)
                cw = *reinterpret_cast<const unsigned*>(certain + i);
                uw = *reinterpret_cast<const unsigned*>(uncertain + i);
            } else {
                for (int u = 0; u < left; ++u) {
                    cw |= (unsigned)certain[i + u] << (8 * u);
                    uw |= (unsigned)uncertain[i + u] << (8 * u);
                }
            }
#pragma unroll
            for (int u = 0; u < V; ++u) {
                if (u >= left) break;
                const int c = (int)((cw >> (8 * u)) & 255u), w = (int)((uw >> (8 * u)) & 255u);
                const bool out = c > thresh;
                outw |= (unsigned)out << (8 * u);
                if (!out && c + w > thresh) {                                     // the uncertain values could change the verdict: the caller looks again
                    const int at = atomicAdd(open_count, 1);
                    if (at < cap) open_rows[at] = i + u;
                }
            }
            uint8_t* np_ = nan_out + (size_t)nvar * mask_ld + i;                  // (zeros: a non-finite profile value makes the selection decline)
            uint8_t* op_ = outl_out + (size_t)nvar * mask_ld + i;
            if (V == 4 && left == V) {
                *reinterpret_cast<unsigned*>(np_) = 0u;
                *reinterpret_cast<unsigned*>(op_) = outw;
            } else {
                for (int u = 0; u < left; ++u) {
                    np_[u] = 0;
                    op_[u] = (uint8_t)(outw >> (8 * u));
                }
            }
        }
    }
}

}  // namespace

extern "C" int pem_campaign_masks_f64_dev(size_t n, int nvar, const double* const* vars, const double* q, int q_ld, int row25, int row75,
                                          double iqr_factor, uint8_t* nan_out,
                                          uint8_t* outl_out, size_t mask_ld, const uint8_t* row_certain, const uint8_t* row_uncertain, int thresh,
                                          int64_t* open_rows, int32_t* open_count, int cap, pem_stream_t stream) {
    if (nvar < 0 || nvar > CM_MAX_VARS) return pem::fail(PEM_ERR_INVALID_ARG, "pem_campaign_masks: 0 <= nvar <= %d", CM_MAX_VARS);
    if (!nan_out || !outl_out || (nvar && (!vars || !q))) return pem::fail(PEM_ERR_INVALID_ARG, "pem_campaign_masks: NULL array");
    if (nvar && (q_ld < nvar || row25 < 0 || row75 < 0)) return pem::fail(PEM_ERR_INVALID_ARG, "pem_campaign_masks: rows of q hold nvar values, q_ld apart");
    if ((row_certain != nullptr) != (row_uncertain != nullptr) || (row_certain && (!open_rows || !open_count || cap < 0)))
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_campaign_masks: the premask counts come with open_rows and open_count");
    if (mask_ld < n) return pem::fail(PEM_ERR_INVALID_ARG, "pem_campaign_masks: rows of the mask arrays shorter than n");
    if (n == 0) return PEM_OK;
    if (int rc = pem::check_device()) return rc;
    CampaignVars cv{};
    for (int k = 0; k < nvar; ++k) {
        if (!vars[k]) return pem::fail(PEM_ERR_INVALID_ARG, "pem_campaign_masks: NULL variable");
        cv.v[k] = vars[k];
    }
    int cus = 0;
    HIP_TRY(pem::device_cus(&cus));
    // four samples per thread and 4-byte mask stores when every row of the mask arrays starts on a 4-byte boundary
    const bool words = mask_ld % 4 == 0 && reinterpret_cast<uintptr_t>(nan_out) % 4 == 0 && reinterpret_cast<uintptr_t>(outl_out) % 4 == 0 &&
                       reinterpret_cast<uintptr_t>(row_certain) % 4 == 0 && reinterpret_cast<uintptr_t>(row_uncertain) % 4 == 0;
    const int v = words ? 4 : 1;
    long long blocks = (((long long)n + v - 1) / v + MBLOCK - 1) / MBLOCK;
    if (blocks > (long long)cus * 16) blocks = (long long)cus * 16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (words)
        hipLaunchKernelGGL(campaign_masks_kernel<4>, dim3((unsigned)blocks), dim3(MBLOCK), 0, st, (long long)n, nvar, cv, q, q_ld, row25, row75, iqr_factor,
                           nan_out, outl_out, mask_ld, row_certain, row_uncertain, thresh, (long long*)open_rows, (int*)open_count, cap);
    else
        hipLaunchKernelGGL(campaign_masks_kernel<1>, dim3((unsigned)blocks), dim3(MBLOCK), 0, st, (long long)n, nvar, cv, q, q_ld, row25, row75, iqr_factor,
                           nan_out, outl_out, mask_ld, row_certain, row_uncertain, thresh, (long long*)open_rows, (int*)open_count, cap);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

extern "C" int pem_row_masks_f64_dev(size_t n, int m, const double* data, size_t ld, const double* lo, const double* hi, uint8_t* nan_out,
                                     int32_t* outside_out, pem_stream_t stream) {
    if (m < 1 || m > PEM_ROW_MASKS_MAX_M) return pem::fail(PEM_ERR_INVALID_ARG, "pem_row_masks: 1 <= m <= %d entries per sample", PEM_ROW_MASKS_MAX_M);
    if (ld < (size_t)m) return pem::fail(PEM_ERR_INVALID_ARG, "pem_row_masks: leading dimension smaller than m");
    if (n == 0) return PEM_OK;
    if (!data || !lo || !hi || !nan_out || !outside_out) return pem::fail(PEM_ERR_INVALID_ARG, "pem_row_masks: NULL array");
    if (int rc = pem::check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long rpw = m <= 64 ? 64 / m : 1, groups = ((long long)n + rpw - 1) / rpw;
    const int rows = m <= 64 ? MROWS : (m <= 128 ? 8 : (m <= 256 ? 4 : 2));   // row groups a wave takes at a time (the kernels' ROWS)
    long long blocks = (groups + (long long)MWAVES * rows - 1) / ((long long)MWAVES * rows);
    int cus = 0;
    HIP_TRY(pem::device_cus(&cus));
    const long long cap = (long long)cus * 8;                             // 32 waves per CU: nothing but loads in flight hides the latency
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks), blk(MBLOCK);
    if (m <= 64)
        hipLaunchKernelGGL(row_masks_narrow_kernel, grid, blk, 0, st, (long long)n, m, ld, data, lo, hi, nan_out, outside_out);
    else if (m <= 128)
        hipLaunchKernelGGL(row_masks_wide_kernel<2>, grid, blk, 0, st, (long long)n, m, ld, data, lo, hi, nan_out, outside_out);
    else if (m <= 256)
        hipLaunchKernelGGL(row_masks_wide_kernel<4>, grid, blk, 0, st, (long long)n, m, ld, data, lo, hi, nan_out, outside_out);
    else
        hipLaunchKernelGGL(row_masks_wide_kernel<8>, grid, blk, 0, st, (long long)n, m, ld, data, lo, hi, nan_out, outside_out);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}
