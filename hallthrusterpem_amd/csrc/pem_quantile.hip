// pem_quantile.hip -- per-column order statistics over the SAMPLE axis of a row-major [n][m] array (gfx950): the percentiles
// behind the NaN / interquartile-range masks of scripts/gen_data.py:125-174 (`np.percentile(arr, 25 | 75, axis=0)`) and the
// 5 / 50 / 95 % bands of scripts/pem_v0/monte_carlo.py:363-658, at forward-UQ sizes: 1e7 samples x 91 angles is 7.3 GB, and
// a sort-based quantile (torch.quantile) refuses more than 2^24 values per column and sorts everything to pick two of them.
//
// Exact selection, not an estimate: the caller (drivers.column_percentiles) hands over the RANKS numpy's method 'linear' reads
// -- floor((n - 1) q) and the one above -- with its interpolation weight, this file returns x_(rank) for every column and
// applies numpy's _lerp, so the result equals np.percentile bit for bit (NaN in a column -> NaN, as there).
//
// Four streaming passes over the data, each coalesced (a lane owns fixed columns: consecutive lanes read consecutive
// addresses of a row, or of several short rows) and each bound by HBM:
//   1. min / max per column of the order-preserving integer image of the values (and whether a NaN is present);
//   2. a histogram per column over [min, max] (BINS1 bins, counted in LDS, merged with one atomic per bin and workgroup);
//      -> for every wanted rank: the bin that holds it and its rank inside the bin;
//   3. a second histogram per wanted rank inside its bin (BINS2 sub-bins);
//      -> the sub-bin and the rank inside it: by now 1 / (BINS1 BINS2) of a column's values are left per rank;
//   4. those values are copied out (exact sizes are known from pass 3; ranks that share a sub-bin share a list).
// Then one workgroup per list sorts it in LDS and reads the ranks off; a list too long for LDS -- heavy ties, e.g. the
// 1e-20 profile of invalid samples -- is narrowed by further histograms over the list itself until it fits or is one value.
// Monotone binning is all the method needs: a bin index that never decreases with the key, so that everything in a lower
// bin is <= everything in a higher one and equal keys share a bin.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "pem_common.h"
#include "pem_hip.h"
#include "pem_qfused.h"

namespace {

typedef unsigned long long u64;
constexpr int QBLOCK = 512;               // two waves per SIMD share one workgroup's LDS histogram
constexpr int QWAVES = QBLOCK / 64;
constexpr int LDS_WORDS = 36864;          // 144 KB of 32-bit counters per workgroup
constexpr int SORT_CAP = 4096;            // keys a workgroup's LDS sort buffer holds (32 KB)
constexpr int SORT_DIRECT = 512;          // ... and sorts without narrowing the list first: a bitonic sort of 4096 keys is 78 passes with a
                                          // barrier each (80 us for the 910 lists of a campaign), one 1024-bin narrowing round three
constexpr int MAX_NC = 4;                 // columns per lane: m <= 256
constexpr int UNROLL = 8;                 // row groups a wave requests before it consumes any (loads in flight per lane: UNROLL x NC)
static_assert(PEM_QUANTILE_MAX_Q == 6, "hist2_kernel / compact_kernel are instantiated for 2, 4, ... 12 ranks per column");

// order-preserving image of a double (NaN excluded by the callers): negative values reversed, sign bit flipped
__device__ __forceinline__ u64 key_of(double x) {
    const u64 b = (u64)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(u64 k) {
    const u64 b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
// Binning of the streaming passes, in integer arithmetic (a u64 -> f64 conversion is six instructions on this chip, and the
// passes over 7 GB were bound by them): d = (k - lo) >> shift fits 31 bits, bin = floor(d * mult / 2^32) with
// mult = floor(2^32 BINS / (D + 1)), D = (hi - lo) >> shift -- never decreases with the key and stays below BINS; the low
// 32 bits of the product are the position inside the bin, scaled to BINS2 sub-bins the same way.
struct Scale {
    u64 lo;
    unsigned mult;
    int shift;
};
__device__ __forceinline__ int bin_of(u64 k, const Scale& s, unsigned& frac) {
    const unsigned d = (unsigned)((k - s.lo) >> s.shift);
    const u64 p = (u64)d * s.mult;
    frac = (unsigned)p;
    return (int)(p >> 32);
}
__device__ __forceinline__ int subbin_of(unsigned frac, int bins2) { return (int)__umulhi(frac, (unsigned)bins2); }
// The keys of bin b (bins2 = 1, g = b) or of sub-bin (b, sb) (g = b bins2 + sb) of a column's binning, [klo, khi], by bisection on
// the monotone map d -> bin(d) bins2 + sub-bin(d).  The streaming passes test a value's HIGH word against these ranges (two 32-bit
// instructions per wanted bin) before they bin it in full: a value outside every wanted bin -- all but a few per cent -- costs
// neither the 64-bit subtraction nor the quarter-rate multiply of bin_of.  An empty bin comes out as klo > khi.
__device__ __forceinline__ void keys_of_bin(const Scale& sc, u64 kmax, int bins2, long long g, u64& klo, u64& khi) {
    const u64 D = kmax >= sc.lo ? (kmax - sc.lo) >> sc.shift : 0;
    auto first = [&](long long G) {
        u64 lo = 0, hi = D + 1;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1, prod = mid * sc.mult;
            const long long gd = (long long)(prod >> 32) * bins2 + (bins2 > 1 ? (long long)__umulhi((unsigned)prod, (unsigned)bins2) : 0);
            if (gd >= G) hi = mid;
            else lo = mid + 1;
        }
        return lo;
    };
    const u64 d0 = first(g), d1 = first(g + 1);
    klo = sc.lo + (d0 << sc.shift);
    khi = d1 > D ? kmax : sc.lo + (d1 << sc.shift) - 1;
    if (d0 > D || d0 >= d1) {
        klo = ~0ull;
        khi = 0ull;
    }
}
// the select kernel's own refinement (rare, over short lists): bins in double arithmetic over any key range
__device__ __forceinline__ int bin_of_d(u64 k, u64 lo, double inv, int bins) {
    const int b = (int)((double)(k - lo) * inv);
    return b < bins - 1 ? b : bins - 1;
}

struct Column {          // per column
    u64 kmin, kmax;
    int has_nan, pad;    // has_nan: the number of NaNs the min / max pass met (the pilot form's counting pass only flags: 1)
    unsigned mult;       // floor(2^32 BINS1 / (((kmax - kmin) >> shift) + 1))
    int shift;           // so that (kmax - kmin) >> shift < 2^31
};
struct Target {          // per (column, wanted rank)
    u64 rank;            // in: 0-based rank in the column; then the rank inside the current bin / list
    u64 count;           // values in the target's sub-bin (= length of its list)
    u64 offset;          // start of its list in the candidate buffer
    u64 cursor;          // append position during pass 4
    u64 answer;          // the key x_(rank)
    int bin1, bin2;
    int owner;           // index of the target (of this column) whose list this one shares, or its own index
    int done;            // answer already known (constant column)
};

// lane -> the columns it owns.  m <= 64: a wave instruction covers rpw = 64 / m whole rows (lane = row-in-group * m + column);
// m > 64: one row per wave instruction and chunk, column = lane + 64 chunk.
struct Lanes {
    int rpw, active, col0, rsub;
    size_t cs = 1;       // elements between consecutive columns of a row (1: row-major [n][ld]; n with ld = 1: the transposed [m][n])
    __device__ Lanes(int m, int lane, size_t col_stride = 1) : cs(col_stride) {
        if (m <= 64) {
            rpw = 64 / m;
            active = lane < rpw * m;
            col0 = lane % m;
            rsub = lane / m;
        } else {
            rpw = 1;
            active = 1;
            col0 = lane;
            rsub = 0;
        }
    }
};

// Every value of the array once: f(j, column, x) for the lane's j-th column.  A wave takes UNROLL consecutive row groups at a
// time and requests all their values before it consumes the first (with one workgroup of 144 KB of LDS per CU the memory
// latency has to be covered inside the wave).
template <int NC, int U = UNROLL, class F>
__device__ __forceinline__ void stream_values(long long n, int m, size_t ld, const double* __restrict__ data, const Lanes& L, F f) {
    const long long wave = (long long)blockIdx.x * QWAVES + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * QWAVES;
    const long long groups = (n + L.rpw - 1) / L.rpw;
    for (long long g0 = wave * U; g0 < groups; g0 += nwaves * U) {
        double x[U][NC];
        bool ok[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = (g0 + u) * L.rpw + L.rsub;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = L.col0 + 64 * j;
                ok[u][j] = L.active && c < m && row < n;
                x[u][j] = ok[u][j] ? data[(size_t)row * ld + (size_t)c * L.cs] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < NC; ++j)
                if (ok[u][j]) f(j, L.col0 + 64 * j, x[u][j]);
        }
    }
}

struct Wanted {          // the call's quantiles travel in the kernel arguments (nq <= 3): no host-to-device copy to wait for
    u64 prev[PEM_QUANTILE_MAX_Q], next[PEM_QUANTILE_MAX_Q];
    double gamma[PEM_QUANTILE_MAX_Q];
};

__global__ void init_columns_kernel(Column* col, Target* tg, int m, int nq, Wanted w) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < m) {
        col[c].kmin = ~0ull;
        col[c].kmax = 0ull;
        col[c].has_nan = 0;
        for (int q = 0; q < nq; ++q) {
            Target t = {};
            t.rank = w.prev[q];
            tg[c * 2 * nq + 2 * q] = t;
            t.rank = w.next[q];
            tg[c * 2 * nq + 2 * q + 1] = t;
        }
    }
}

// ---- pass 1: min / max / NaN per column ------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(QBLOCK) void minmax_kernel(long long n, int m, size_t ld, size_t cs, const double* __restrict__ data, Column* __restrict__ col) {
    const Lanes L(m, threadIdx.x & 63, cs);
    u64 lo[NC], hi[NC];
    int nan[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        lo[j] = ~0ull;
        hi[j] = 0ull;
        nan[j] = 0;
    }
    stream_values<NC>(n, m, ld, data, L, [&](int j, int, double x) {
        if (x != x) nan[j] += 1;                   // (counted: the histogram's total is checked against n - NaNs, decide1_kernel)
        else {
            const u64 k = key_of(x);
            lo[j] = k < lo[j] ? k : lo[j];
            hi[j] = k > hi[j] ? k : hi[j];
        }
    });
    // workgroup first (LDS), then one global atomic per column and workgroup: with few columns every lane of the grid owns
    // the same ones, and 2.6e5 lanes queueing on one address cost 7 ms
    __shared__ u64 s_lo[64 * MAX_NC], s_hi[64 * MAX_NC];
    __shared__ int s_nan[64 * MAX_NC];
    for (int c = threadIdx.x; c < m; c += QBLOCK) {
        s_lo[c] = ~0ull;
        s_hi[c] = 0ull;
        s_nan[c] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        if (L.active && c < m) {
            if (lo[j] <= hi[j]) {
                atomicMin(&s_lo[c], lo[j]);
                atomicMax(&s_hi[c], hi[j]);
            }
            if (nan[j]) atomicAdd(&s_nan[c], nan[j]);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < m; c += QBLOCK) {
        if (s_lo[c] <= s_hi[c]) {
            atomicMin(&col[c].kmin, s_lo[c]);
            atomicMax(&col[c].kmax, s_hi[c]);
        }
        if (s_nan[c]) atomicAdd(&col[c].has_nan, s_nan[c]);
    }
}

__global__ void scale_columns_kernel(Column* col, int m, int bins1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < m) {
        const Column k = col[c];
        const u64 span = k.kmax >= k.kmin ? k.kmax - k.kmin : 0;
        int shift = 0;
        while ((span >> shift) >> 31) ++shift;
        col[c].shift = shift;
        const u64 mult = (((u64)bins1) << 32) / ((span >> shift) + 1);        // >= 2^32 when there are fewer keys than bins:
        col[c].mult = mult > 0xffffffffull ? 0xffffffffu : (unsigned)mult;      // capped (two keys may then share a bin: still monotone)
    }
}

// ---- pass 2: histogram per column ------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(QBLOCK) void hist1_kernel(long long n, int m, size_t ld, size_t cs, const double* __restrict__ data, const Column* __restrict__ col,
                                                        int bins1, unsigned* __restrict__ hist1) {
    extern __shared__ unsigned lds_hist[];
    for (int i = threadIdx.x; i < m * bins1; i += QBLOCK) lds_hist[i] = 0;
    __syncthreads();
    const Lanes L(m, threadIdx.x & 63, cs);
    Scale sc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        const bool on = L.active && c < m;
        sc[j].lo = on ? col[c].kmin : 0;
        sc[j].mult = on ? col[c].mult : 0;
        sc[j].shift = on ? col[c].shift : 0;
    }
    stream_values<NC>(n, m, ld, data, L, [&](int j, int c, double x) {
        if (x == x) {
            unsigned frac;
            atomicAdd(&lds_hist[c * bins1 + bin_of(key_of(x), sc[j], frac)], 1u);
        }
    });
    __syncthreads();
    for (int i = threadIdx.x; i < m * bins1; i += QBLOCK) {
        const unsigned v = lds_hist[i];
        if (v) atomicAdd(&hist1[i], v);
    }
}

// The bin of a histogram that holds `rank` (0-based among the counted values), searched by one WAVE: every lane sums a
// chunk of the bins, the lanes' sums are scanned with shuffles, the lane whose chunk holds the rank walks it.  (One thread
// walking 4096 bins in global memory took 0.4-0.9 ms per launch -- more than the four passes over a scalar QoI's data.)
struct Found {
    int bin;
    u64 before;      // values in the bins below
    unsigned count;  // values in the bin
};
__device__ __forceinline__ Found find_bin(const unsigned* __restrict__ h, int bins, u64 rank, int lane) {
    const int per = (bins + 63) / 64, b0 = lane * per;
    u64 mine = 0;
    for (int b = b0; b < b0 + per && b < bins; ++b) mine += h[b];
    u64 incl = mine;
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const u64 up = __shfl_up(incl, sh);
        if (lane >= sh) incl += up;
    }
    const unsigned long long holds = __ballot(incl > rank);
    const int src = holds ? __builtin_ctzll(holds) : 63;
    u64 cum = __shfl(incl - mine, src);
    const int s0 = src * per;
    int bin = s0;
    unsigned cnt = 0;
    if (lane == src) {
        int last = s0 + per < bins ? s0 + per : bins;
        for (; bin < last - 1; ++bin) {
            if (cum + h[bin] > rank) break;
            cum += h[bin];
        }
        if (bin > bins - 1) bin = bins - 1;
        cnt = h[bin];
    }
    Found f;
    f.bin = __shfl(bin, src);
    f.before = __shfl(cum, src);
    f.count = (unsigned)__shfl((int)cnt, src);
    return f;
}

// one wave per (column, target): the bin that holds the rank
// (and an invariant while the histogram is at hand -- ADVICE r3: an LDS atomic lost in the counting pass would shift a rank silently,
// both later passes agreeing on the wrong count: the column's bins must add up to its rows minus its NaNs, or the call fails)
__global__ __launch_bounds__(64) void decide1_kernel(int m, int nt, const Column* __restrict__ col, const unsigned* __restrict__ hist1, int bins1,
                                                      Target* __restrict__ tg, long long rows, int* __restrict__ inconsistent) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int c = i / nt;
    if (i == c * nt) {
        u64 sum = 0;
        for (int b = lane; b < bins1; b += 64) sum += hist1[(size_t)c * bins1 + b];
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) sum += __shfl_xor(sum, sh);
        if (lane == 0 && sum != (u64)(rows - col[c].has_nan)) atomicExch(inconsistent, 1 + c);
    }
    Target t = tg[i];
    t.owner = i - c * nt;
    if (col[c].kmin >= col[c].kmax) {          // a constant column (or one without a finite value: its result is NaN anyway)
        t.done = 1;
        t.answer = col[c].kmin;
        t.bin1 = -1;
    } else {
        const Found f = find_bin(hist1 + (size_t)c * bins1, bins1, t.rank, lane);
        t.bin1 = f.bin;
        t.rank -= f.before;
    }
    if (lane == 0) tg[i] = t;
}

// ---- pass 3: histogram of every target's bin -----------------------------------------------------------------------------
template <int NC, int NT>
__global__ __launch_bounds__(QBLOCK) void hist2_kernel(long long n, int m, size_t ld, size_t cs, const double* __restrict__ data, const Column* __restrict__ col,
                                                        const Target* __restrict__ tg, int bins1, int bins2,
                                                        unsigned* __restrict__ hist2) {
    extern __shared__ unsigned lds_hist[];
    for (int i = threadIdx.x; i < m * NT * bins2; i += QBLOCK) lds_hist[i] = 0;
    __syncthreads();
    const Lanes L(m, threadIdx.x & 63, cs);
    Scale sc[NC];
    int tb[NC][NT];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        const bool on = L.active && c < m;
        sc[j].lo = on ? col[c].kmin : 0;
        sc[j].mult = on ? col[c].mult : 0;
        sc[j].shift = on ? col[c].shift : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // targets of a column often share a bin: each DISTINCT bin is counted once, under the first target that has it
            int b = on ? tg[c * NT + t].bin1 : -1;
#pragma unroll
            for (int u = 0; u < t; ++u)
                if (tb[j][u] == b || (on && tg[c * NT + u].bin1 == b)) b = -1;
            tb[j][t] = b;
        }
    }
    stream_values<NC>(n, m, ld, data, L, [&](int j, int c, double x) {
        if (x == x) {
            const u64 k = key_of(x);
            unsigned frac;
            const int b = bin_of(k, sc[j], frac);
            int hit = -1;                                   // at most one target holds a given bin (deduplicated above): one branch, not NT
#pragma unroll
            for (int t = 0; t < NT; ++t) hit = tb[j][t] == b ? t : hit;
            if (hit >= 0) atomicAdd(&lds_hist[(c * NT + hit) * bins2 + subbin_of(frac, bins2)], 1u);
        }
    });
    __syncthreads();
    for (int i = threadIdx.x; i < m * NT * bins2; i += QBLOCK) {
        const unsigned v = lds_hist[i];
        if (v) atomicAdd(&hist2[i], v);
    }
}

// one wave per (column, target): the sub-bin that holds the rank (the histogram sits under the column's first target with
// the same bin)
__global__ __launch_bounds__(64) void decide2_kernel(int nt, const unsigned* __restrict__ hist2, int bins2, Target* __restrict__ tg) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int c = i / nt, t = i - c * nt;
    Target T = tg[i];
    if (T.done) return;
    int first = t;
    for (int u = t - 1; u >= 0; --u)
        if (!tg[c * nt + u].done && tg[c * nt + u].bin1 == T.bin1) first = u;
    const Found f = find_bin(hist2 + (size_t)(c * nt + first) * bins2, bins2, T.rank, lane);
    T.bin2 = f.bin;
    T.rank -= f.before;
    T.count = f.count;
    if (lane == 0) tg[i] = T;
}

// ONE workgroup, one thread per column: which targets of a column share a list (same bin and sub-bin), and where the lists
// start in the candidate buffer (a scan over the columns in LDS)
__global__ __launch_bounds__(64 * MAX_NC) void layout_kernel(int m, int nt, Target* __restrict__ tg, u64* __restrict__ total) {
    __shared__ u64 start[64 * MAX_NC + 1];
    constexpr int NTMAX = 2 * PEM_QUANTILE_MAX_Q;
    const int c = threadIdx.x;
    // (the column's targets are read ONCE, side by side, into registers: walked through global memory -- a dependent load per
    // comparison, ~50 per column -- this one workgroup took 50 us of a campaign's critical path)
    int bin1[NTMAX], bin2[NTMAX], owner[NTMAX];
    bool done[NTMAX];
    u64 count[NTMAX];
    u64 mine = 0;
    if (c < m) {
#pragma unroll
        for (int t = 0; t < NTMAX; ++t) {
            const Target& T = tg[c * nt + (t < nt ? t : 0)];
            done[t] = t >= nt || T.done != 0;
            bin1[t] = T.bin1;
            bin2[t] = T.bin2;
            count[t] = T.count;
            owner[t] = t;
        }
#pragma unroll
        for (int t = 0; t < NTMAX; ++t) {
            if (done[t]) continue;
            bool found = false;
#pragma unroll
            for (int u = 0; u < NTMAX; ++u) {
                if (u < t && !found && !done[u] && bin1[u] == bin1[t] && bin2[u] == bin2[t]) {
                    owner[t] = owner[u];
                    found = true;
                }
            }
            if (owner[t] == t) mine += count[t];
        }
    }
    start[c + 1] = mine;
    if (c == 0) start[0] = 0;
    __syncthreads();
    if (c == 0) {
        for (int k = 1; k <= 64 * MAX_NC; ++k) start[k] += start[k - 1];          // 256 additions in LDS
        total[0] = start[m];
    }
    __syncthreads();
    if (c < m) {
        u64 off = start[c];
        u64 offs[NTMAX];
#pragma unroll
        for (int t = 0; t < NTMAX; ++t) {
            offs[t] = 0;
            if (done[t]) continue;
            if (owner[t] == t) {
                offs[t] = off;
                off += count[t];
            } else {
#pragma unroll
                for (int u = 0; u < NTMAX; ++u)
                    if (u < t && u == owner[t]) offs[t] = offs[u];
            }
            tg[c * nt + t].owner = owner[t];
            tg[c * nt + t].offset = offs[t];
        }
    }
}

// ---- pass 4: copy the candidates out ----------------------------------------------------------------------------------------
template <int NC, int NT>
__global__ __launch_bounds__(QBLOCK) void compact_kernel(long long n, int m, size_t ld, size_t cs, const double* __restrict__ data, const Column* __restrict__ col,
                                                          Target* __restrict__ tg, int bins1, int bins2, u64* __restrict__ cand) {
    const Lanes L(m, threadIdx.x & 63, cs);
    Scale sc[NC];
    int tb1[NC][NT], tb2[NC][NT];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        const bool on = L.active && c < m;
        sc[j].lo = on ? col[c].kmin : 0;
        sc[j].mult = on ? col[c].mult : 0;
        sc[j].shift = on ? col[c].shift : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool own = on && !tg[c * NT + t].done && tg[c * NT + t].owner == t;       // only list owners collect
            tb1[j][t] = own ? tg[c * NT + t].bin1 : -1;
            tb2[j][t] = own ? tg[c * NT + t].bin2 : -1;
        }
    }
    stream_values<NC>(n, m, ld, data, L, [&](int j, int c, double x) {
        if (x == x) {
            const u64 k = key_of(x);
            unsigned frac;
            const int b = bin_of(k, sc[j], frac);
            const int sb = subbin_of(frac, bins2);
            int hit = -1;                                   // list owners have distinct (bin, sub-bin) pairs: at most one matches
#pragma unroll
            for (int t = 0; t < NT; ++t) hit = (tb1[j][t] == b && tb2[j][t] == sb) ? t : hit;
            if (hit >= 0) {
                Target& T = tg[c * NT + hit];
                cand[T.offset + atomicAdd(&T.cursor, 1ull)] = k;
            }
        }
    });
}

// ---- the lists: one workgroup per (column, target) ----------------------------------------------------------------------------
__device__ void block_sort(u64* s, int np2) {      // bitonic, np2 a power of two <= SORT_CAP, all QBLOCK threads
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += QBLOCK) {
                const int p = i ^ j;
                if (p > i) {
                    const u64 a = s[i], b = s[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        s[i] = b;
                        s[p] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// x_(rank) of the keys list(0) .. list(len - 1) that lie in [0, top], by one workgroup: sorted in LDS when they fit, narrowed
// by 1024-bin histograms over the list itself until they do (heavy ties) or one key is left.  `list` is any accessor: the
// contiguous candidate list of the single-GPU path, or the padded per-rank pieces of the sharded one (padding = ~0 > top).
// (`short_list`, when given: set if the keys gathered for the sort are not as many as were counted a moment earlier -- a
// slot allocation lost, as the copy pass of the pilot form once did; the caller refuses the result.)
template <class List>
__device__ u64 select_from(const List& list, u64 len, u64 rank, u64 top, int* short_list = nullptr) {
    __shared__ u64 keys[SORT_CAP];
    __shared__ unsigned hist[1024];
    __shared__ u64 s_lo, s_hi, s_cnt, s_rank;
    __shared__ int s_bin;
    u64 lo = 0, hi = top;
    for (;;) {
        // the keys of the list inside [lo, hi]: how many, their smallest and largest
        if (threadIdx.x == 0) {
            s_lo = ~0ull;
            s_hi = 0;
            s_cnt = 0;
        }
        __syncthreads();
        u64 mn = ~0ull, mx = 0, cnt = 0;
        for (u64 i = threadIdx.x; i < len; i += QBLOCK) {
            const u64 k = list(i);
            if (k >= lo && k <= hi) {
                mn = k < mn ? k : mn;
                mx = k > mx ? k : mx;
                ++cnt;
            }
        }
        if (cnt) {
            atomicMin(&s_lo, mn);
            atomicMax(&s_hi, mx);
            atomicAdd(&s_cnt, cnt);
        }
        __syncthreads();
        lo = s_lo;
        hi = s_hi;
        const u64 inside = s_cnt;
        __syncthreads();
        if (lo >= hi) return lo;               // one value left (or, defensively, nothing)
        if (inside <= SORT_DIRECT) {
            int np2 = 1;
            while (np2 < (int)inside) np2 <<= 1;
            if (threadIdx.x == 0) s_cnt = 0;
            for (int i = threadIdx.x; i < np2; i += QBLOCK) keys[i] = ~0ull;
            __syncthreads();
            for (u64 i = threadIdx.x; i < len; i += QBLOCK) {
                const u64 k = list(i);
                if (k >= lo && k <= hi) keys[atomicAdd(&s_cnt, 1ull)] = k;
            }
            __syncthreads();
            if (short_list && threadIdx.x == 0 && s_cnt != inside) atomicExch(short_list, 1);
            block_sort(keys, np2);
            const u64 r = keys[rank < inside ? rank : inside - 1];
            __syncthreads();
            return r;
        }
        // more keys than a short sort takes (or than LDS holds): 1024-bin histogram over [lo, hi], keep the bin that holds the rank
        for (int i = threadIdx.x; i < 1024; i += QBLOCK) hist[i] = 0;
        __syncthreads();
        const double inv = 1024.0 / ((double)(hi - lo) + 1.0);
        for (u64 i = threadIdx.x; i < len; i += QBLOCK) {
            const u64 k = list(i);
            if (k >= lo && k <= hi) atomicAdd(&hist[bin_of_d(k, lo, inv, 1024)], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {                // the bin that holds the rank: one wave, 16 bins per lane, a scan over the lanes' sums
            const int lane = threadIdx.x;
            u64 mine = 0;
#pragma unroll
            for (int b = 0; b < 16; ++b) mine += hist[16 * lane + b];
            u64 incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u64 up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            const u64 before = incl - mine;
            // the first lane whose running total exceeds the rank owns it (the last lane, defensively, if none does)
            const bool owns = incl > rank && before <= rank;
            const u64 any = __ballot(owns);
            if (any ? owns : lane == 63) {
                u64 cum = before;
                int b = 16 * lane;
                for (; b < 16 * lane + 15; ++b) {
                    if (cum + hist[b] > rank) break;
                    cum += hist[b];
                }
                s_bin = b;
                s_rank = rank - cum;
            }
        }
        __syncthreads();
        const int keep = s_bin;
        rank = s_rank;
        // the keys of that bin form an interval of keys (monotone binning): its ends are found by the next round's min / max
        if (threadIdx.x == 0) {
            s_lo = ~0ull;
            s_hi = 0;
        }
        __syncthreads();
        mn = ~0ull;
        mx = 0;
        for (u64 i = threadIdx.x; i < len; i += QBLOCK) {
            const u64 k = list(i);
            if (k >= lo && k <= hi && bin_of_d(k, lo, inv, 1024) == keep) {
                mn = k < mn ? k : mn;
                mx = k > mx ? k : mx;
            }
        }
        if (mn <= mx) {
            atomicMin(&s_lo, mn);
            atomicMax(&s_hi, mx);
        }
        __syncthreads();
        lo = s_lo;
        hi = s_hi;
        __syncthreads();
    }
}

// (`incomplete`: a list whose copy pass delivered another number of values than the histogram counted -- the two passes bin
// the same way, so this says the code is wrong, not the data; the host refuses the result.  It has happened: a version of
// compact_bracket_kernel that parked its hits in LDS lost one value in a few thousand -- percentiles one rank off, in some runs;
// the cause was not found (the parking pattern alone does not lose anything: tools/microbench/lds_slot_alloc.hip) and the
// parking, no faster than appending on the spot, was dropped.)
__global__ __launch_bounds__(QBLOCK) void select_kernel(int nt, Target* __restrict__ tg, const u64* __restrict__ cand, int* __restrict__ incomplete) {
    Target& T = tg[blockIdx.x];
    if (T.done) return;
    if (threadIdx.x == 0 && T.owner == (int)(blockIdx.x % nt) && T.cursor != T.count && atomicExch(incomplete, 1) == 0) {
        u64* note = reinterpret_cast<u64*>(incomplete) + 1;               // which list, for the error message
        note[0] = blockIdx.x;
        note[1] = T.cursor;
        note[2] = T.count;
    }
    const u64* list = cand + T.offset;
    const u64 answer = select_from([list](u64 i) { return list[i]; }, T.count, T.rank, ~0ull, incomplete);
    if (threadIdx.x == 0) {
        T.answer = answer;
        T.done = 1;
    }
}

// out[q][c] = numpy's _lerp(x_(prev), x_(next), gamma); NaN where the column holds one
__global__ void finish_kernel(int m, int nq, const Column* __restrict__ col, const Target* __restrict__ tg, Wanted w,
                              double* __restrict__ out) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * nq) return;
    const int q = i / m, c = i - q * m;
    const double a = value_of(tg[c * 2 * nq + 2 * q].answer), b = value_of(tg[c * 2 * nq + 2 * q + 1].answer), t = w.gamma[q];
    const double diff = b - a;
    double r = a + diff * t;
    if (t >= 0.5) r = b - diff * (1.0 - t);
    if (col[c].has_nan || col[c].kmin > col[c].kmax) r = __builtin_nan("");
    out[(size_t)q * m + c] = r;
}

// ---- the pilot form (round 3): two passes over the data instead of four ------------------------------------------------------
// The selection above spends its first two passes finding, per wanted rank, a key interval that holds 1/BINS1 of the column.
// A strided subsample -- every PILOT-th row, 3 % of the data -- gives such an interval for the price of the four passes over
// the subsample alone: the order statistics of the subsample a few standard deviations either side of the wanted quantile
// bracket it ("the bracket" [lo, hi] per column and quantile; both ranks numpy reads for a quantile lie in one bracket).
// Then ONE pass counts, per bracket, the values below it and a histogram inside it (a value outside every bracket costs two
// compares per bracket and no LDS atomic -- pass 2 above pays one per value), the wanted ranks become (sub-bin, rank inside)
// exactly as in decide2_kernel, and ONE pass copies the few values of those sub-bins out; lists, sort and interpolation are the
// ones above.  Exactness does not rest on the subsample: the counts say whether a wanted rank really lies inside its bracket
// (below <= rank < below + inside), and if any does not -- data ordered so that the strided rows misrepresent it -- the call
// falls back to the four passes.  The subsample only decides how often that happens: with brackets 7 sigma wide, for
// exchangeable rows (Monte-Carlo, Latin hypercube and Saltelli designs), about once in 1e9 calls.
using pem::Bracket;     // per (column, quantile): the keys lo .. hi, binned on their HIGH WORDS (round 4; csrc/pem_qfused.h)
// Binning inside a bracket (round 4).  The counting pass met a value of SOME bracket in nearly every wave instruction (64 lanes
// x nq brackets x 1.3 %), so whatever a value inside a bracket costs, every value paid: with the sub-bin taken from the 64-bit
// offset k - lo (subtract, shift, quarter-rate multiply, per bracket) three quantiles ran at 1.35 ms per 1e7 x 91 and five at
// 3.0 ms -- instruction issue, not memory.  The sub-bin is now a function of the key's high word alone: t = kh - loh,
// bin = floor(t mult / 2^32) -- never decreasing with the key, so still a valid binning -- which a lane gets from the
// subtraction that also tells it whether the value lies below the bracket (the borrow) or inside (t <= words).  Only a key whose
// high word EQUALS that of lo or hi (one value in 1e5) is compared in full.  A bracket narrower than `bins` high words -- a
// column with a relative spread below 1e-5 -- puts several sub-bins' worth of values into one list; select_from narrows such a
// list by histograms of its own, as it does for ties.
// The fused form's pilot does not need the subsample's order statistics themselves: any key at or below x_(r_lo) will do as a
// bracket's lower end, any key at or above x_(r_hi) as its upper end (the counts of the full run prove the bracket, whatever it
// is).  So its run stops after the second histogram: a lower end becomes the FIRST key of the sub-bin that holds its rank, an
// upper end the LAST key of its sub-bin -- no candidate lists, no copy pass over the subsample, no sort (a sub-bin holds a few
// dozen of the subsample's ~4000 values between the ends).  Targets 2q / 2q + 1 are a bracket's lower / upper end.
__global__ __launch_bounds__(64) void pilot_edges_kernel(int total, int nt, int bins2, const Column* __restrict__ col, Target* __restrict__ tg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    Target T = tg[i];
    if (T.done) return;
    const Column C = col[i / nt];
    Scale sc;
    sc.lo = C.kmin;
    sc.mult = C.mult;
    sc.shift = C.shift;
    u64 klo, khi;
    keys_of_bin(sc, C.kmax, bins2, (long long)T.bin1 * bins2 + T.bin2, klo, khi);
    tg[i].answer = ((i % nt) & 1) ? khi : klo;
}

struct PilotEnds {       // a bracket end that would lie outside the subsample is open: the smallest / largest key
    int open_lo[PEM_QUANTILE_MAX_Q], open_hi[PEM_QUANTILE_MAX_Q];
};

// brackets from the order statistics of the subsample (targets 2q, 2q + 1 of the pilot run: the lower and the upper end).
// The ends are moved outwards to whole high words -- lo to the first key of its word, hi to the last key of its -- so that
// "below", "inside" and the sub-bin are all functions of a key's high word: any interval that holds the wanted ranks will do,
// and this one is at most two parts in a million wider.  The exception is a bracket whose ends are ONE key (the subsample's
// order statistics coincide: a constant column, or heavy ties such as the 1e-20 profile of invalid samples): it stays that key,
// is counted by comparing whole keys (mult = 0 marks it; `any_single` tells the counting pass to take its general loop) and
// answers its ranks without a list.
__global__ void brackets_kernel(int m, int nq, const Target* __restrict__ ptg, PilotEnds e, int bins, Bracket* __restrict__ br,
                                int* __restrict__ any_single) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * nq) return;
    const int c = i / nq, q = i - c * nq;
    u64 lo = e.open_lo[q] ? 0ull : ptg[c * 2 * nq + 2 * q].answer;
    u64 hi = e.open_hi[q] ? ~0ull : ptg[c * 2 * nq + 2 * q + 1].answer;
    if (hi < lo) hi = lo;                  // (a subsample column without a finite value: the column's result is NaN anyway)
    Bracket b;
    const bool single = lo == hi;
    if (!single) {
        lo &= 0xffffffff00000000ull;
        hi |= 0x00000000ffffffffull;
    }
    b.lo = lo;
    b.hi = hi;
    b.loh = (unsigned)(lo >> 32);
    b.words = (unsigned)(hi >> 32) - b.loh;
    const u64 mult = (((u64)bins) << 32) / ((u64)b.words + 1);
    b.mult = single ? 0u : (mult > 0xffffffffull ? 0xffffffffu : (unsigned)mult);
    b.pad = 0;
    br[i] = b;
    if (single) atomicExch(any_single, 1);
}

// the high word of key_of(x) from the high word of x alone (three instructions; the passes below decide nearly every value on it)
__device__ __forceinline__ unsigned key_high(double x) {
    const int bh = __double2hiint(x);
    return (unsigned)bh ^ ((unsigned)(bh >> 31) | 0x80000000u);
}
// the smallest t in [0, words + 1] whose bin is >= b (words + 1: none) -- the binning is monotone, so this is where sub-bin b begins
__device__ __forceinline__ u64 first_t_of_bin(unsigned mult, int b, unsigned words) {
    u64 lo = 0, hi = (u64)words + 1;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if ((int)((mid * mult) >> 32) >= b) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// pass A: per (column, quantile) the number of values below the bracket and a histogram of those inside it; NaN per column
// Up to 128 columns and three quantiles two workgroups share a CU (16 waves): the bracket histogram gets half the LDS (LDS_WORDS_A)
// and the registers have to fit twice; four columns per lane, or more quantiles, take a CU each.
constexpr int LDS_WORDS_A = 20224;        // 79 KB of 32-bit counters per workgroup
template <int NC, int NQ>
constexpr int bracket_waves_per_simd() { return NC <= 2 ? 4 : 2; }
template <int NC, int NQ>
__global__ __launch_bounds__(QBLOCK) __attribute__((amdgpu_waves_per_eu(bracket_waves_per_simd<NC, NQ>()))) void bracket_hist_kernel(long long n, int m, size_t ld, size_t cs, const double* __restrict__ data,
                                                               const Bracket* __restrict__ br, const int* __restrict__ any_single, int bins,
                                                               Column* __restrict__ col, u64* __restrict__ below, unsigned* __restrict__ hist) {
    extern __shared__ unsigned lds_hist[];                      // [m][NQ][bins] | below [m][NQ]
    unsigned* lds_below = lds_hist + m * NQ * bins;
    for (int i = threadIdx.x; i < m * NQ * (bins + 1); i += QBLOCK) lds_hist[i] = 0;
    __syncthreads();
    const Lanes L(m, threadIdx.x & 63, cs);
    unsigned bloh[NC][NQ], bwords[NC][NQ], bmult[NC][NQ], nbelow[NC][NQ];
    int nan[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        const bool on = L.active && c < m;
        nan[j] = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            bloh[j][q] = on ? br[c * NQ + q].loh : 0;
            bwords[j][q] = on ? br[c * NQ + q].words : 0;
            bmult[j][q] = on ? br[c * NQ + q].mult : 0;
            nbelow[j][q] = 0;
        }
    }
    // Per value and bracket: one subtraction of high words -- its borrow says "below", its result t <= words says "inside" and is
    // the sub-bin's argument; nothing else.  A NaN is flagged and otherwise counted as whatever its bits say: the column's result
    // is NaN whatever the counts are.  The general loop (taken by every wave when some bracket is a single key) compares whole
    // keys for those brackets.
    // (four brackets or more per column: four row groups in flight per wave instead of eight -- the brackets' registers and sixteen
    // values did not fit the 128 registers of four waves per SIMD: 9-41 spilled)
    constexpr int U = NC >= 4 ? 4 : ((NC >= 2 && NQ >= 6) ? 2 : ((NC >= 2 && NQ >= 4) ? 4 : UNROLL));
    if (*any_single == 0) {
        stream_values<NC, U>(n, m, ld, data, L, [&](int j, int c, double x) {
            const bool num = x == x;
            nan[j] |= num ? 0 : 1;
            const unsigned kh = key_high(x);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                unsigned t;
                nbelow[j][q] += __builtin_sub_overflow(kh, bloh[j][q], &t) ? 1u : 0u;
                // (a borrow leaves t above every `words`: 2^32 - loh + kh > hih - loh)
                if (t <= bwords[j][q] && num) atomicAdd(&lds_hist[(c * NQ + q) * bins + (int)__umulhi(t, bmult[j][q])], 1u);
            }
        });
    } else {
        stream_values<NC, U>(n, m, ld, data, L, [&](int j, int c, double x) {
            const bool num = x == x;
            nan[j] |= num ? 0 : 1;
            const unsigned kh = key_high(x);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                unsigned t;
                nbelow[j][q] += __builtin_sub_overflow(kh, bloh[j][q], &t) ? 1u : 0u;
                if (t <= bwords[j][q] && num) {
                    if (bmult[j][q] != 0) {
                        atomicAdd(&lds_hist[(c * NQ + q) * bins + (int)__umulhi(t, bmult[j][q])], 1u);
                    } else {                                    // a bracket of one key: its word holds keys below, equal and above
                        const u64 k = key_of(x), only = br[c * NQ + q].lo;
                        if (k < only) nbelow[j][q] += 1u;
                        else if (k == only) atomicAdd(&lds_hist[(c * NQ + q) * bins], 1u);
                    }
                }
            }
        });
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        if (L.active && c < m) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (nbelow[j][q]) atomicAdd(&lds_below[c * NQ + q], nbelow[j][q]);
            if (nan[j]) atomicOr(&col[c].has_nan, 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < m * NQ * bins; i += QBLOCK) {
        const unsigned v = lds_hist[i];
        if (v) atomicAdd(&hist[i], v);
    }
    for (int i = threadIdx.x; i < m * NQ; i += QBLOCK) {
        const unsigned v = lds_below[i];
        if (v) atomicAdd(&below[i], (u64)v);
    }
}

// one wave per (column, target): is the rank inside its bracket, and if so in which sub-bin (decide2_kernel's role; bin1 = the
// quantile, so that layout_kernel's "same bin and sub-bin" is "same bracket and sub-bin")
__global__ __launch_bounds__(64) void decide_bracket_kernel(int nt, Column* __restrict__ col, const Bracket* __restrict__ br,
                                                             const u64* __restrict__ below, const unsigned* __restrict__ hist, int bins,
                                                             Target* __restrict__ tg, int* __restrict__ outside) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int c = i / nt, t = i - c * nt, q = t >> 1, nq = nt >> 1;
    Target T = tg[i];
    if (t == 0 && lane == 0) {             // finish_kernel reads kmin > kmax as "no finite value": every such column holds a NaN here
        col[c].kmin = 0ull;
        col[c].kmax = ~0ull;
    }
    T.owner = t;
    if (col[c].has_nan) {                   // the column's result is NaN whatever the ranks are
        T.done = 1;
        T.answer = 0;
    } else {
        const Bracket b = br[c * nq + q];
        const u64 lowc = below[c * nq + q];
        const u64 r = T.rank - lowc;
        const Found f = find_bin(hist + (size_t)(c * nq + q) * bins, bins, r, lane);
        if (T.rank < lowc || f.before + f.count <= r) {      // not in the bracket: the call falls back to the four passes
            T.done = 1;
            T.answer = 0;
            if (lane == 0) atomicExch(outside, 1);
        } else if (b.lo == b.hi) {         // a bracket of one key (a constant column): the answer is that key
            T.done = 1;
            T.answer = b.lo;
        } else {
            T.bin1 = q;
            T.bin2 = f.bin;
            T.rank = r - f.before;
            T.count = f.count;
        }
    }
    if (lane == 0) tg[i] = T;
}

// pass B: copy out the values of the chosen sub-bins.  (Parking the hits in LDS and appending them after the last row, so that no
// wave waits for a global atomic in mid-stream, measured the same 1.67 ms as appending on the spot: not kept.)
template <int NC, int NQ>
__global__ __launch_bounds__(QBLOCK) void compact_bracket_kernel(long long n, int m, size_t ld, size_t cs, const double* __restrict__ data,
                                                                  const Bracket* __restrict__ br, Target* __restrict__ tg,
                                                                  u64* __restrict__ cand) {
    constexpr int NT = 2 * NQ;
    const Lanes L(m, threadIdx.x & 63, cs);
    // per (column, quantile): the range of high words (as offsets t from the bracket's first) of the sub-bins its list owners
    // collect (two adjacent ranks: one sub-bin, or two neighbours), found from the binning itself; a value is tested against it
    // with one subtraction and one comparison
    unsigned rloh[NC][NQ], rwords[NC][NQ];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        const bool on = L.active && c < m;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            u64 lo = ~0ull, hi = 0ull;                          // as high words (64-bit so that "none" fits)
#pragma unroll
            for (int t = 2 * q; t < 2 * q + 2; ++t) {
                const bool own = on && !tg[c * NT + t].done && tg[c * NT + t].owner == t;   // only list owners collect
                if (own) {
                    const Bracket B = br[c * NQ + q];
                    const int b2 = tg[c * NT + t].bin2;
                    const u64 t0 = first_t_of_bin(B.mult, b2, B.words), t1 = first_t_of_bin(B.mult, b2 + 1, B.words);
                    const u64 h0 = (u64)B.loh + t0, h1 = (u64)B.loh + (t1 > (u64)B.words ? (u64)B.words : t1 - 1);
                    lo = h0 < lo ? h0 : lo;
                    hi = h1 > hi ? h1 : hi;
                }
            }
            if (lo > hi) {                                      // no owner: a range no high word lies in
                rloh[j][q] = 0xffffffffu;
                rwords[j][q] = 0u;
            } else {
                rloh[j][q] = (unsigned)lo;
                rwords[j][q] = (unsigned)(hi - lo);
            }
        }
    }
    stream_values<NC>(n, m, ld, data, L, [&](int j, int c, double x) {
        const unsigned kh = key_high(x);
        bool near = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) near |= kh - rloh[j][q] <= rwords[j][q];
        if (near && x == x) {                                   // (about one wave instruction in ten)
            const u64 k = key_of(x);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (kh - rloh[j][q] <= rwords[j][q]) {
                    const Bracket B = br[c * NQ + q];
                    {                                           // (list owners' brackets end on whole words: the high word has decided)
                        const int b = (int)__umulhi(kh - B.loh, B.mult);
                        // (the two targets of a quantile share a list when they share the sub-bin: at most one of them owns it)
                        const Target& T0 = tg[c * NT + 2 * q];
                        const Target& T1 = tg[c * NT + 2 * q + 1];
                        const int hit = (!T0.done && T0.owner == 2 * q && T0.bin2 == b) ? 2 * q
                                        : ((!T1.done && T1.owner == 2 * q + 1 && T1.bin2 == b) ? 2 * q + 1 : -1);
                        if (hit >= 0) {
                            Target& T = tg[c * NT + hit];
                            cand[T.offset + atomicAdd(&T.cursor, 1ull)] = k;
                        }
                    }
                }
            }
        }
    });
}

// ---- the fused form (round 4): counts and records come from the kernel that PRODUCES the array (csrc/pem_qfused.h) ------------
// pass A's below-counts are already there; its histogram inside the brackets, and pass B's copy, run over the records -- the few
// per cent of the values that lie inside a bracket -- instead of over the array.

// a producer's counting launch assumes brackets that end on whole words (not a single key) and do not overlap within a column
__global__ void bracket_check_kernel(int m, int nq, const Bracket* __restrict__ br, int* __restrict__ unfit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * nq) return;
    const int c = i / nq, q = i - c * nq;
    const Bracket b = br[i];
    bool bad = b.mult == 0;
    for (int u = 0; u < nq; ++u) {
        if (u == q) continue;
        const Bracket o = br[c * nq + u];
        if ((u64)b.loh <= (u64)o.loh + o.words && (u64)o.loh <= (u64)b.loh + b.words) bad = true;
    }
    if (bad) atomicExch(unfit, 1);
}

// The premask's thresholds (csrc/pem_kernels.hip count_round): p25 and p75 lie in their brackets [A, B] and [C, D], so numpy's
// lo = p25 - f (p75 - p25) and hi = p75 + f (p75 - p25) (gen_data.py:163-168; every operation rounded on its own, every one of
// them monotone in p25 and p75 for f >= 0) lie in [lo_min, lo_max] and [hi_min, hi_max].  Per column the high words of the
// four ends' keys; `bad` when an end is zero (the key order splits -0 from +0, numpy's comparison does not), not finite, or f < 0.
__global__ void premask_bounds_kernel(int m, int nq, const Bracket* __restrict__ br, int q25, int q75, double f, uint4* __restrict__ thr,
                                      int* __restrict__ bad) {
#pragma clang fp contract(off)
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    const double A = value_of(br[c * nq + q25].lo), B = value_of(br[c * nq + q25].hi);
    const double C = value_of(br[c * nq + q75].lo), D = value_of(br[c * nq + q75].hi);
    const double iqr_max = D - A, iqr_min = C - B;
    const double lo_min = A - f * iqr_max, lo_max = B - f * iqr_min, hi_min = C + f * iqr_min, hi_max = D + f * iqr_max;
    const double v[4] = {lo_min, lo_max, hi_min, hi_max};
    bool ok = f >= 0.0;
    for (int i = 0; i < 4; ++i) ok = ok && __builtin_isfinite(v[i]) && v[i] != 0.0;
    if (!ok) atomicExch(bad, 1);
    thr[c] = make_uint4(key_high(lo_min), key_high(lo_max), key_high(hi_min), key_high(hi_max));
}

// the records the producer wrote are the values the record histogram counted (sums: record_reduce_kernel's)
__global__ void record_total_check_kernel(const u64* __restrict__ sums, int* __restrict__ inconsistent, const int* __restrict__ prod_flags) {
    // (a run the producer flagged -- overflow, a non-finite value -- is discarded anyway, and its records need not add up)
    if (threadIdx.x == 0 && sums[0] != sums[1] && !prod_flags[0] && !prod_flags[1]) atomicExch(inconsistent, -1);
}

constexpr int REC_LISTS_MAX = 64 * 2 * PEM_QUANTILE_MAX_Q;     // (column, quantile) pairs the record kernels keep a table of (m <= 128)

// the list (column * nq + quantile) of a record: the bracket of its column whose high words hold the key's (they do not overlap)
template <int NQ>
__device__ __forceinline__ int record_list(const uint4* __restrict__ s_br, const pem::Record& e, unsigned& t, unsigned& mult, bool& none) {
    const unsigned kh = (unsigned)(e.key >> 32);
    const int c0 = (int)e.col * NQ;
    int cq = c0;
    uint4 b[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) b[q] = s_br[c0 + q];
    t = kh - b[0].x;
    mult = b[0].z;
    none = t > b[0].y;
#pragma unroll
    for (int q = 1; q < NQ; ++q) {
        const unsigned tq = kh - b[q].x;
        if (tq <= b[q].y) {
            cq = c0 + q;
            t = tq;
            mult = b[q].z;
            none = false;
        }
    }
    return cq;
}

constexpr int REC_UNROLL = 4;   // records a thread has in flight

// (the record passes run one workgroup per CU -- its histogram takes the CU's LDS -- so the workgroup is 1024 threads: with 256 the
// passes were latency-bound at 2.5 TB/s over the records)
constexpr int RBLOCK = 1024;
template <int NQ>
__global__ __launch_bounds__(RBLOCK) void record_hist_kernel(const pem::Record* __restrict__ rec, const unsigned* __restrict__ rec_count, unsigned cap,
                                                              unsigned waves, const Bracket* __restrict__ br, int lists, int bins,
                                                              unsigned* __restrict__ hist) {
    extern __shared__ unsigned lds_hist[];                      // [lists][bins]
    __shared__ uint4 s_br[REC_LISTS_MAX];                       // {loh, words, mult, -}
    for (int i = threadIdx.x; i < lists * bins; i += RBLOCK) lds_hist[i] = 0;
    for (int i = threadIdx.x; i < lists; i += RBLOCK) s_br[i] = make_uint4(br[i].loh, br[i].words, br[i].mult, 0u);
    __syncthreads();
    for (unsigned w = blockIdx.x; w < waves; w += gridDim.x) {
        const unsigned cnt = rec_count[w] < cap ? rec_count[w] : cap;
        const pem::Record* r = rec + (size_t)w * cap;
        for (unsigned i0 = threadIdx.x; i0 < cnt; i0 += RBLOCK * REC_UNROLL) {
            pem::Record e[REC_UNROLL];
#pragma unroll
            for (int u = 0; u < REC_UNROLL; ++u) e[u] = r[i0 + u * RBLOCK < cnt ? i0 + u * RBLOCK : i0];
#pragma unroll
            for (int u = 0; u < REC_UNROLL; ++u) {
                unsigned t, mult;
                bool none;
                const int cq = record_list<NQ>(s_br, e[u], t, mult, none);
                // (none: not a value of any bracket -- a NaN's bits; the producer has flagged the run)
                if (!none && i0 + u * RBLOCK < cnt) atomicAdd(&lds_hist[cq * bins + (int)__umulhi(t, mult)], 1u);
            }
        }
    }
    __syncthreads();
    // the workgroup's own counts, whole (no atomics): summed over the workgroups by record_reduce_kernel, and read again -- per chosen
    // sub-bin -- by record_offsets_kernel, which gives every workgroup its own place in every list
    unsigned* mine = hist + (size_t)blockIdx.x * lists * bins;
    for (int i = threadIdx.x; i < lists * bins; i += RBLOCK) mine[i] = lds_hist[i];
}

// 64 cells x 4 slices of the workgroups per block, eight loads in flight per thread (one thread per cell walking the 256 partial
// histograms took 60-120 us of the critical path for 30 MB); the histogram's total and the producer's record count ride along
// (sums[0], sums[1]: record_total_check_kernel compares them)
__global__ __launch_bounds__(256) void record_reduce_kernel(const unsigned* __restrict__ part, int groups, int cells, unsigned* __restrict__ hist,
                                                            const unsigned* __restrict__ rec_count, unsigned cap, unsigned waves,
                                                            u64* __restrict__ sums) {
    __shared__ unsigned s_part[4][64];
    __shared__ u64 s_tot;
    const int cx = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cx;
    if (threadIdx.x == 0) s_tot = 0;
    unsigned sum = 0;
    if (i < cells) {
        int g = slice;
        for (; g + 28 < groups; g += 32) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(g + 4 * u) * cells + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; g < groups; g += 4) sum += part[(size_t)g * cells + i];
    }
    s_part[slice][cx] = sum;
    __syncthreads();
    if (slice == 0) {
        const unsigned all = s_part[0][cx] + s_part[1][cx] + s_part[2][cx] + s_part[3][cx];
        if (i < cells) {
            hist[i] = all;
            atomicAdd(&s_tot, (u64)all);
        }
    }
    if (blockIdx.x == 0) {
        u64 r = 0;
        for (unsigned w = threadIdx.x; w < waves; w += 256) r += rec_count[w] < cap ? rec_count[w] : cap;
        if (r) atomicAdd(&sums[1], r);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_tot) atomicAdd(&sums[0], s_tot);
}

// One wave per target that owns a list: where each workgroup of the copy pass appends its hits -- the exclusive prefix, over the
// workgroups, of their counts in the list's sub-bin (record_hist_kernel's own histograms).  The copy pass then needs no global
// atomic (it drew one per hit: 5.6e5 returning atomics on 910 addresses, 0.36 ms for 580 MB of records).
__global__ __launch_bounds__(64) void record_offsets_kernel(int nt, int bins, const Target* __restrict__ tg, const unsigned* __restrict__ part,
                                                             int groups, int cells, unsigned* __restrict__ woff) {
    const int i = blockIdx.x, lane = threadIdx.x, targets = gridDim.x;
    const Target T = tg[i];
    const int t = i % nt;
    const bool own = !T.done && T.owner == t;
    const int cell = (i / 2) * bins + (own ? T.bin2 : 0);      // targets 2q, 2q + 1 of column c belong to list c nq + q = i / 2
    unsigned base = 0;
    for (int g0 = 0; g0 < groups; g0 += 64) {
        const int g = g0 + lane;
        const unsigned v = (own && g < groups) ? part[(size_t)g * cells + cell] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const unsigned up = __shfl_up(incl, sh);
            if (lane >= sh) incl += up;
        }
        if (g < groups) woff[(size_t)g * targets + i] = base + incl - v;
        base += __shfl(incl, 63);
    }
}

template <int NQ>
__global__ __launch_bounds__(RBLOCK) void record_compact_kernel(const pem::Record* __restrict__ rec, const unsigned* __restrict__ rec_count, unsigned cap,
                                                                 unsigned waves, const Bracket* __restrict__ br, int lists, Target* __restrict__ tg,
                                                                 const unsigned* __restrict__ woff, u64* __restrict__ cand) {
    __shared__ uint4 s_br[REC_LISTS_MAX];                       // {loh, words, mult, -}
    __shared__ int2 s_bin[REC_LISTS_MAX];                       // the sub-bins the quantile's two targets collect (-1: not a list owner)
    __shared__ unsigned s_cur[2 * REC_LISTS_MAX];               // this workgroup's cursor in every target's list (from record_offsets_kernel)
    __shared__ u64 s_base[2 * REC_LISTS_MAX];                   // the lists' starts in `cand`
    const unsigned* my_off = woff + (size_t)blockIdx.x * 2 * lists;
    for (int i = threadIdx.x; i < lists; i += RBLOCK) {
        s_br[i] = make_uint4(br[i].loh, br[i].words, br[i].mult, 0u);
        const Target &T0 = tg[2 * i], &T1 = tg[2 * i + 1];     // targets 2q, 2q + 1 of column c sit at (c nq + q) 2
        s_bin[i] = make_int2((!T0.done && (T0.owner & 1) == 0) ? T0.bin2 : -1, (!T1.done && (T1.owner & 1) == 1) ? T1.bin2 : -1);
        s_cur[2 * i] = my_off[2 * i];
        s_cur[2 * i + 1] = my_off[2 * i + 1];
        s_base[2 * i] = T0.offset;
        s_base[2 * i + 1] = T1.offset;
    }
    __syncthreads();
    // (the same records as this workgroup counted in record_hist_kernel: same grid, same walk)
    for (unsigned w = blockIdx.x; w < waves; w += gridDim.x) {
        const unsigned cnt = rec_count[w] < cap ? rec_count[w] : cap;
        const pem::Record* r = rec + (size_t)w * cap;
        for (unsigned i0 = threadIdx.x; i0 < cnt; i0 += RBLOCK * REC_UNROLL) {
            pem::Record e[REC_UNROLL];
#pragma unroll
            for (int u = 0; u < REC_UNROLL; ++u) e[u] = r[i0 + u * RBLOCK < cnt ? i0 + u * RBLOCK : i0];
#pragma unroll
            for (int u = 0; u < REC_UNROLL; ++u) {
                unsigned t, mult;
                bool none;
                const int cq = record_list<NQ>(s_br, e[u], t, mult, none);
                if (none || i0 + u * RBLOCK >= cnt) continue;
                const int bin = (int)__umulhi(t, mult);
                const int2 want = s_bin[cq];
                const int hit = want.x == bin ? 0 : (want.y == bin ? 1 : -1);
                if (hit >= 0) cand[s_base[2 * cq + hit] + atomicAdd(&s_cur[2 * cq + hit], 1u)] = e[u].key;
            }
        }
    }
    __syncthreads();
    // what this workgroup appended, onto the lists' counters: select_kernel compares them with the counts (the "incomplete" check)
    for (int i = threadIdx.x; i < 2 * lists; i += RBLOCK) {
        const unsigned wrote = s_cur[i] - my_off[i];
        if (wrote) atomicAdd(&tg[i].cursor, (u64)wrote);
    }
}

// ---- the multi-rank form: histograms over caller-given key ranges -----------------------------------------------------------
// Samples sharded over ranks cannot be copied out and sorted in one place cheaply, but histograms add: every rank counts its
// own values inside the current range of every wanted rank, the counts are all-reduced (<= 144 KB), all ranks pick the same
// bin and narrow the range to it -- until a range is one key (hallthrusterpem_amd/percentiles.py drives the levels).
// Range r of column c: keys klo[c][r] .. khi[c][r]; d = (k - klo) >> shift with shift such that the span fits 31 bits;
// bin = d when the shifted span is smaller than `bins` (every key its own bin: the next range is a single key), else
// floor(d mult / 2^32), mult = min(2^32 - 1, floor(2^32 bins / (span' + 1))).  percentiles.py inverts exactly this map.
struct RangeScale {
    u64 lo, hi;
    unsigned mult;
    int shift, identity;
};
__device__ __forceinline__ RangeScale range_scale(u64 lo, u64 hi, int bins) {
    RangeScale r;
    r.lo = lo;
    r.hi = hi;
    const u64 span = hi >= lo ? hi - lo : 0;
    int shift = 0;
    while ((span >> shift) >> 31) ++shift;
    const u64 d = span >> shift;
    r.shift = shift;
    r.identity = d < (u64)bins;
    const u64 mult = (((u64)bins) << 32) / (d + 1);
    r.mult = mult > 0xffffffffull ? 0xffffffffu : (unsigned)mult;
    return r;
}

template <int NC, int NR>
__global__ __launch_bounds__(QBLOCK) void range_hist_kernel(long long n, int m, size_t ld, const double* __restrict__ data,
                                                             const u64* __restrict__ klo, const u64* __restrict__ khi, int bins,
                                                             unsigned* __restrict__ hist) {
    extern __shared__ unsigned lds_hist[];
    for (int i = threadIdx.x; i < m * NR * bins; i += QBLOCK) lds_hist[i] = 0;
    __syncthreads();
    const Lanes L(m, threadIdx.x & 63);
    RangeScale rs[NC][NR];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        const bool on = L.active && c < m;
#pragma unroll
        for (int r = 0; r < NR; ++r) rs[j][r] = on ? range_scale(klo[c * NR + r], khi[c * NR + r], bins) : range_scale(1, 0, bins);
    }
    stream_values<NC, (NC >= 4 && NR >= 4) ? 2 : UNROLL>(n, m, ld, data, L, [&](int j, int c, double x) {
        if (x == x) {
            const u64 k = key_of(x);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const RangeScale& q = rs[j][r];
                if (k >= q.lo && k <= q.hi) {
                    const unsigned d = (unsigned)((k - q.lo) >> q.shift);
                    const int b = q.identity ? (int)d : (int)(((u64)d * q.mult) >> 32);
                    atomicAdd(&lds_hist[(c * NR + r) * bins + b], 1u);
                }
            }
        }
    });
    __syncthreads();
    for (int i = threadIdx.x; i < m * NR * bins; i += QBLOCK) {
        const unsigned v = lds_hist[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// One level's decision on the device (one wave per range): the bin of the all-reduced histogram that holds the wanted rank,
// and the keys of that bin -- the inverse of range_scale's map, as percentiles.bin_interval states it -- become the next range.
__global__ __launch_bounds__(64) void range_narrow_kernel(int bins, const unsigned* __restrict__ hist, u64* __restrict__ klo,
                                                           u64* __restrict__ khi, long long* __restrict__ resid) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const u64 lo = klo[i], hi = khi[i];
    if (lo >= hi) return;                                     // already one key (or an empty column)
    const Found f = find_bin(hist + (size_t)i * bins, bins, (u64)resid[i], lane);
    if (lane != 0) return;
    const RangeScale q = range_scale(lo, hi, bins);
    const u64 b = (u64)f.bin, two32 = 1ull << 32;
    const u64 d_lo = q.identity ? b : (b * two32 + q.mult - 1) / q.mult;
    const u64 d_hi = q.identity ? b : ((b + 1) * two32 + q.mult - 1) / q.mult - 1;
    const bool last = d_hi >= ((hi - lo) >> q.shift);
    u64 nlo = lo + (d_lo << q.shift);
    u64 nhi = last ? hi : lo + (((d_hi + 1) << q.shift) - 1);
    if (nlo < lo) nlo = lo;
    if (nhi > hi) nhi = hi;
    klo[i] = nlo;
    khi[i] = nhi;
    resid[i] -= (long long)f.before;
}

__global__ void export_minmax_kernel(const Column* __restrict__ col, int m, u64* kmin, u64* kmax, int* has_nan) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < m) {
        kmin[c] = col[c].kmin;
        kmax[c] = col[c].kmax;
        has_nan[c] = col[c].has_nan ? 1 : 0;
    }
}

int pow2_at_most(long long x, int cap) {
    int p = 1;
    while (2LL * p <= x && 2 * p <= cap) p <<= 1;
    return p;
}

}  // namespace

// which way the last call of pem_quantiles_f64_dev went: 0 four passes, 1 pilot + two passes, 2 pilot, then four passes (a
// wanted rank outside its bracket)
static std::atomic<int> g_last_path{0};
extern "C" int pem_quantiles_last_path(void) { return g_last_path.load(); }

// `fused` (with `fused_ok`): the array does not exist (yet) -- `data` are its first ceil(n / pilot) rows, written by the producer,
// and pass A / pass B are the producer's counting launch and two passes over its records (csrc/pem_qfused.h).  *fused_ok = 0: the
// brackets were unfit for the producer, a rank fell outside its bracket, or the producer reported an overflow or a non-finite
// sample -- `out` is then not written and the caller takes the passes over the array itself.
// The buffers a selection keeps between calls (grow-only).  Two sets: calls are serialised on a set, and the second one lets the
// scalar QoIs of a campaign be selected on another host thread and stream while the fused form is busy with the profile's records.
namespace {
struct QWork {
    std::mutex mu;
    char* ws_buf = nullptr;
    size_t ws_cap = 0, cand_cap = 0, rec_cap = 0;
    u64* cand_buf = nullptr;
    pem::Record* rec_buf = nullptr;
    unsigned* part_buf = nullptr;              // fused form: the record histograms of every workgroup | their offsets into the lists
    size_t part_cap = 0;
    int ws_dev = -1;
};
QWork g_qwork[2];
}  // namespace

static int quantiles_impl(size_t n, int m, const double* data, size_t ld, size_t cs, int nq, const uint64_t* rank_prev,
                          const uint64_t* rank_next, const double* gamma, double* out, pem_stream_t stream, pem::FusedProducer* fused,
                          int* fused_ok, int slot = 0, const pem::SidePlan* plan = nullptr) {
    if (m < 1 || m > 64 * MAX_NC) return pem::fail(PEM_ERR_INVALID_ARG, "pem_quantiles: 1 <= m <= %d columns", 64 * MAX_NC);
    if (fused && (m > 128 || !fused_ok)) return pem::fail(PEM_ERR_INVALID_ARG, "pem_quantiles: the fused form takes up to 128 columns");
    if (cs < 1 || (cs == 1 ? ld < (size_t)m : (ld != 1 || cs < n)))
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_quantiles: rows of m columns (column stride 1, ld >= m) or columns of n rows (ld 1, column stride >= n)");
    if (nq < 1 || nq > (m <= 128 ? PEM_QUANTILE_MAX_Q : PEM_QUANTILE_MAX_Q_WIDE))
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_quantiles: 1 <= nq <= %d per call (%d for more than 128 columns)", PEM_QUANTILE_MAX_Q, PEM_QUANTILE_MAX_Q_WIDE);
    if (n == 0) return pem::fail(PEM_ERR_INVALID_ARG, "pem_quantiles: no samples");
    if (!data || !rank_prev || !rank_next || !gamma || !out) return pem::fail(PEM_ERR_INVALID_ARG, "pem_quantiles: NULL array");
    for (int q = 0; q < nq; ++q)
        if (rank_prev[q] >= n || rank_next[q] >= n || rank_prev[q] > rank_next[q])
            return pem::fail(PEM_ERR_INVALID_ARG, "pem_quantiles: ranks must satisfy prev <= next < n");
    if (int rc = pem::check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nt = 2 * nq;
    const int bins1 = pow2_at_most(LDS_WORDS / m, 4096), bins2 = pow2_at_most(LDS_WORDS / (m * nt), 4096);
    // (four quantiles or more: the brackets' registers leave room for one workgroup per CU anyway -- it gets all of the LDS)
    int binsA = pow2_at_most((m <= 128 ? LDS_WORDS_A : LDS_WORDS) / (m * nq) - 1, 4096);   // the pilot form's histogram inside a bracket (+ 1 counter)
    if (fused) binsA = pow2_at_most(LDS_WORDS / (m * nq), 4096);          // (the record passes: one workgroup per CU with all of its LDS)
    if (const char* e = getenv("PEM_QUANTILE_BINSA")) binsA = pow2_at_most(atoll(e) < binsA ? atoll(e) : binsA, 4096);

    // the pilot form: every `pilot`-th row brackets the wanted ranks (PEM_QUANTILE_PILOT: the stride, 0 = never;
    // PEM_QUANTILE_PILOT_MIN: the smallest n * m it is used for -- 1e7 x 1 takes 0.24 ms with the four passes and 0.32 with the
    // pilot's dozen launches in front of two, 1e7 x 3 the same either way, 6e5 x 91 0.60 against 0.56: profiles/quantile_pilot_r03.txt)
    long long pilot = 32, pilot_min = 1 << 25;
    if (const char* e = getenv("PEM_QUANTILE_PILOT")) pilot = atoll(e);
    if (const char* e = getenv("PEM_QUANTILE_PILOT_MIN")) pilot_min = atoll(e);
    if (fused) {                                                          // (the producer writes rows 0 .. ceil(n / 32) - 1 at most)
        pilot = 32;
        if (const char* e = getenv("PEM_FUSED_PILOT")) pilot = atoll(e) >= 32 ? atoll(e) : 32;
    }
    if (plan) pilot = 32;                                                 // (the rows the plan's caller has written: 0 .. ceil(n / 32) - 1)
    const bool use_pilot = fused ? true : plan ? (long long)n >= 4 * pilot : (pilot >= 2 && (long long)n * m >= pilot_min && (long long)n >= 4 * pilot);
    if (fused && (long long)n < 4 * pilot) return pem::fail(PEM_ERR_INVALID_ARG, "pem_quantiles: the fused form needs at least %lld rows", 4 * pilot);

    // workspace: columns | targets | hist1 | hist2 | total, outside | brackets | below | histA -- kept between calls (grow-only,
    // one per process; calls are serialised on it, and each one ends with a stream synchronisation before the next may touch it)
    const size_t b_col = sizeof(Column) * m, b_tg = sizeof(Target) * m * nt;
    const size_t b_h1 = sizeof(unsigned) * (size_t)m * bins1, b_h2 = sizeof(unsigned) * (size_t)m * nt * bins2;
    const size_t b_br = sizeof(Bracket) * m * nq, b_bl = sizeof(u64) * m * nq, b_hA = sizeof(unsigned) * (size_t)m * nq * binsA;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_tg = up(b_col), o_h1 = o_tg + up(b_tg), o_h2 = o_h1 + up(b_h1), o_tot = o_h2 + up(b_h2);
    unsigned fused_waves = 0;
    if (fused)
        if (int rc = fused->waves(nq, &fused_waves)) return rc;
    const size_t b_rc = sizeof(unsigned) * fused_waves;
    const size_t o_br = o_tot + 256, o_bl = o_br + up(b_br), o_hA = o_bl + up(b_bl), o_rc = o_hA + up(b_hA), o_pm = o_rc + up(b_rc);
    const size_t o_end = o_pm + up(sizeof(uint4) * (size_t)m);
    QWork& work = g_qwork[slot];
    std::lock_guard<std::mutex> lock(work.mu);
    char*& ws_buf = work.ws_buf;
    size_t &ws_cap = work.ws_cap, &cand_cap = work.cand_cap, &rec_cap = work.rec_cap, &part_cap = work.part_cap;
    u64*& cand_buf = work.cand_buf;
    pem::Record*& rec_buf = work.rec_buf;
    unsigned*& part_buf = work.part_buf;
    int& ws_dev = work.ws_dev;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != ws_dev || ws_cap < o_end) {
        if (ws_buf) (void)hipFree(ws_buf);
        if (cand_buf && dev != ws_dev) {
            (void)hipFree(cand_buf);
            cand_buf = nullptr;
            cand_cap = 0;
        }
        if (rec_buf && dev != ws_dev) {
            (void)hipFree(rec_buf);
            rec_buf = nullptr;
            rec_cap = 0;
        }
        if (part_buf && dev != ws_dev) {
            (void)hipFree(part_buf);
            part_buf = nullptr;
            part_cap = 0;
        }
        ws_buf = nullptr;
        ws_cap = 0;
        HIP_TRY(hipMalloc(&ws_buf, o_end));
        ws_cap = o_end;
        ws_dev = dev;
    }
    char* ws = ws_buf;
    Column* col = reinterpret_cast<Column*>(ws);
    Target* tg = reinterpret_cast<Target*>(ws + o_tg);
    unsigned* hist1 = reinterpret_cast<unsigned*>(ws + o_h1);
    unsigned* hist2 = reinterpret_cast<unsigned*>(ws + o_h2);
    u64* total = reinterpret_cast<u64*>(ws + o_tot);                       // total[0]: candidates; total[1] (as int): a rank outside its bracket
    int* outside = reinterpret_cast<int*>(total + 1);
    int* incomplete = reinterpret_cast<int*>(total + 2);                   // a list shorter or longer than counted (select_kernel), 4 words
    int* any_single = reinterpret_cast<int*>(total + 8);                   // a bracket of one key exists (brackets_kernel)
    int* unfit = reinterpret_cast<int*>(total + 9);                        // fused form: brackets a producer cannot count against
    int* prod_flags = unfit + 1;                                           // fused form: the producer's overflow / non-finite flags (2 ints)
    unsigned* rec_count = reinterpret_cast<unsigned*>(ws + o_rc);
    uint4* pm_thr = reinterpret_cast<uint4*>(ws + o_pm);
    int* pm_bad = unfit + 3;                                               // fused form: premask bounds unfit
    int* inconsistent = unfit + 4;                                         // a histogram's total is not the number of values counted (1 + column)
    Bracket* br = reinterpret_cast<Bracket*>(ws + o_br);
    u64* below = reinterpret_cast<u64*>(ws + o_bl);
    unsigned* histA = reinterpret_cast<unsigned*>(ws + o_hA);
    auto cleanup = [&](int code) {
        (void)hipStreamSynchronize(st);
        return code;
    };
#define Q_TRY(expr)                                                                                          \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return cleanup(pem::fail(PEM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_))); \
    } while (0)

    Wanted w{};
    for (int q = 0; q < nq; ++q) {
        w.prev[q] = rank_prev[q];
        w.next[q] = rank_next[q];
        w.gamma[q] = gamma[q];
    }
    const int cblocks = (m + 63) / 64;
    const int nc = m <= 64 ? 1 : (m <= 128 ? 2 : 4);
    const size_t lds1 = (size_t)m * bins1 * 4, lds2 = (size_t)m * nt * bins2 * 4, ldsA = (size_t)m * nq * (binsA + 1) * 4;
    const dim3 blk(QBLOCK);
    auto grid_for = [&](size_t rows) {
        const long long rpw = m <= 64 ? 64 / m : 1, groups = ((long long)rows + rpw - 1) / rpw;
        long long blocks = (groups + (long long)QWAVES * UNROLL - 1) / ((long long)QWAVES * UNROLL);
        if (blocks > 256 * 2) blocks = 256 * 2;
        return dim3((unsigned)blocks);
    };
    auto grow_candidates = [&](u64 need) -> hipError_t {
        if (cand_cap >= need + 1) return hipSuccess;
        if (cand_buf) (void)hipFree(cand_buf);
        cand_buf = nullptr;
        cand_cap = 0;
        // (a quarter more than asked for: the lists of two campaigns differ by a few values, and a hipFree + hipMalloc in the middle of
        // a call costs a device synchronisation and a millisecond)
        const u64 room = need + need / 4 + 1024;
        const hipError_t e = hipMalloc(&cand_buf, (size_t)room * sizeof(u64));
        if (e == hipSuccess) cand_cap = room;
        return e;
    };
#define Q_LDS(KERN) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define Q_BY_NC(CALL)              \
    do {                           \
        if (nc == 1) { CALL(1); }  \
        else if (nc == 2) { CALL(2); } \
        else { CALL(4); }          \
    } while (0)
#define Q_BY_NT_SMALL(CALL, NC_)       \
    do {                               \
        if (nt == 2) { CALL(NC_, 2); } \
        else if (nt == 4) { CALL(NC_, 4); } \
        else { CALL(NC_, 6); }         \
    } while (0)
#define Q_BY_NT_ALL(CALL, NC_)         \
    do {                               \
        if (nt == 2) { CALL(NC_, 2); } \
        else if (nt == 4) { CALL(NC_, 4); } \
        else if (nt == 6) { CALL(NC_, 6); } \
        else if (nt == 8) { CALL(NC_, 8); } \
        else if (nt == 10) { CALL(NC_, 10); } \
        else { CALL(NC_, 12); }        \
    } while (0)
#define Q_BY_NQ_SMALL(CALL, NC_)       \
    do {                               \
        if (nq == 1) { CALL(NC_, 1); } \
        else if (nq == 2) { CALL(NC_, 2); } \
        else { CALL(NC_, 3); }         \
    } while (0)
#define Q_BY_NQ_ALL(CALL, NC_)         \
    do {                               \
        if (nq == 1) { CALL(NC_, 1); } \
        else if (nq == 2) { CALL(NC_, 2); } \
        else if (nq == 3) { CALL(NC_, 3); } \
        else if (nq == 4) { CALL(NC_, 4); } \
        else if (nq == 5) { CALL(NC_, 5); } \
        else { CALL(NC_, 6); }         \
    } while (0)
// more than three quantiles per call for up to 128 columns only (four columns per lane run out of registers: Q_MAX_WIDE)
#define Q_BY_NT_1 Q_BY_NT_ALL
#define Q_BY_NT_2 Q_BY_NT_ALL
#define Q_BY_NT_4 Q_BY_NT_SMALL
#define Q_BY_NQ_1 Q_BY_NQ_ALL
#define Q_BY_NQ_2 Q_BY_NQ_ALL
#define Q_BY_NQ_4 Q_BY_NQ_SMALL

    // the four passes over rows 0, step, 2 step, ... (`rows` of them, leading dimension ldd): x_(rank) of every target ends in tg[].answer
    // (edges: stop after the second histogram and take the sub-bins' first / last keys as the answers -- pilot_edges_kernel)
    auto four_passes = [&](size_t rows, size_t ldd, const Wanted& ww, bool edges = false) -> int {
        const dim3 grid = grid_for(rows);
        Q_TRY(hipMemsetAsync(hist1, 0, o_tot + 256 - o_h1, st));            // hist1, hist2, total
        hipLaunchKernelGGL(init_columns_kernel, dim3(cblocks), dim3(64), 0, st, col, tg, m, nq, ww);
#define Q_MINMAX(NC_) hipLaunchKernelGGL(minmax_kernel<NC_>, grid, blk, 0, st, (long long)rows, m, ldd, cs, data, col)
        Q_BY_NC(Q_MINMAX);
        hipLaunchKernelGGL(scale_columns_kernel, dim3(cblocks), dim3(64), 0, st, col, m, bins1);
#define Q_HIST1(NC_)                                                                                       \
    Q_LDS(hist1_kernel<NC_>);                                                                              \
    hipLaunchKernelGGL(hist1_kernel<NC_>, grid, blk, lds1, st, (long long)rows, m, ldd, cs, data, col, bins1, hist1)
        Q_BY_NC(Q_HIST1);
        hipLaunchKernelGGL(decide1_kernel, dim3((unsigned)(m * nt)), dim3(64), 0, st, m, nt, col, hist1, bins1, tg, (long long)rows, inconsistent);
#define Q_HIST2_(NC_, NT_)                                                                                                  \
    Q_LDS((hist2_kernel<NC_, NT_>));                                                                                        \
    hipLaunchKernelGGL((hist2_kernel<NC_, NT_>), grid, blk, lds2, st, (long long)rows, m, ldd, cs, data, col, tg, bins1, bins2, hist2)
#define Q_HIST2(NC_) Q_BY_NT_##NC_(Q_HIST2_, NC_)
        Q_BY_NC(Q_HIST2);
        hipLaunchKernelGGL(decide2_kernel, dim3((unsigned)(m * nt)), dim3(64), 0, st, nt, hist2, bins2, tg);
        if (edges) {
            hipLaunchKernelGGL(pilot_edges_kernel, dim3((unsigned)((m * nt + 63) / 64)), dim3(64), 0, st, m * nt, nt, bins2, col, tg);
            Q_TRY(hipGetLastError());
            return PEM_OK;
        }
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64 * MAX_NC), 0, st, m, nt, tg, total);
        Q_TRY(hipGetLastError());
        u64 h_total = 0;
        Q_TRY(hipMemcpyAsync(&h_total, total, sizeof(u64), hipMemcpyDeviceToHost, st));
        Q_TRY(hipStreamSynchronize(st));
        Q_TRY(grow_candidates(h_total));
        u64* cand = cand_buf;
#define Q_COMPACT_(NC_, NT_) \
    hipLaunchKernelGGL((compact_kernel<NC_, NT_>), grid, blk, 0, st, (long long)rows, m, ldd, cs, data, col, tg, bins1, bins2, cand)
#define Q_COMPACT(NC_) Q_BY_NT_##NC_(Q_COMPACT_, NC_)
        Q_BY_NC(Q_COMPACT);
        hipLaunchKernelGGL(select_kernel, dim3((unsigned)(m * nt)), blk, 0, st, nt, tg, cand, incomplete);
        Q_TRY(hipGetLastError());
        return PEM_OK;
    };

    int path = 0;
    bool answered = false;
    if (plan && !use_pilot && plan->before_full)                           // (too few rows for a subsample: everything is awaited)
        if (int rc = plan->before_full(plan->ctx, st)) return cleanup(rc);
    if (use_pilot) {
        // 1. the subsample's order statistics around every wanted quantile: 7 standard deviations of the subsample's rank, + 8
        const size_t rows_p = (n + (size_t)pilot - 1) / (size_t)pilot;
        Wanted pw{};
        PilotEnds ends{};
        const double n1 = n > 1 ? (double)(n - 1) : 1.0, np1 = (double)(rows_p - 1);
        for (int q = 0; q < nq; ++q) {
            const double p_lo = (double)rank_prev[q] / n1, p_hi = (double)rank_next[q] / n1;
            // (the fused form: 5 -- every value inside a bracket costs a record, in the counting launch and in both passes over the
            // records: 3.7-3.85 ms per 1e7-sample campaign with 6, 3.6 with 5, 3.5 with 4.5 on one box.  A rank outside its bracket
            // -- 2.9e-7 per bracket end, 910 ends: one campaign in four thousand -- is noticed by the counts and costs that call the
            // stored-profile route; PEM_FUSED_SIGMA moves it)
            double sig = fused ? 5.0 : 7.0;
            if (fused)
                if (const char* e = getenv("PEM_FUSED_SIGMA")) sig = atof(e) >= 3.0 ? atof(e) : 3.0;
            const double d_lo = sig * sqrt((double)rows_p * p_lo * (1.0 - p_lo)) + 8.0, d_hi = sig * sqrt((double)rows_p * p_hi * (1.0 - p_hi)) + 8.0;
            const double r_lo = floor(p_lo * np1 - d_lo) - 1.0, r_hi = ceil(p_hi * np1 + d_hi) + 1.0;
            ends.open_lo[q] = r_lo < 0.0;
            ends.open_hi[q] = r_hi > np1;
            pw.prev[q] = ends.open_lo[q] ? 0 : (u64)r_lo;
            pw.next[q] = ends.open_hi[q] ? (u64)(rows_p - 1) : (u64)r_hi;
        }
        if (fused) {                                                        // the producer writes the pilot rows (contiguous) first
            if (int rc = fused->pilot(rows_p, const_cast<double*>(data), st)) return cleanup(rc);
            const bool edges = !(getenv("PEM_FUSED_PILOT_EXACT") && atoi(getenv("PEM_FUSED_PILOT_EXACT")));
            if (int rc = four_passes(rows_p, ld, pw, edges)) return rc;
        } else if (plan) {                                                  // the FIRST rows_p rows are the subsample; the rest is awaited
            if (int rc = four_passes(rows_p, ld, pw)) return rc;
            if (plan->before_full)
                if (int rc = plan->before_full(plan->ctx, st)) return cleanup(rc);
        } else if (int rc = four_passes(rows_p, ld * (size_t)pilot, pw)) return rc;
        // 2. pass A over everything: below / inside counts; the ranks inside their brackets become sub-bins
        const dim3 grid = grid_for(n);
        Q_TRY(hipMemsetAsync(total, 0, 256, st));                           // total, outside, any_single
        Q_TRY(hipMemsetAsync(below, 0, o_end - o_bl, st));                  // below, histA
        hipLaunchKernelGGL(brackets_kernel, dim3((unsigned)((m * nq + 63) / 64)), dim3(64), 0, st, m, nq, tg, ends, binsA, br, any_single);
        hipLaunchKernelGGL(init_columns_kernel, dim3(cblocks), dim3(64), 0, st, col, tg, m, nq, w);
#define Q_BRHIST_(NC_, NQ_)                                                                                                  \
    Q_LDS((bracket_hist_kernel<NC_, NQ_>));                                                                                  \
    hipLaunchKernelGGL((bracket_hist_kernel<NC_, NQ_>), grid, blk, ldsA, st, (long long)n, m, ld, cs, data, br, any_single, binsA, col, below, histA)
#define Q_BRHIST(NC_) Q_BY_NQ_##NC_(Q_BRHIST_, NC_)
        unsigned rcap = 0;
        bool pm_launched = false;                                           // fused form: the counting launch carried the premask
        if (!fused) {
            Q_BY_NC(Q_BRHIST);
        } else {
            // brackets the producer can count against?  (one small read: nothing else has to wait for the host here)
            hipLaunchKernelGGL(bracket_check_kernel, dim3((unsigned)((m * nq + 63) / 64)), dim3(64), 0, st, m, nq, br, unfit);
            const bool want_pm = fused->pm_q25 >= 0 && fused->pm_q25 < nq && fused->pm_q75 >= 0 && fused->pm_q75 < nq && nq <= 5 &&
                                 fused->pm_certain && fused->pm_uncertain;
            fused->pm_done = 0;
            if (want_pm)
                hipLaunchKernelGGL(premask_bounds_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, m, nq, br, fused->pm_q25, fused->pm_q75,
                                   fused->pm_factor, pm_thr, pm_bad);
            // (No host round trip here -- it cost every campaign 40 us of an idle GPU: whether the brackets are fit for the producer and
            // the premask's bounds usable is read back WITH the counts, after the counting launch.  Unfit brackets -- heavy ties,
            // overlapping percentiles -- are counted against all the same, harmlessly, and the run is then declined: the producer's
            // launch has written every output but the percentiles by then.)
            const bool use_pm = want_pm;
            pm_launched = want_pm;
            // record room per wave: the brackets hold `expected` values of the array (their ranks in the subsample say so) -- half as
            // much again, and a few thousand for the waves that get more than their share
            double expected = 0.0;
            for (int q = 0; q < nq; ++q) expected += (double)(pw.next[q] - pw.prev[q] + 1) * (double)pilot * (double)m;
            rcap = (unsigned)(1.5 * expected / (double)fused_waves) + 4096u;
            if (const char* e = getenv("PEM_QUANTILE_RECORD_CAP")) rcap = (unsigned)atoll(e);      // (tests: force the overflow path)
            const size_t need = (size_t)fused_waves * rcap;
            if (rec_cap < need) {
                if (rec_buf) (void)hipFree(rec_buf);
                rec_buf = nullptr;
                rec_cap = 0;
                Q_TRY(hipMalloc(&rec_buf, need * sizeof(pem::Record)));
                rec_cap = need;
            }
            pem::CountIO cio;
            cio.br = br;
            cio.nq = nq;
            cio.below = reinterpret_cast<unsigned long long*>(below);
            cio.rec = rec_buf;
            cio.rec_count = rec_count;
            cio.cap = rcap;
            cio.flags = prod_flags;
            if (use_pm) {
                cio.premask = pm_thr;
                cio.row_certain = fused->pm_certain;
                cio.row_uncertain = fused->pm_uncertain;
            }
            if (int rc = fused->count(cio, st)) return cleanup(rc);
            int cus = 256;
            Q_TRY(pem::device_cus(&cus));
            const int cells = m * nq * binsA;
            {
                const size_t need = (size_t)cus * cells + (size_t)cus * m * nt;
                if (part_cap < need) {
                    if (part_buf) (void)hipFree(part_buf);
                    part_buf = nullptr;
                    part_cap = 0;
                    Q_TRY(hipMalloc(&part_buf, need * sizeof(unsigned)));
                    part_cap = need;
                }
            }
#define Q_RECHIST(NQ_)                                                                                                       \
    do {                                                                                                                     \
        static pem::LdsAttrOnce attr;                                                                                        \
        Q_TRY(attr.ensure(reinterpret_cast<const void*>(record_hist_kernel<NQ_>), 148 * 1024)); /* (+ 12 KB of static tables) */ \
        hipLaunchKernelGGL(record_hist_kernel<NQ_>, dim3((unsigned)cus), dim3(RBLOCK), (size_t)m * nq * binsA * 4, st, rec_buf, rec_count, rcap, fused_waves, \
                           br, m * nq, binsA, part_buf);                                                                     \
    } while (0)
            switch (nq) {
                case 1: Q_RECHIST(1); break;
                case 2: Q_RECHIST(2); break;
                case 3: Q_RECHIST(3); break;
                case 4: Q_RECHIST(4); break;
                case 5: Q_RECHIST(5); break;
                default: Q_RECHIST(6); break;
            }
#undef Q_RECHIST
            u64* rec_sums = total + 16;                                         // (zeroed with `total` above)
            hipLaunchKernelGGL(record_reduce_kernel, dim3((unsigned)((cells + 63) / 64)), dim3(256), 0, st, part_buf, cus, cells, histA, rec_count, rcap,
                               fused_waves, rec_sums);
            hipLaunchKernelGGL(record_total_check_kernel, dim3(1), dim3(64), 0, st, rec_sums, inconsistent, prod_flags);
        }
        hipLaunchKernelGGL(decide_bracket_kernel, dim3((unsigned)(m * nt)), dim3(64), 0, st, nt, col, br, below, histA, binsA, tg, outside);
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64 * MAX_NC), 0, st, m, nt, tg, total);
        Q_TRY(hipGetLastError());
        u64 h_tot[2] = {0, 0};
        int h_unfit4[4] = {0, 0, 0, 0};                                    // fused form: unfit | producer flags (2) | premask bounds unfit
        Q_TRY(hipMemcpyAsync(h_tot, total, 2 * sizeof(u64), hipMemcpyDeviceToHost, st));
        if (fused) Q_TRY(hipMemcpyAsync(h_unfit4, unfit, sizeof h_unfit4, hipMemcpyDeviceToHost, st));
        Q_TRY(hipStreamSynchronize(st));
        if (fused) fused->pm_done = (pm_launched && h_unfit4[3] == 0) ? 1 : 0;
        if (fused && (h_unfit4[0] || (int)(h_tot[1] & 0xffffffffull) != 0 || h_unfit4[1] || h_unfit4[2])) {
            *fused_ok = 0;                     // unfit brackets, a rank outside its bracket, record overflow, a non-finite sample
            return cleanup(PEM_OK);
        }
        if ((int)(h_tot[1] & 0xffffffffull) == 0) {
            // 3. pass B: the values of those sub-bins, then the lists as in the four-pass form
            Q_TRY(grow_candidates(h_tot[0]));
            u64* cand = cand_buf;
#define Q_BRCOMPACT_(NC_, NQ_) \
    hipLaunchKernelGGL((compact_bracket_kernel<NC_, NQ_>), grid, blk, 0, st, (long long)n, m, ld, cs, data, br, tg, cand)
#define Q_BRCOMPACT(NC_) Q_BY_NQ_##NC_(Q_BRCOMPACT_, NC_)
            if (!fused) {
                Q_BY_NC(Q_BRCOMPACT);
            } else {
                int cus = 256;
                Q_TRY(pem::device_cus(&cus));
                unsigned* woff = part_buf + (size_t)cus * m * nq * binsA;
                hipLaunchKernelGGL(record_offsets_kernel, dim3((unsigned)(m * nt)), dim3(64), 0, st, nt, binsA, tg, part_buf, cus, m * nq * binsA, woff);
#define Q_RECCOMPACT(NQ_) \
    hipLaunchKernelGGL(record_compact_kernel<NQ_>, dim3((unsigned)cus), dim3(RBLOCK), 0, st, rec_buf, rec_count, rcap, fused_waves, br, m * nq, tg, woff, cand)
                switch (nq) {
                    case 1: Q_RECCOMPACT(1); break;
                    case 2: Q_RECCOMPACT(2); break;
                    case 3: Q_RECCOMPACT(3); break;
                    case 4: Q_RECCOMPACT(4); break;
                    case 5: Q_RECCOMPACT(5); break;
                    default: Q_RECCOMPACT(6); break;
                }
#undef Q_RECCOMPACT
            }
            hipLaunchKernelGGL(select_kernel, dim3((unsigned)(m * nt)), blk, 0, st, nt, tg, cand, incomplete);
            Q_TRY(hipGetLastError());
            answered = true;
            path = 1;
        } else {
            path = 2;
        }
    }
    if (!answered)
        if (int rc = four_passes(n, ld, w)) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3((unsigned)((m * nq + 63) / 64)), dim3(64), 0, st, m, nq, col, tg, w, out);
    Q_TRY(hipGetLastError());
    u64 h_incomplete[4] = {0, 0, 0, 0};
    int h_inconsistent = 0;
    Q_TRY(hipMemcpyAsync(h_incomplete, incomplete, sizeof h_incomplete, hipMemcpyDeviceToHost, st));
    Q_TRY(hipMemcpyAsync(&h_inconsistent, inconsistent, sizeof(int), hipMemcpyDeviceToHost, st));
    Q_TRY(hipStreamSynchronize(st));
    if (h_inconsistent > 0)
        return pem::fail(PEM_ERR_HIP, "pem_quantiles: the histogram of column %d does not add up to its number of values (internal error; result discarded; path %d)",
                         h_inconsistent - 1, path);
    if (h_inconsistent < 0)
        return pem::fail(PEM_ERR_HIP, "pem_quantiles: the record histogram does not add up to the number of records (internal error; result discarded)");
    if ((h_incomplete[0] & 0xffffffffull) && h_incomplete[2] == h_incomplete[3])
        return pem::fail(PEM_ERR_HIP, "pem_quantiles: a candidate list was sorted with fewer keys than it holds (internal error; result discarded; path %d)", path);
    if (h_incomplete[0] & 0xffffffffull)
        return pem::fail(PEM_ERR_HIP,
                         "pem_quantiles: the list of column %llu, rank %llu holds %llu values where %llu were counted (internal error; result "
                         "discarded; path %d)",
                         h_incomplete[1] / nt, h_incomplete[1] % nt, h_incomplete[2], h_incomplete[3], path);
    if (slot == 0) g_last_path.store(path);
    if (fused_ok) *fused_ok = 1;
#undef Q_BRCOMPACT
#undef Q_BRCOMPACT_
#undef Q_BRHIST
#undef Q_BRHIST_
#undef Q_COMPACT
#undef Q_COMPACT_
#undef Q_HIST2
#undef Q_HIST2_
#undef Q_HIST1
#undef Q_MINMAX
#undef Q_BY_NQ_4
#undef Q_BY_NQ_2
#undef Q_BY_NQ_1
#undef Q_BY_NT_4
#undef Q_BY_NT_2
#undef Q_BY_NT_1
#undef Q_BY_NQ_ALL
#undef Q_BY_NQ_SMALL
#undef Q_BY_NT_ALL
#undef Q_BY_NT_SMALL
#undef Q_BY_NC
#undef Q_LDS
#undef Q_TRY
    return cleanup(PEM_OK);
}

extern "C" int pem_quantiles_strided_f64_dev(size_t n, int m, const double* data, size_t ld, size_t cs, int nq, const uint64_t* rank_prev,
                                             const uint64_t* rank_next, const double* gamma, double* out, pem_stream_t stream) {
    return quantiles_impl(n, m, data, ld, cs, nq, rank_prev, rank_next, gamma, out, stream, nullptr, nullptr);
}

int pem::quantiles_fused(size_t n, int m, int nq, const uint64_t* rank_prev, const uint64_t* rank_next, const double* gamma, double* pilot_rows,
                         pem::FusedProducer& prod, double* out, int* fused_ok, hipStream_t st) {
    return quantiles_impl(n, m, pilot_rows, (size_t)m, 1, nq, rank_prev, rank_next, gamma, out, static_cast<pem_stream_t>(st), &prod, fused_ok);
}

int pem::quantiles_side(size_t n, int m, const double* data, size_t ld, size_t cs, int nq, const uint64_t* rank_prev, const uint64_t* rank_next,
                        const double* gamma, double* out, hipStream_t st, const pem::SidePlan* plan) {
    return quantiles_impl(n, m, data, ld, cs, nq, rank_prev, rank_next, gamma, out, static_cast<pem_stream_t>(st), nullptr, nullptr, 1, plan);
}

extern "C" int pem_quantiles_f64_dev(size_t n, int m, const double* data, size_t ld, int nq, const uint64_t* rank_prev, const uint64_t* rank_next,
                                     const double* gamma, double* out, pem_stream_t stream) {
    if (ld < (size_t)(m > 0 ? m : 0)) return pem::fail(PEM_ERR_INVALID_ARG, "pem_quantiles: leading dimension smaller than m");
    return pem_quantiles_strided_f64_dev(n, m, data, ld, 1, nq, rank_prev, rank_next, gamma, out, stream);
}

// ---- multi-rank building blocks (one level each; hallthrusterpem_amd/percentiles.py) ---------------------------------------
namespace {

dim3 stream_grid(size_t n, int m) {
    const long long rpw = m <= 64 ? 64 / m : 1, groups = ((long long)n + rpw - 1) / rpw;
    long long blocks = (groups + (long long)QWAVES * UNROLL - 1) / ((long long)QWAVES * UNROLL);
    if (blocks > 256 * 2) blocks = 256 * 2;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks);
}

}  // namespace

extern "C" int pem_key_minmax_f64_dev(size_t n, int m, const double* data, size_t ld, uint64_t* kmin, uint64_t* kmax, int32_t* has_nan,
                                      pem_stream_t stream) {
    if (m < 1 || m > 64 * MAX_NC) return pem::fail(PEM_ERR_INVALID_ARG, "pem_key_minmax: 1 <= m <= %d columns", 64 * MAX_NC);
    if (ld < (size_t)m) return pem::fail(PEM_ERR_INVALID_ARG, "pem_key_minmax: leading dimension smaller than m");
    if (!kmin || !kmax || !has_nan || (n && !data)) return pem::fail(PEM_ERR_INVALID_ARG, "pem_key_minmax: NULL array");
    if (int rc = pem::check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Column* col = nullptr;
    HIP_TRY(hipMalloc(&col, sizeof(Column) * m));
    Target* no_targets = nullptr;
    hipLaunchKernelGGL(init_columns_kernel, dim3((m + 63) / 64), dim3(64), 0, st, col, no_targets, m, 0, Wanted{});
    if (n) {
        const int nc = m <= 64 ? 1 : (m <= 128 ? 2 : 4);
        const dim3 grid = stream_grid(n, m), blk(QBLOCK);
        if (nc == 1) hipLaunchKernelGGL(minmax_kernel<1>, grid, blk, 0, st, (long long)n, m, ld, (size_t)1, data, col);
        else if (nc == 2) hipLaunchKernelGGL(minmax_kernel<2>, grid, blk, 0, st, (long long)n, m, ld, (size_t)1, data, col);
        else hipLaunchKernelGGL(minmax_kernel<4>, grid, blk, 0, st, (long long)n, m, ld, (size_t)1, data, col);
    }
    hipLaunchKernelGGL(export_minmax_kernel, dim3((m + 63) / 64), dim3(64), 0, st, col, m, (u64*)kmin, (u64*)kmax, (int*)has_nan);
    const hipError_t e = hipGetLastError();
    (void)hipStreamSynchronize(st);
    (void)hipFree(col);
    if (e != hipSuccess) return pem::fail(PEM_ERR_HIP, "pem_key_minmax: %s", hipGetErrorString(e));
    return PEM_OK;
}

extern "C" int pem_range_hist_f64_dev(size_t n, int m, const double* data, size_t ld, int nr, const uint64_t* klo, const uint64_t* khi,
                                      int bins, uint32_t* hist, pem_stream_t stream) {
    if (m < 1 || m > 64 * MAX_NC) return pem::fail(PEM_ERR_INVALID_ARG, "pem_range_hist: 1 <= m <= %d columns", 64 * MAX_NC);
    if (nr != 1 && nr != 2 && nr != 4 && nr != 6) return pem::fail(PEM_ERR_INVALID_ARG, "pem_range_hist: 1, 2, 4 or 6 ranges per column");
    if (bins < 1 || (long long)m * nr * bins > LDS_WORDS)
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_range_hist: m * nr * bins must not exceed %d", LDS_WORDS);
    if (ld < (size_t)m) return pem::fail(PEM_ERR_INVALID_ARG, "pem_range_hist: leading dimension smaller than m");
    if (!klo || !khi || !hist || (n && !data)) return pem::fail(PEM_ERR_INVALID_ARG, "pem_range_hist: NULL array");
    if (int rc = pem::check_device()) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(hist, 0, sizeof(uint32_t) * (size_t)m * nr * bins, st));
    if (n == 0) return PEM_OK;
    const int nc = m <= 64 ? 1 : (m <= 128 ? 2 : 4);
    const dim3 grid = stream_grid(n, m), blk(QBLOCK);
    const size_t lds = (size_t)m * nr * bins * 4;
#define R_LAUNCH(NC_, NR_)                                                                                                       \
    do {                                                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(range_hist_kernel<NC_, NR_>),                                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                      \
        hipLaunchKernelGGL((range_hist_kernel<NC_, NR_>), grid, blk, lds, st, (long long)n, m, ld, data, (const u64*)klo,       \
                           (const u64*)khi, bins, (unsigned*)hist);                                                              \
    } while (0)
#define R_BY_NR(NC_)                       \
    do {                                   \
        if (nr == 1) R_LAUNCH(NC_, 1);     \
        else if (nr == 2) R_LAUNCH(NC_, 2); \
        else if (nr == 4) R_LAUNCH(NC_, 4); \
        else R_LAUNCH(NC_, 6);             \
    } while (0)
    if (nc == 1) R_BY_NR(1);
    else if (nc == 2) R_BY_NR(2);
    else R_BY_NR(4);
#undef R_BY_NR
#undef R_LAUNCH
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

extern "C" int pem_range_narrow_dev(int n_ranges, int bins, const uint32_t* hist, uint64_t* klo, uint64_t* khi, int64_t* resid,
                                    pem_stream_t stream) {
    if (n_ranges < 1 || bins < 1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_range_narrow: need ranges and bins");
    if (!hist || !klo || !khi || !resid) return pem::fail(PEM_ERR_INVALID_ARG, "pem_range_narrow: NULL array");
    if (int rc = pem::check_device()) return rc;
    hipLaunchKernelGGL(range_narrow_kernel, dim3((unsigned)n_ranges), dim3(64), 0, static_cast<hipStream_t>(stream), bins,
                       (const unsigned*)hist, (u64*)klo, (u64*)khi, (long long*)resid);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

// ---- the sharded selection on the single-GPU selection's passes (round 3) -------------------------------------------------------
// The level loop above narrows a key range by 64 bins per streaming pass: eleven passes for 1e7 x 91 values.  The single-GPU
// selection needs four, and its two histograms ADD across ranks just as well: every rank runs the same passes over its own
// rows, the per-column min / max and the two histograms are all-reduced between them (91 x 256 and 91 x 6 x 64 counters:
// nothing on xGMI), every rank takes the same decisions, and what is left per wanted rank -- 1 / (bins1 bins2) of a column --
// is copied out into fixed-length padded lists that are all-gathered and selected from by one workgroup per list.
// hallthrusterpem_amd/percentiles.py drives the stages and restates them in numpy for the gloo rehearsal on the CPU.
// State lives in plain caller-owned device arrays (so that the collectives can run on them):
//   kmin / kmax [m] u64, has_nan [m] i32; hist1 [m][bins1] u32; hist2 [m][nt][bins2] u32;
//   resid [m nt] u64 (in: the wanted 0-based ranks, rewritten to the rank inside the chosen bin, then sub-bin);
//   bin1 / bin2 [m nt] i32; done [m nt] i32 + answer [m nt] u64 (constant columns are decided at once);
//   cand [m nt][L] u64 padded with ~0, cursor [m nt] u32 (values a list was offered: > L means it overflowed).
namespace {

__device__ __forceinline__ Scale column_scale(u64 kmin, u64 kmax, int bins1) {      // scale_columns_kernel, per lane
    Scale sc;
    const u64 span = kmax >= kmin ? kmax - kmin : 0;
    int shift = 0;
    while ((span >> shift) >> 31) ++shift;
    const u64 mult = (((u64)bins1) << 32) / ((span >> shift) + 1);
    sc.lo = kmin;
    sc.shift = shift;
    sc.mult = mult > 0xffffffffull ? 0xffffffffu : (unsigned)mult;
    return sc;
}


__global__ void qsel_init_kernel(int m, u64* kmin, u64* kmax, int* has_nan) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < m) {
        kmin[c] = ~0ull;
        kmax[c] = 0ull;
        has_nan[c] = 0;
    }
}

template <int NC>
__global__ __launch_bounds__(QBLOCK) void qsel_minmax_kernel(long long n, int m, size_t ld, const double* __restrict__ data, u64* __restrict__ kmin,
                                                              u64* __restrict__ kmax, int* __restrict__ has_nan) {
    const Lanes L(m, threadIdx.x & 63);
    u64 lo[NC], hi[NC];
    int nan[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        lo[j] = ~0ull;
        hi[j] = 0ull;
        nan[j] = 0;
    }
    stream_values<NC>(n, m, ld, data, L, [&](int j, int, double x) {
        if (x != x) nan[j] = 1;
        else {
            const u64 k = key_of(x);
            lo[j] = k < lo[j] ? k : lo[j];
            hi[j] = k > hi[j] ? k : hi[j];
        }
    });
    __shared__ u64 s_lo[64 * MAX_NC], s_hi[64 * MAX_NC];
    __shared__ int s_nan[64 * MAX_NC];
    for (int c = threadIdx.x; c < m; c += QBLOCK) {
        s_lo[c] = ~0ull;
        s_hi[c] = 0ull;
        s_nan[c] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        if (L.active && c < m) {
            if (lo[j] <= hi[j]) {
                atomicMin(&s_lo[c], lo[j]);
                atomicMax(&s_hi[c], hi[j]);
            }
            if (nan[j]) atomicOr(&s_nan[c], 1);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < m; c += QBLOCK) {
        if (s_lo[c] <= s_hi[c]) {
            atomicMin(&kmin[c], s_lo[c]);
            atomicMax(&kmax[c], s_hi[c]);
        }
        if (s_nan[c]) atomicOr(&has_nan[c], 1);
    }
}

template <int NC>
__global__ __launch_bounds__(QBLOCK) void qsel_hist1_kernel(long long n, int m, size_t ld, const double* __restrict__ data, const u64* __restrict__ kmin,
                                                             const u64* __restrict__ kmax, int bins1, unsigned* __restrict__ hist1) {
    extern __shared__ unsigned lds_hist[];
    for (int i = threadIdx.x; i < m * bins1; i += QBLOCK) lds_hist[i] = 0;
    __syncthreads();
    const Lanes L(m, threadIdx.x & 63);
    Scale sc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        const bool on = L.active && c < m;
        sc[j] = on ? column_scale(kmin[c], kmax[c], bins1) : column_scale(1, 0, bins1);
    }
    stream_values<NC>(n, m, ld, data, L, [&](int j, int c, double x) {
        if (x == x) {
            unsigned frac;
            atomicAdd(&lds_hist[c * bins1 + bin_of(key_of(x), sc[j], frac)], 1u);
        }
    });
    __syncthreads();
    for (int i = threadIdx.x; i < m * bins1; i += QBLOCK) {
        const unsigned v = lds_hist[i];
        if (v) atomicAdd(&hist1[i], v);
    }
}

// one wave per (column, target): the bin of the (all-reduced) histogram that holds the wanted rank
__global__ __launch_bounds__(64) void qsel_decide1_kernel(int nt, const u64* __restrict__ kmin, const u64* __restrict__ kmax,
                                                           const unsigned* __restrict__ hist1, int bins1, u64* __restrict__ resid,
                                                           int* __restrict__ bin1, int* __restrict__ done, u64* __restrict__ answer) {
    const int i = blockIdx.x, lane = threadIdx.x, c = i / nt;
    if (kmin[c] >= kmax[c]) {                 // a constant column, or one without a value (its result is NaN anyway)
        if (lane == 0) {
            done[i] = 1;
            answer[i] = kmin[c];
            bin1[i] = -1;
        }
        return;
    }
    const Found f = find_bin(hist1 + (size_t)c * bins1, bins1, resid[i], lane);
    if (lane == 0) {
        done[i] = 0;
        bin1[i] = f.bin;
        resid[i] -= f.before;
    }
}

template <int NC, int NT>
__global__ __launch_bounds__(QBLOCK) void qsel_hist2_kernel(long long n, int m, size_t ld, const double* __restrict__ data, const u64* __restrict__ kmin,
                                                             const u64* __restrict__ kmax, const int* __restrict__ bin1, int bins1, int bins2,
                                                             unsigned* __restrict__ hist2) {
    extern __shared__ unsigned lds_hist[];
    for (int i = threadIdx.x; i < m * NT * bins2; i += QBLOCK) lds_hist[i] = 0;
    __syncthreads();
    const Lanes L(m, threadIdx.x & 63);
    Scale sc[NC];
    int tb[NC][NT];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        const bool on = L.active && c < m;
        sc[j] = on ? column_scale(kmin[c], kmax[c], bins1) : column_scale(1, 0, bins1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // targets of a column often share a bin: each DISTINCT bin is counted once, under the first target that has it
            int b = on ? bin1[c * NT + t] : -1;
#pragma unroll
            for (int u = 0; u < t; ++u)
                if (on && bin1[c * NT + u] == b) b = -1;
            tb[j][t] = b;
        }
    }
    // (no high-word test in front of the binning here, as the copy pass has: a first-level bin around the median holds 5-10 % of a
    // column, so nearly every wave instruction has a lane inside one and the test only adds to the work: 1972 -> 2255 us for 1e7 x 91)
    stream_values<NC>(n, m, ld, data, L, [&](int j, int c, double x) {
        if (x == x) {
            unsigned frac;
            const int b = bin_of(key_of(x), sc[j], frac);
            int hit = -1;                                   // at most one target holds a given bin (deduplicated above): one branch, not NT
#pragma unroll
            for (int t = 0; t < NT; ++t) hit = tb[j][t] == b ? t : hit;
            if (hit >= 0) atomicAdd(&lds_hist[(c * NT + hit) * bins2 + subbin_of(frac, bins2)], 1u);
        }
    });
    __syncthreads();
    for (int i = threadIdx.x; i < m * NT * bins2; i += QBLOCK) {
        const unsigned v = lds_hist[i];
        if (v) atomicAdd(&hist2[i], v);
    }
}

// one wave per (column, target): the sub-bin that holds the rank, from the all-reduced counts; how many of THIS rank's values
// sit in it, from the local ones (the histogram of a bin is kept under the column's first target that has it)
__global__ __launch_bounds__(64) void qsel_decide2_kernel(int nt, const unsigned* __restrict__ hist2, const unsigned* __restrict__ hist2_local,
                                                           int bins2, const int* __restrict__ bin1, const int* __restrict__ done,
                                                           u64* __restrict__ resid, int* __restrict__ bin2, u64* __restrict__ count,
                                                           unsigned* __restrict__ count_local) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int c = i / nt, t = i - c * nt;
    if (done[i]) {
        if (lane == 0) {
            bin2[i] = -1;
            count[i] = 0;
            count_local[i] = 0;
        }
        return;
    }
    int first = t;
    for (int u = t - 1; u >= 0; --u)
        if (!done[c * nt + u] && bin1[c * nt + u] == bin1[i]) first = u;
    const Found f = find_bin(hist2 + (size_t)(c * nt + first) * bins2, bins2, resid[i], lane);
    if (lane == 0) {
        bin2[i] = f.bin;
        resid[i] -= f.before;
        count[i] = f.count;
        count_local[i] = hist2_local[(size_t)(c * nt + first) * bins2 + f.bin];
    }
}

// this rank's values of every wanted (bin, sub-bin), one padded list per (column, target) -- only the first target of a column
// with a given pair collects (the others read its list)
template <int NC, int NT>
__global__ __launch_bounds__(QBLOCK) void qsel_compact_kernel(long long n, int m, size_t ld, const double* __restrict__ data, const u64* __restrict__ kmin,
                                                               const u64* __restrict__ kmax, const int* __restrict__ bin1, const int* __restrict__ bin2,
                                                               const int* __restrict__ done, int bins1, int bins2, unsigned list_len,
                                                               u64* __restrict__ cand, unsigned* __restrict__ cursor) {
    // the key range of every collected (bin, sub-bin), found once per workgroup -- thread i bisects for list i -- and shared
    // through LDS: done by every lane for its own columns, the bisections were a quarter of a 1.25e6-row shard's pass
    __shared__ unsigned s_rloh[64 * MAX_NC * 2 * PEM_QUANTILE_MAX_Q], s_rwords[64 * MAX_NC * 2 * PEM_QUANTILE_MAX_Q];
    for (int i = threadIdx.x; i < m * NT; i += QBLOCK) {
        const int c = i / NT, t = i - c * NT;
        bool own = !done[i];
        for (int u = 0; u < t; ++u)
            if (own && !done[c * NT + u] && bin1[c * NT + u] == bin1[i] && bin2[c * NT + u] == bin2[i]) own = false;
        u64 klo = ~0ull, khi = 0ull;
        if (own) keys_of_bin(column_scale(kmin[c], kmax[c], bins1), kmax[c], bins2, (long long)bin1[i] * bins2 + bin2[i], klo, khi);
        if (klo > khi) klo = khi = ~0ull;                       // nothing to collect: a range no value's key lies in
        s_rloh[i] = (unsigned)(klo >> 32);
        s_rwords[i] = (unsigned)(khi >> 32) - (unsigned)(klo >> 32);
    }
    __syncthreads();
    const Lanes L(m, threadIdx.x & 63);
    Scale sc[NC];
    int tb1[NC][NT], tb2[NC][NT];
    unsigned rloh[NC][NT], rwords[NC][NT];                  // high words of the collected sub-bins' key ranges
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = L.col0 + 64 * j;
        const bool on = L.active && c < m;
        sc[j] = on ? column_scale(kmin[c], kmax[c], bins1) : column_scale(1, 0, bins1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bool own = on && !done[c * NT + t];
#pragma unroll
            for (int u = 0; u < t; ++u)
                if (own && !done[c * NT + u] && bin1[c * NT + u] == bin1[c * NT + t] && bin2[c * NT + u] == bin2[c * NT + t]) own = false;
            tb1[j][t] = own ? bin1[c * NT + t] : -1;
            tb2[j][t] = own ? bin2[c * NT + t] : -1;
            rloh[j][t] = on ? s_rloh[c * NT + t] : 0xffffffffu;
            rwords[j][t] = on ? s_rwords[c * NT + t] : 0u;
        }
    }
    stream_values<NC>(n, m, ld, data, L, [&](int j, int c, double x) {
        const unsigned kh = key_high(x);
        bool near = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) near |= kh - rloh[j][t] <= rwords[j][t];
        if (near && x == x) {
            const u64 k = key_of(x);
            unsigned frac;
            const int b = bin_of(k, sc[j], frac);
            const int sb = subbin_of(frac, bins2);
            int hit = -1;                                   // list owners have distinct (bin, sub-bin) pairs: at most one matches
#pragma unroll
            for (int t = 0; t < NT; ++t) hit = (tb1[j][t] == b && tb2[j][t] == sb) ? t : hit;
            if (hit >= 0) {
                const unsigned at = atomicAdd(&cursor[c * NT + hit], 1u);
                if (at < list_len) cand[(size_t)(c * NT + hit) * list_len + at] = k;
            }
        }
    });
}

// one workgroup per (column, target): x_(resid) of the union of every rank's list of its (bin, sub-bin)
__global__ __launch_bounds__(QBLOCK) void qsel_select_kernel(int nt, int world, unsigned list_len, size_t rank_stride, const u64* __restrict__ gathered,
                                                              const int* __restrict__ bin1, const int* __restrict__ bin2, const int* __restrict__ done,
                                                              const u64* __restrict__ resid, u64* __restrict__ answer) {
    const int i = blockIdx.x;
    if (done[i]) return;
    const int c = i / nt, t = i - c * nt;
    int owner = t;
    for (int u = t - 1; u >= 0; --u)
        if (!done[c * nt + u] && bin1[c * nt + u] == bin1[i] && bin2[c * nt + u] == bin2[i]) owner = u;
    const u64* base = gathered + (size_t)(c * nt + owner) * list_len;
    const u64 r = select_from([=](u64 k) { return base[(size_t)(k / list_len) * rank_stride + (size_t)(k % list_len)]; },
                              (u64)world * list_len, resid[i], 0xfff0000000000000ull /* key of +inf: the padding lies above it */);
    if (threadIdx.x == 0) answer[i] = r;
}

int qsel_check(const char* who, size_t n, int m, size_t ld, const void* data) {
    if (m < 1 || m > 64 * MAX_NC) return pem::fail(PEM_ERR_INVALID_ARG, "%s: 1 <= m <= %d columns", who, 64 * MAX_NC);
    if (ld < (size_t)m) return pem::fail(PEM_ERR_INVALID_ARG, "%s: leading dimension smaller than m", who);
    if (n && !data) return pem::fail(PEM_ERR_INVALID_ARG, "%s: NULL data", who);
    return pem::check_device();
}

}  // namespace

extern "C" {

int pem_qsel_bins(int m, int nt, int* bins1, int* bins2) {
    if (m < 1 || m > 64 * MAX_NC || nt < 1 || nt > 2 * PEM_QUANTILE_MAX_Q || !bins1 || !bins2)
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_bins: 1 <= m <= %d, 1 <= nt <= %d", 64 * MAX_NC, 2 * PEM_QUANTILE_MAX_Q);
    *bins1 = pow2_at_most(LDS_WORDS / m, 4096);
    *bins2 = pow2_at_most(LDS_WORDS / (m * nt), 4096);
    return PEM_OK;
}

int pem_qsel_minmax_f64_dev(size_t n, int m, const double* data, size_t ld, uint64_t* kmin, uint64_t* kmax, int32_t* has_nan,
                            pem_stream_t stream) {
    if (int rc = qsel_check("pem_qsel_minmax", n, m, ld, data)) return rc;
    if (!kmin || !kmax || !has_nan) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_minmax: NULL array");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(qsel_init_kernel, dim3((m + 63) / 64), dim3(64), 0, st, m, (u64*)kmin, (u64*)kmax, (int*)has_nan);
    if (n) {
        const int nc = m <= 64 ? 1 : (m <= 128 ? 2 : 4);
        const dim3 grid = stream_grid(n, m), blk(QBLOCK);
        if (nc == 1) hipLaunchKernelGGL(qsel_minmax_kernel<1>, grid, blk, 0, st, (long long)n, m, ld, data, (u64*)kmin, (u64*)kmax, (int*)has_nan);
        else if (nc == 2) hipLaunchKernelGGL(qsel_minmax_kernel<2>, grid, blk, 0, st, (long long)n, m, ld, data, (u64*)kmin, (u64*)kmax, (int*)has_nan);
        else hipLaunchKernelGGL(qsel_minmax_kernel<4>, grid, blk, 0, st, (long long)n, m, ld, data, (u64*)kmin, (u64*)kmax, (int*)has_nan);
    }
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

int pem_qsel_hist1_f64_dev(size_t n, int m, const double* data, size_t ld, const uint64_t* kmin, const uint64_t* kmax, int bins1,
                           uint32_t* hist1, pem_stream_t stream) {
    if (int rc = qsel_check("pem_qsel_hist1", n, m, ld, data)) return rc;
    if (!kmin || !kmax || !hist1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_hist1: NULL array");
    if (bins1 < 1 || (long long)m * bins1 > LDS_WORDS) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_hist1: m * bins1 must not exceed %d", LDS_WORDS);
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(hist1, 0, sizeof(uint32_t) * (size_t)m * bins1, st));
    if (n == 0) return PEM_OK;
    const int nc = m <= 64 ? 1 : (m <= 128 ? 2 : 4);
    const dim3 grid = stream_grid(n, m), blk(QBLOCK);
    const size_t lds = (size_t)m * bins1 * 4;
#define H1(NC_)                                                                                                                  \
    do {                                                                                                                         \
        static pem::LdsAttrOnce attr;                                                                                            \
        HIP_TRY(attr.ensure(reinterpret_cast<const void*>(qsel_hist1_kernel<NC_>)));                                             \
        hipLaunchKernelGGL(qsel_hist1_kernel<NC_>, grid, blk, lds, st, (long long)n, m, ld, data, (const u64*)kmin, (const u64*)kmax, \
                           bins1, (unsigned*)hist1);                                                                             \
    } while (0)
    if (nc == 1) H1(1);
    else if (nc == 2) H1(2);
    else H1(4);
#undef H1
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

int pem_qsel_decide1_dev(int m, int nt, const uint64_t* kmin, const uint64_t* kmax, const uint32_t* hist1, int bins1, uint64_t* resid,
                         int32_t* bin1, int32_t* done, uint64_t* answer, pem_stream_t stream) {
    if (m < 1 || nt < 1 || bins1 < 1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_decide1: need columns, targets and bins");
    if (!kmin || !kmax || !hist1 || !resid || !bin1 || !done || !answer) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_decide1: NULL array");
    if (int rc = pem::check_device()) return rc;
    hipLaunchKernelGGL(qsel_decide1_kernel, dim3((unsigned)(m * nt)), dim3(64), 0, static_cast<hipStream_t>(stream), nt, (const u64*)kmin,
                       (const u64*)kmax, (const unsigned*)hist1, bins1, (u64*)resid, (int*)bin1, (int*)done, (u64*)answer);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

int pem_qsel_hist2_f64_dev(size_t n, int m, const double* data, size_t ld, const uint64_t* kmin, const uint64_t* kmax, int nt,
                           const int32_t* bin1, int bins1, int bins2, uint32_t* hist2, pem_stream_t stream) {
    if (int rc = qsel_check("pem_qsel_hist2", n, m, ld, data)) return rc;
    if (nt < 2 || nt > 2 * PEM_QUANTILE_MAX_Q || (nt & 1)) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_hist2: 2, 4, ... %d targets per column", 2 * PEM_QUANTILE_MAX_Q);
    if (!kmin || !kmax || !bin1 || !hist2) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_hist2: NULL array");
    if (bins1 < 1 || bins2 < 1 || (long long)m * nt * bins2 > LDS_WORDS)
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_hist2: m * nt * bins2 must not exceed %d", LDS_WORDS);
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(hist2, 0, sizeof(uint32_t) * (size_t)m * nt * bins2, st));
    if (n == 0) return PEM_OK;
    const int nc = m <= 64 ? 1 : (m <= 128 ? 2 : 4);
    const dim3 grid = stream_grid(n, m), blk(QBLOCK);
    const size_t lds = (size_t)m * nt * bins2 * 4;
#define H2(NC_, NT_)                                                                                                               \
    do {                                                                                                                           \
        static pem::LdsAttrOnce attr;                                                                                              \
        HIP_TRY(attr.ensure(reinterpret_cast<const void*>(qsel_hist2_kernel<NC_, NT_>)));                                          \
        hipLaunchKernelGGL((qsel_hist2_kernel<NC_, NT_>), grid, blk, lds, st, (long long)n, m, ld, data, (const u64*)kmin, (const u64*)kmax, \
                           (const int*)bin1, bins1, bins2, (unsigned*)hist2);                                                      \
    } while (0)
#define H2_NT(NC_)                       \
    do {                                 \
        if (nt == 2) H2(NC_, 2);         \
        else if (nt == 4) H2(NC_, 4);    \
        else if (nt == 6) H2(NC_, 6);    \
        else if (nt == 8) H2(NC_, 8);    \
        else if (nt == 10) H2(NC_, 10);  \
        else H2(NC_, 12);                \
    } while (0)
#define H2_NT_SMALL(NC_)                 \
    do {                                 \
        if (nt == 2) H2(NC_, 2);         \
        else if (nt == 4) H2(NC_, 4);    \
        else H2(NC_, 6);                 \
    } while (0)
    if (nc == 4 && nt > 2 * PEM_QUANTILE_MAX_Q_WIDE) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_hist2: at most %d targets for more than 128 columns", 2 * PEM_QUANTILE_MAX_Q_WIDE);
    if (nc == 1) H2_NT(1);
    else if (nc == 2) H2_NT(2);
    else H2_NT_SMALL(4);
#undef H2_NT_SMALL
#undef H2_NT
#undef H2
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

int pem_qsel_decide2_dev(int m, int nt, const uint32_t* hist2, const uint32_t* hist2_local, int bins2, const int32_t* bin1, const int32_t* done,
                         uint64_t* resid, int32_t* bin2, uint64_t* count, uint32_t* count_local, pem_stream_t stream) {
    if (m < 1 || nt < 1 || bins2 < 1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_decide2: need columns, targets and bins");
    if (!hist2 || !hist2_local || !bin1 || !done || !resid || !bin2 || !count || !count_local)
        return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_decide2: NULL array");
    if (int rc = pem::check_device()) return rc;
    hipLaunchKernelGGL(qsel_decide2_kernel, dim3((unsigned)(m * nt)), dim3(64), 0, static_cast<hipStream_t>(stream), nt, (const unsigned*)hist2,
                       (const unsigned*)hist2_local, bins2, (const int*)bin1, (const int*)done, (u64*)resid, (int*)bin2, (u64*)count,
                       (unsigned*)count_local);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

int pem_qsel_compact_f64_dev(size_t n, int m, const double* data, size_t ld, const uint64_t* kmin, const uint64_t* kmax, int nt,
                             const int32_t* bin1, const int32_t* bin2, const int32_t* done, int bins1, int bins2, uint32_t list_len,
                             uint64_t* cand, uint32_t* cursor, pem_stream_t stream) {
    if (int rc = qsel_check("pem_qsel_compact", n, m, ld, data)) return rc;
    if (nt < 2 || nt > 2 * PEM_QUANTILE_MAX_Q || (nt & 1)) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_compact: 2, 4, ... %d targets per column", 2 * PEM_QUANTILE_MAX_Q);
    if (!kmin || !kmax || !bin1 || !bin2 || !done || !cand || !cursor || list_len < 1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_compact: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(cand, 0xff, sizeof(uint64_t) * (size_t)m * nt * list_len, st));       // padding: ~0, above every key
    HIP_TRY(hipMemsetAsync(cursor, 0, sizeof(uint32_t) * (size_t)m * nt, st));
    if (n == 0) return PEM_OK;
    const int nc = m <= 64 ? 1 : (m <= 128 ? 2 : 4);
    const dim3 grid = stream_grid(n, m), blk(QBLOCK);
#define CP(NC_, NT_)                                                                                                                  \
    hipLaunchKernelGGL((qsel_compact_kernel<NC_, NT_>), grid, blk, 0, st, (long long)n, m, ld, data, (const u64*)kmin, (const u64*)kmax, \
                       (const int*)bin1, (const int*)bin2, (const int*)done, bins1, bins2, (unsigned)list_len, (u64*)cand, (unsigned*)cursor)
#define CP_NT(NC_)                       \
    do {                                 \
        if (nt == 2) CP(NC_, 2);         \
        else if (nt == 4) CP(NC_, 4);    \
        else if (nt == 6) CP(NC_, 6);    \
        else if (nt == 8) CP(NC_, 8);    \
        else if (nt == 10) CP(NC_, 10);  \
        else CP(NC_, 12);                \
    } while (0)
#define CP_NT_SMALL(NC_)                 \
    do {                                 \
        if (nt == 2) CP(NC_, 2);         \
        else if (nt == 4) CP(NC_, 4);    \
        else CP(NC_, 6);                 \
    } while (0)
    if (nc == 4 && nt > 2 * PEM_QUANTILE_MAX_Q_WIDE) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_compact: at most %d targets for more than 128 columns", 2 * PEM_QUANTILE_MAX_Q_WIDE);
    if (nc == 1) CP_NT(1);
    else if (nc == 2) CP_NT(2);
    else CP_NT_SMALL(4);
#undef CP_NT_SMALL
#undef CP_NT
#undef CP
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

int pem_qsel_select_dev(int m, int nt, int world, uint32_t list_len, const uint64_t* gathered, const int32_t* bin1, const int32_t* bin2,
                        const int32_t* done, const uint64_t* resid, uint64_t* answer, pem_stream_t stream) {
    if (m < 1 || nt < 1 || world < 1 || list_len < 1) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_select: need columns, targets, ranks and a list length");
    if (!gathered || !bin1 || !bin2 || !done || !resid || !answer) return pem::fail(PEM_ERR_INVALID_ARG, "pem_qsel_select: NULL array");
    if (int rc = pem::check_device()) return rc;
    hipLaunchKernelGGL(qsel_select_kernel, dim3((unsigned)(m * nt)), dim3(QBLOCK), 0, static_cast<hipStream_t>(stream), nt, world,
                       (unsigned)list_len, (size_t)m * nt * list_len, (const u64*)gathered, (const int*)bin1, (const int*)bin2, (const int*)done,
                       (const u64*)resid, (u64*)answer);
    HIP_TRY(hipGetLastError());
    return PEM_OK;
}

}  // extern "C"
