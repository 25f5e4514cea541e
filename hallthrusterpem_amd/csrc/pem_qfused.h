// pem_qfused.h -- what the evaluation kernel (csrc/pem_kernels.hip) and the percentile selection (csrc/pem_quantile.hip) share
// for the FUSED campaign statistics (round 4): the percentiles of scripts/gen_data.py:125-174 (p25 / p75 of the IQR masks) and
// scripts/pem_v0/monte_carlo.py:363-658 (5 / 50 / 95 % bands) of the 91-point profile, counted where the profile is produced.
//
// The pilot form of the selection (pem_quantile.hip) reads the profile twice after it has been written: one pass counts, per
// (angle, quantile), the values below a bracket and a histogram inside it, one pass copies the chosen sub-bins out.  Here the
// evaluation kernel does the first pass's work on the round tile it has staged in LDS anyway -- lane = angle, one subtraction of
// high words per (value, bracket) -- and writes the few per cent of values that lie INSIDE a bracket out as records; both passes
// then run over the records (4 % of the data) instead of the profile.  Internal to libpem_hip.so: not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pem {

struct Bracket {         // per (column, quantile): the keys lo .. hi, binned on their HIGH WORDS
    unsigned long long lo, hi;
    unsigned loh;        // high word of lo
    unsigned words;      // high word of hi - high word of lo
    unsigned mult;       // floor(2^32 bins / (words + 1)), capped at 2^32 - 1: bin = floor((kh - loh) mult / 2^32), kh = the key's high word;
                         // 0: a bracket of ONE key (lo == hi), compared in full
    unsigned pad;
};

struct Record {          // a profile value inside a bracket
    unsigned long long key;     // its order-preserving image (pem_quantile.hip key_of)
    unsigned long long col;     // its column (which of the column's brackets holds the key: their high words say)
};

// order-preserving image of a double (negative values reversed, sign bit flipped), and its high word from the value's high word alone
__device__ __forceinline__ unsigned long long order_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ unsigned order_key_high(int high_word) {
    return (unsigned)high_word ^ ((unsigned)(high_word >> 31) | 0x80000000u);
}

// what the counting launch of the evaluation kernel reads and writes (device pointers)
struct CountIO {
    const Bracket* br = nullptr;          // [91][nq]; every bracket ends on whole words and no two of a column overlap (the caller checked)
    int nq = 0;
    unsigned long long* below = nullptr;  // [91][nq] += values below the bracket
    Record* rec = nullptr;                // [waves][cap]: the records of each wave of the launch
    unsigned* rec_count = nullptr;        // [waves]: records the wave produced (more than cap: it ran out of room, flags[0] is set)
    unsigned cap = 0;
    int* flags = nullptr;                 // [0] a wave's region overflowed, [1] a sample with a non-finite profile was met
    // the premask (optional, nq <= 5): per column the high words {certainly below, possibly below, possibly above, certainly above} of the
    // outlier bounds' intervals; per sample the number of its values outside the bounds for certain / uncertain
    const uint4* premask = nullptr;
    uint8_t* row_certain = nullptr;
    uint8_t* row_uncertain = nullptr;
};

// one fused Monte-Carlo launch (the arguments of pem_coupled_mc_f64_dev)
struct McLaunch {
    size_t n = 0;
    unsigned long long first_index = 0, seed = 0;
    unsigned stream_id = 0;
    int kind[15];
    double a[15], b[15];
    double torr2pa = 0, radius = 0;
    double* x_out = nullptr;
    size_t ld = 0;
    double *V_cc = nullptr, *I_B0 = nullptr, *T = nullptr, *j_ion = nullptr, *div_angle = nullptr, *T_c = nullptr;
    uint8_t* invalid = nullptr;
};

// csrc/pem_kernels.hip
__attribute__((visibility("hidden"))) int launch_coupled_mc(const McLaunch& a, hipStream_t st);
__attribute__((visibility("hidden"))) int coupled_count_waves(size_t n, int nq, bool store_profile, unsigned* waves);
__attribute__((visibility("hidden"))) int launch_coupled_mc_count(const McLaunch& a, const CountIO& c, bool store_profile, hipStream_t st);

// csrc/pem_quantile.hip: the selection driven by a producer of counts and records instead of by passes over an array
struct FusedProducer {
    virtual ~FusedProducer() {}
    // the premask (see CountIO): which two of the quantiles are the quartiles, the IQR factor, where the per-sample counts go;
    // pm_done says whether the counting launch produced them (not with unfit bounds: zero, non-finite, a negative factor)
    int pm_q25 = -1, pm_q75 = -1;
    double pm_factor = 0.0;
    uint8_t *pm_certain = nullptr, *pm_uncertain = nullptr;
    int pm_done = 0;
    virtual int pilot(size_t rows, double* dst, hipStream_t st) = 0;                 // write the first `rows` rows of the [n][m] array to dst
    virtual int waves(int nq, unsigned* waves) = 0;                                   // waves of the counting launch
    virtual int count(const CountIO& io, hipStream_t st) = 0;                         // produce the whole array, counting
};
__attribute__((visibility("hidden"))) int quantiles_fused(size_t n, int m, int nq, const uint64_t* rank_prev, const uint64_t* rank_next,
                                                          const double* gamma, double* pilot_rows, FusedProducer& prod, double* out,
                                                          int* fused_ok, hipStream_t st);
// the plain selection (pem_quantiles_strided_f64_dev) on the SECOND workspace of the library: may run on another host thread and stream
// while quantiles_fused is under way (the scalar QoIs of a campaign, selected while the profile's records are being sorted)
// `plan` (optional): the first ceil(n / 32) rows exist before the rest does -- a campaign's pilot evaluation wrote them -- so the
// selection takes its pilot form with THOSE rows as the subsample (exchangeable rows: a Monte-Carlo design) and calls
// `before_full(ctx, st)` before its first pass over all n rows: the caller blocks there until the launch that writes them is
// under way and makes `st` wait for it.  The subsample's passes then run while that launch does.
struct SidePlan {
    int (*before_full)(void* ctx, hipStream_t st) = nullptr;
    void* ctx = nullptr;
};
__attribute__((visibility("hidden"))) int quantiles_side(size_t n, int m, const double* data, size_t ld, size_t cs, int nq,
                                                         const uint64_t* rank_prev, const uint64_t* rank_next, const double* gamma, double* out,
                                                         hipStream_t st, const SidePlan* plan = nullptr);

}  // namespace pem
