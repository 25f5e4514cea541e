// Does HBM care which wave writes what?  The coupled kernel's 2048 persistent waves each stream their own 46.6 KB tile
// (16 samples x 91 doubles per round, four rounds), so at any instant the chip writes 2048 separate 1-KB pieces 46 KB apart;
// a fill kernel's waves write ADJACENT 1-KB pieces.  Same bytes, same 16-byte non-temporal stores, 8 buffers in rotation
// (7.5 GB: nothing stays in the 256 MB Infinity Cache).  TILE = bytes a wave writes contiguously before it jumps.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/write_pattern.hip -o /tmp/wp && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void writer(f64x2* __restrict__ out, long long pieces_total, int pieces_per_tile) {
    // piece = 1 KB = 64 lanes x 16 B.  Wave w takes tiles w, w + nwaves, ...; a tile is pieces_per_tile consecutive pieces.
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const long long ntiles = pieces_total / pieces_per_tile;
    const f64x2 v = {1.0 + lane, 2.0};
    for (long long t = wave; t < ntiles; t += nwaves) {
        f64x2* dst = out + t * pieces_per_tile * 64 + lane;
        for (int p = 0; p < pieces_per_tile; ++p) {
            if (NT) __builtin_nontemporal_store(v, dst + (long long)p * 64);
            else dst[(long long)p * 64] = v;
        }
    }
}

// one-shot grid: every wave writes ONE tile of pieces_per_tile pieces and exits (no persistent loop)
template <bool NT>
__global__ __launch_bounds__(256) void writer_once(f64x2* __restrict__ out, long long pieces_total, int pieces_per_tile) {
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((t + 1) * pieces_per_tile > pieces_total) return;
    const f64x2 v = {1.0 + lane, 2.0};
    f64x2* dst = out + t * pieces_per_tile * 64 + lane;
    for (int p = 0; p < pieces_per_tile; ++p) {
        if (NT) __builtin_nontemporal_store(v, dst + (long long)p * 64);
        else dst[(long long)p * 64] = v;
    }
}

// E8: one-shot, ONE tile per wave, workgroups that share a XCD (b, b + 8, ...) take one contiguous eighth of the tiles in order
// (csrc/pem_common.h xcd_contiguous_block: what plume_r1_kernel's profile modes do since round 3)
__global__ __launch_bounds__(256) void writer_once_xcd(f64x2* __restrict__ out, long long pieces_total, int pieces_per_tile) {
    const int lane = threadIdx.x & 63;
    const unsigned q = gridDim.x >> 3, r = gridDim.x & 7, x = blockIdx.x & 7;
    const long long vb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
    const long long t = vb * 4 + (threadIdx.x >> 6);
    if ((t + 1) * pieces_per_tile > pieces_total) return;
    const f64x2 v = {1.0 + lane, 2.0};
    f64x2* dst = out + t * pieces_per_tile * 64 + lane;
    for (int p = 0; p < pieces_per_tile; ++p) __builtin_nontemporal_store(v, dst + (long long)p * 64);
}
// E9 (round 4): workgroup-cooperative rounds.  A workgroup still owns four consecutive 46-KB tiles (one per wave's
// prelude), but writes them ONE AFTER THE OTHER, all four waves on the same tile -- wave w its quarter (12 / 12 / 11 / 11 KB), a
// workgroup barrier between tiles as the real kernel would need one to publish the next tile's parameters: a workgroup is then one
// sequential 184-KB stream (64 per XCD) where E8's four waves are four streams 46 KB apart (256 per XCD).  Same bytes, same order of
// workgroups over the XCDs as E8.
__global__ __launch_bounds__(256) void writer_coop_xcd(f64x2* __restrict__ out, long long pieces_total, int pieces_per_tile, int barrier) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned q = gridDim.x >> 3, r = gridDim.x & 7, x = blockIdx.x & 7;
    const long long vb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
    const f64x2 v = {1.0 + lane, 2.0};
    const int base = pieces_per_tile / 4, extra = pieces_per_tile & 3;
    const int p0 = w * base + (w < extra ? w : extra), np = base + (w < extra ? 1 : 0);
    for (int i = 0; i < 4; ++i) {
        const long long t = vb * 4 + i;
        if ((t + 1) * pieces_per_tile <= pieces_total) {
            f64x2* dst = out + (t * pieces_per_tile + p0) * 64 + lane;
            for (int p = 0; p < np; ++p) __builtin_nontemporal_store(v, dst + (long long)p * 64);
        }
        if (barrier) __syncthreads();
    }
}
// E1: one-shot, 1 KB per wave, but workgroup b writes chunk perm(b): the address order of the dispatch order is destroyed
__global__ __launch_bounds__(256) void writer_scrambled(f64x2* __restrict__ out, long long nblocks, long long mult) {
    const int lane = threadIdx.x & 63;
    const long long b = ((long long)blockIdx.x * mult) % nblocks;
    const f64x2 v = {1.0 + lane, 2.0};
    out[(b * 4 + (threadIdx.x >> 6)) * 64 + lane] = v;
}
// E5 / E6: one-shot, 1 KB per wave (4 KB per workgroup), workgroup b -> chunk c(b):
//   mode 1: c = b + 1 (mod N): XCD x (= b mod 8) writes the 4-KB chunks of residue x + 1 -- still one residue class per XCD
//   mode 2: c = (b mod 8) * (N / 8) + b / 8: XCD x sweeps the x-th eighth of the buffer: every XCD visits every residue class
//   mode 3: c = b with the three low bits replaced by a hash of the rest: the residue a XCD writes changes from chunk to chunk
__global__ __launch_bounds__(256) void writer_affinity(f64x2* __restrict__ out, long long nblocks, int mode) {
    const int lane = threadIdx.x & 63;
    const long long b = blockIdx.x;
    long long c = b;
    if (mode == 1) c = (b + 1) % nblocks;
    if (mode == 2) c = (b & 7) * (nblocks / 8) + (b >> 3);
    if (mode == 3) c = (b & ~7LL) | ((b ^ (b >> 3) ^ (b >> 7)) & 7);
    const f64x2 v = {1.0 + lane, 2.0};
    out[(c * 4 + (threadIdx.x >> 6)) * 64 + lane] = v;
}
// E7: one-shot, 1 KB per wave; XCD x (= b mod 8) owns the x-th eighth of the buffer and writes it as S interleaved sequential
// streams: its j-th workgroup (j = b / 8) appends a 4-KB chunk to stream j mod S.  S = 1 is E5's third mode.
__global__ __launch_bounds__(256) void writer_streams(f64x2* __restrict__ out, long long nblocks, int S) {
    const int lane = threadIdx.x & 63;
    const long long b = blockIdx.x, x = b & 7, j = b >> 3, per_xcd = nblocks / 8, per_stream = per_xcd / S;
    const long long st = j % S, pos = j / S;
    if (pos >= per_stream) return;
    const long long c = x * per_xcd + st * per_stream + pos;
    const f64x2 v = {1.0 + lane, 2.0};
    out[(c * 4 + (threadIdx.x >> 6)) * 64 + lane] = v;
}
// E2: one-shot, a workgroup owns a region of 4 * ppt pieces; its four waves interleave piece by piece (wave w: pieces w, w + 4, ...)
__global__ __launch_bounds__(256) void writer_interleaved(f64x2* __restrict__ out, long long pieces_total, int ppt) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long base = (long long)blockIdx.x * 4 * ppt;
    if (base + 4LL * ppt > pieces_total) return;
    const f64x2 v = {1.0 + lane, 2.0};
    for (int p = 0; p < ppt; ++p) out[(base + 4LL * p + w) * 64 + lane] = v;
}

int main() {
    const long long bytes = 940000000LL / 1024 * 1024, pieces = bytes / 1024;
    const int NB = 8;
    f64x2* buf[NB];
    for (int i = 0; i < NB; ++i) hipMalloc(&buf[i], bytes + 65536);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int nt = 1; nt >= 0; --nt) {
        for (int ppt : {1, 12, 46}) {
            for (int blocks : {512, 1024, 2048, 4096, 16384}) {
                const long long use = pieces / ppt * ppt;
                auto launch = [&](int i) {
                    if (nt) hipLaunchKernelGGL(writer<true>, dim3(blocks), dim3(256), 0, 0, buf[i % NB], use, ppt);
                    else hipLaunchKernelGGL(writer<false>, dim3(blocks), dim3(256), 0, 0, buf[i % NB], use, ppt);
                };
                for (int i = 0; i < NB; ++i) launch(i);
                hipDeviceSynchronize();
                hipEventRecord(a);
                const int reps = 40;
                for (int i = 0; i < reps; ++i) launch(i);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                printf("%s stores, contiguous run per wave %4d KB, %5d workgroups of 4 waves: %7.1f us per 0.94 GB = %5.2f TB/s\n",
                       nt ? "nt   " : "plain", ppt, blocks, ms / reps * 1e3, use * 1024.0 / (ms / reps * 1e-3) / 1e12);
            }
        }
    }
    for (int nt = 1; nt >= 0; --nt) {
        for (int ppt : {1, 4, 12, 46, 184}) {
            const long long use = pieces / ppt * ppt, tiles = use / ppt;
            const unsigned blocks = (unsigned)((tiles + 3) / 4);
            auto launch = [&](int i) {
                if (nt) hipLaunchKernelGGL(writer_once<true>, dim3(blocks), dim3(256), 0, 0, buf[i % NB], use, ppt);
                else hipLaunchKernelGGL(writer_once<false>, dim3(blocks), dim3(256), 0, 0, buf[i % NB], use, ppt);
            };
            for (int i = 0; i < NB; ++i) launch(i);
            hipDeviceSynchronize();
            hipEventRecord(a);
            const int reps = 40;
            for (int i = 0; i < reps; ++i) launch(i);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("%s stores, ONE tile of %4d KB per wave, %7u workgroups (one-shot grid): %7.1f us per 0.94 GB = %5.2f TB/s\n",
                   nt ? "nt   " : "plain", ppt, blocks, ms / reps * 1e3, use * 1024.0 / (ms / reps * 1e-3) / 1e12);
        }
    }
    auto timeit = [&](auto launch, const char* what, double bytes_written) {
        for (int i = 0; i < NB; ++i) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(a);
        const int reps = 40;
        for (int i = 0; i < reps; ++i) launch(i);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("%s: %7.1f us = %5.2f TB/s\n", what, ms / reps * 1e3, bytes_written / (ms / reps * 1e-3) / 1e12);
    };
    {
        const long long nblocks = pieces / 4;
        for (long long mult : {1LL, 7919LL, 104729LL}) {
            char what[128];
            snprintf(what, sizeof what, "E1 one-shot 1 KB per wave, workgroup b -> chunk (b * %lld) mod N", mult);
            timeit([&](int i) { hipLaunchKernelGGL(writer_scrambled, dim3((unsigned)nblocks), dim3(256), 0, 0, buf[i % NB], nblocks, mult); }, what,
                   nblocks * 4096.0);
        }
        {
            const long long nb8 = nblocks / 8 * 8;
            const char* names[4] = {"c = b (address order = dispatch order)", "c = b + 1 (each XCD one residue class, shifted)",
                                    "c: XCD x sweeps the x-th eighth of the buffer", "c: low three bits hashed (residue per XCD changes every chunk)"};
            for (int mode = 0; mode < 4; ++mode) {
                char what[160];
                snprintf(what, sizeof what, "E5 one-shot 1 KB per wave, %s", names[mode]);
                timeit([&](int i) { hipLaunchKernelGGL(writer_affinity, dim3((unsigned)nb8), dim3(256), 0, 0, buf[i % NB], nb8, mode); }, what, nb8 * 4096.0);
            }
        }
        for (int S : {1, 2, 4, 8, 16, 64, 256}) {
            const long long nb8 = nblocks / 8 * 8;
            char what[160];
            snprintf(what, sizeof what, "E7 one-shot 1 KB per wave, every XCD writes its eighth as %3d interleaved sequential streams", S);
            timeit([&](int i) { hipLaunchKernelGGL(writer_streams, dim3((unsigned)nb8), dim3(256), 0, 0, buf[i % NB], nb8, S); }, what,
                   (double)(nb8 / 8 / S * S * 8) * 4096.0);
        }
        for (int ppt : {2, 3}) {     // E3: one-shot, 2 and 3 KB per wave (adjacent pieces)
            const long long use = pieces / ppt * ppt, tiles = use / ppt;
            char what[128];
            snprintf(what, sizeof what, "E3 one-shot, ONE tile of %d KB per wave (plain stores)", ppt);
            timeit([&](int i) { hipLaunchKernelGGL(writer_once<false>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, buf[i % NB], use, ppt); },
                   what, use * 1024.0);
        }
        for (int blocks : {489, 500, 611, 1000, 1223}) {   // E4: persistent, 1 KB per wave and iteration, wave counts that are not powers of two
            char what[128];
            snprintf(what, sizeof what, "E4 persistent, 1 KB per wave and iteration, %d workgroups (plain stores)", blocks);
            timeit([&](int i) { hipLaunchKernelGGL(writer<false>, dim3(blocks), dim3(256), 0, 0, buf[i % NB], pieces, 1); }, what, pieces * 1024.0);
        }
        for (int ppt : {3, 12, 46}) {
            const long long use = pieces / (4 * ppt) * (4 * ppt);
            char what[160];
            snprintf(what, sizeof what, "E2 one-shot, workgroup region %4d KB, its 4 waves interleaved piece by piece", 4 * ppt);
            timeit([&](int i) { hipLaunchKernelGGL(writer_interleaved, dim3((unsigned)(use / (4 * ppt))), dim3(256), 0, 0, buf[i % NB], use, ppt); },
                   what, use * 1024.0);
        }
    }
    for (int ppt : {1, 4, 12, 46, 184}) {   // E8: one-shot, XCD-contiguous tile order (non-temporal stores)
        const long long use = pieces / ppt * ppt, tiles = use / ppt;
        char what[160];
        snprintf(what, sizeof what, "E8 one-shot, ONE tile of %3d KB per wave, XCD-contiguous tile order (nt stores)", ppt);
        timeit([&](int i) { hipLaunchKernelGGL(writer_once_xcd, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, buf[i % NB], use, ppt); }, what,
               use * 1024.0);
    }
    for (int barrier : {0, 1}) {            // E9 beside E8 at the kernel's tile size, interleaved three times
        for (int rep = 0; rep < 3; ++rep) {
            const int ppt = 46;
            const long long use = pieces / ppt * ppt, tiles = use / ppt;
            char what[200];
            snprintf(what, sizeof what, "E8 one-shot, ONE tile of  46 KB per wave, XCD-contiguous tile order (nt stores) [A/B %d]", rep);
            timeit([&](int i) { hipLaunchKernelGGL(writer_once_xcd, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, buf[i % NB], use, ppt); }, what,
                   use * 1024.0);
            snprintf(what, sizeof what, "E9 one-shot, workgroup writes its four 46-KB tiles one after the other, four waves per tile%s [A/B %d]",
                     barrier ? ", barrier between tiles" : "", rep);
            timeit([&](int i) { hipLaunchKernelGGL(writer_coop_xcd, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, 0, buf[i % NB], use, ppt, barrier); },
                   what, use * 1024.0);
        }
    }
    return 0;
}
