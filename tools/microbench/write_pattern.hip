// Does HBM care which wave writes what?  The coupled kernel's 2048 persistent waves each stream their own 46.6 KB tile
// (16 samples x 91 doubles per round, four rounds), so at any instant the chip writes 2048 separate 1-KB pieces 46 KB apart;
// a fill kernel's waves write ADJACENT 1-KB pieces.  Same bytes, same 16-byte non-temporal stores, 8 buffers in rotation
// (7.5 GB: nothing stays in the 256 MB Infinity Cache).  TILE = bytes a wave writes contiguously before it jumps.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/write_pattern.hip -o /tmp/wp && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void writer(f64x2* __restrict__ out, long long pieces_total, int pieces_per_tile) {
    // piece = 1 KB = 64 lanes x 16 B.  Wave w takes tiles w, w + nwaves, ...; a tile is pieces_per_tile consecutive pieces.
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const long long ntiles = pieces_total / pieces_per_tile;
    const f64x2 v = {1.0 + lane, 2.0};
    for (long long t = wave; t < ntiles; t += nwaves) {
        f64x2* dst = out + t * pieces_per_tile * 64 + lane;
        for (int p = 0; p < pieces_per_tile; ++p) __builtin_nontemporal_store(v, dst + (long long)p * 64);
    }
}

int main() {
    const long long bytes = 940000000LL / 1024 * 1024, pieces = bytes / 1024;
    const int NB = 8;
    f64x2* buf[NB];
    for (int i = 0; i < NB; ++i) hipMalloc(&buf[i], bytes + 65536);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int ppt : {1, 4, 12, 46, 184, 736}) {
        for (int blocks : {512, 1024}) {
            const long long use = pieces / ppt * ppt;
            for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(writer, dim3(blocks), dim3(256), 0, 0, buf[i], use, ppt);
            hipDeviceSynchronize();
            hipEventRecord(a);
            const int reps = 40;
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(writer, dim3(blocks), dim3(256), 0, 0, buf[i % NB], use, ppt);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("contiguous run per wave %4d KB, %4d workgroups of 4 waves: %7.1f us per 0.94 GB = %5.2f TB/s\n", ppt, blocks,
                   ms / reps * 1e3, use * 1024.0 / (ms / reps * 1e-3) / 1e12);
        }
    }
    return 0;
}
