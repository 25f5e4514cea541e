// The read-side twin of write_pattern.hip: 0.94 GB read per launch (eight buffers in rotation), every lane sums what it reads
// and the wave writes 8 bytes.  Persistent waves reading `ppt`-KB tiles against a one-shot grid in address order.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/read_pattern.hip -o /tmp/rp && /tmp/rp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void reader(const f64x2* __restrict__ in, double* __restrict__ out, long long pieces_total, int ppt, int oneshot) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = oneshot ? (1LL << 62) : (long long)gridDim.x * 4;
    const long long ntiles = pieces_total / ppt;
    double s = 0.0;
    for (long long t = wave; t < ntiles; t += nwaves) {
        const f64x2* src = in + t * ppt * 64 + lane;
        f64x2 v[12];
        int p = 0;
        for (; p + 12 <= ppt; p += 12) {
#pragma unroll
            for (int u = 0; u < 12; ++u) v[u] = src[(long long)(p + u) * 64];
#pragma unroll
            for (int u = 0; u < 12; ++u) s += v[u].x + v[u].y;
        }
        for (; p < ppt; ++p) {
            const f64x2 w = src[(long long)p * 64];
            s += w.x + w.y;
        }
        if (oneshot) break;
    }
    if (s == 12345.678) out[wave] = s;      // (never true: keeps the loads)
}

int main() {
    const long long bytes = 940000000LL / 1024 * 1024, pieces = bytes / 1024;
    const int NB = 8;
    f64x2* buf[NB];
    double* out;
    hipMalloc(&out, 8 << 20);
    for (int i = 0; i < NB; ++i) {
        hipMalloc(&buf[i], bytes + 65536);
        hipMemset(buf[i], 0, bytes);
    }
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto timeit = [&](int blocks, int ppt, int oneshot) {
        const long long use = pieces / ppt * ppt;
        if (oneshot) blocks = (int)((use / ppt + 3) / 4);
        for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(reader, dim3(blocks), dim3(256), 0, 0, buf[i], out, use, ppt, oneshot);
        hipDeviceSynchronize();
        hipEventRecord(a);
        const int reps = 40;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(reader, dim3(blocks), dim3(256), 0, 0, buf[i % NB], out, use, ppt, oneshot);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("%s, %3d KB per wave and visit, %6d workgroups: %7.1f us per 0.94 GB = %5.2f TB/s\n", oneshot ? "one-shot  " : "persistent", ppt, blocks,
               ms / reps * 1e3, use * 1024.0 / (ms / reps * 1e-3) / 1e12);
    };
    for (int ppt : {1, 4, 12, 46}) {
        for (int blocks : {512, 1024, 2048}) timeit(blocks, ppt, 0);
        timeit(0, ppt, 1);
    }
    return 0;
}
