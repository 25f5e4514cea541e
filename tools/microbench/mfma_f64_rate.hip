// Issue rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on gfx950 (one wave per SIMD, independent accumulators).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f64_rate.hip -o /tmp/mfma && /tmp/mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, long long* cyc) {
    f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void fma_loop(double* out, int iters, long long* cyc) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double x = 1.0 + 1e-9, y = 1e-3;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a0 = fma(a0, x, y); a1 = fma(a1, x, y); a2 = fma(a2, x, y); a3 = fma(a3, x, y);
        a4 = fma(a4, x, y); a5 = fma(a5, x, y); a6 = fma(a6, x, y); a7 = fma(a7, x, y);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h[256];
    hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(256), 0, 0, out, iters, cyc); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("mfma_f64_16x16x4: %.1f shader cycles per MFMA per SIMD (4 independent chains, 1 wave/SIMD, 256 CUs busy); %.2f TFLOP/s chip\n",
               (double)h[0] / (4.0 * iters), 256.0 * 4 * 4.0 * iters * 2048 / (ms * 1e-3) / 1e12);
        hipEventRecord(a); hipLaunchKernelGGL(fma_loop, dim3(256), dim3(256), 0, 0, out, iters, cyc); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b); hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("v_fma_f64       : %.2f shader cycles per wave64 FMA per SIMD (8 independent chains, 1 wave/SIMD); %.2f TFLOP/s chip\n",
               (double)h[0] / (8.0 * iters), 256.0 * 4 * 8.0 * iters * 128 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
