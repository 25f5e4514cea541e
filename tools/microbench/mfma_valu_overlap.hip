// Does v_mfma_f64_16x16x4_f64 run beside fp64 VALU work, or do the two share the SIMD's fp64 units on gfx950?
// One loop iteration = [one MFMA] + NV independent v_fma_f64, W waves per SIMD on every CU.  If the matrix pipe is a
// separate unit the MFMA's 16 passes hide behind the VALU instructions; if not, times add.  Decides whether a
// register-fed MFMA contraction inside the fused compression mode's angle loop (csrc/pem_kernels.hip) can be "free".
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NV, bool MFMA>
__global__ void mix_loop(double* out, int iters) {
    f64x4 acc = {0, 0, 0, 0};
    double d[NV > 0 ? NV : 1];
#pragma unroll
    for (int j = 0; j < NV; ++j) d[j] = 1.0 + threadIdx.x * 1e-6 + j;
    const double x = 1.0 + threadIdx.x * 1e-9, y = 0.5, xd = 1.0 + 1e-9, yd = 1e-3;
    for (int i = 0; i < iters; ++i) {
        if (MFMA) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) d[j] = __builtin_fma(d[j], xd, yd);
    }
    double s = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
    for (int j = 0; j < NV; ++j) s += d[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static float run(K kern, int w, double* out) {
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256 * w), 0, 0, out, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best * 1e6f / iters;   // ns per loop iteration
}

template <int NV>
static void line(int w, double* out) {
    const float v = NV > 0 ? run(mix_loop<NV, false>, w, out) : 0.0f, m = run(mix_loop<0, true>, w, out), both = run(mix_loop<NV, true>, w, out);
    printf("W=%d waves/SIMD, 1 MFMA + %2d v_fma_f64 per iteration: VALU alone %6.1f ns, MFMA alone %6.1f ns, together %6.1f ns  (sum %6.1f, max %6.1f)\n",
           w, NV, v, m, both, v + m, v > m ? v : m);
}

int main() {
    double* out;
    hipMalloc(&out, 256 * 1024 * 8);
    for (int w : {1, 2, 4}) {
        line<8>(w, out);
        line<16>(w, out);
        line<32>(w, out);
        line<48>(w, out);
    }
    return 0;
}
