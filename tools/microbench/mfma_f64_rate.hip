// Issue rates on gfx950 that the design leans on: v_mfma_f64_16x16x4_f64 vs v_fma_f64 (is the fp64 matrix pipe worth
// using for the rank-r contractions of csrc/pem_svd.hip / pem_surrogate.hip?), and the fp32 instructions the
// fp32-arithmetic reduced-QoI kernel is built from (v_fma_f32, v_pk_fma_f32, v_exp_f32, v_log_f32, v_rcp_f32) beside
// their fp64 counterparts.  Each loop runs W = 1, 2, 4 waves per SIMD on every CU with NACC independent accumulators.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f64_rate.hip -o /tmp/mfma && /tmp/mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ void mfma_loop(double* out, int iters, long long* cyc) {
    f64x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f64x4{0, 0, 0, 0};
    const double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[j], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][j & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// OP: 0 v_fma_f64, 1 v_fma_f32, 2 v_pk_fma_f32, 3 v_exp_f32, 4 v_log_f32, 5 v_rcp_f32, 6 v_rcp_f64, 7 v_sqrt_f64, 8 v_mul_f64
template <int OP, int NACC>
__global__ void valu_loop(double* out, int iters, long long* cyc) {
    double d[NACC];
    float f[NACC];
    f32x2 p[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        d[j] = 1.0 + threadIdx.x * 1e-6 + j;
        f[j] = 1.0f + threadIdx.x * 1e-3f + j;
        p[j] = f32x2{f[j], f[j] + 0.5f};
    }
    const double xd = 1.0 + 1e-9, yd = 1e-3;
    const float xf = 1.0f + 1e-6f, yf = 1e-3f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            if (OP == 0) d[j] = __builtin_fma(d[j], xd, yd);
            if (OP == 1) f[j] = __builtin_fmaf(f[j], xf, yf);
            if (OP == 2) p[j] = __builtin_elementwise_fma(p[j], f32x2{xf, xf}, f32x2{yf, yf});
            if (OP == 3) f[j] = __builtin_amdgcn_exp2f(f[j]) * 0.25f;     // v_exp_f32 (+ a mul to keep values bounded)
            if (OP == 4) f[j] = __builtin_amdgcn_logf(f[j]) + 3.0f;        // v_log_f32
            if (OP == 5) f[j] = __builtin_amdgcn_rcpf(f[j]) + 1.0f;        // v_rcp_f32
            if (OP == 6) d[j] = __builtin_amdgcn_rcp(d[j]) + 1.0;          // v_rcp_f64
            if (OP == 7) d[j] = __builtin_amdgcn_sqrt(d[j]) + 1.0;         // v_sqrt_f64
            if (OP == 8) d[j] = d[j] * xd;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += d[j] + f[j] + p[j].x + p[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
static void run(const char* name, K kern, int waves_per_simd, int nacc, double flop_per_inst, int extra_per_inst, double* out, long long* cyc) {
    long long h[256];
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256 * waves_per_simd), 0, 0, out, iters, cyc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // cycles per instruction per SIMD: one wave's loop time / (instructions of all waves sharing the SIMD)
    const double per_inst = (double)h[0] / ((double)nacc * iters * waves_per_simd);
    const double insts = 256.0 * 4 * waves_per_simd * (double)nacc * iters;
    printf("%-18s W=%d waves/SIMD, %2d accumulators: %7.2f shader cycles per instruction per SIMD", name, waves_per_simd, nacc, per_inst);
    if (flop_per_inst > 0) printf("; %7.2f TFLOP/s chip (wall)", insts * flop_per_inst / (best * 1e-3) / 1e12);
    else printf("; %7.2f G inst/s chip (wall)", insts / (best * 1e-3) / 1e9);
    if (extra_per_inst) printf("  [+%d bounding VALU op per instruction in the loop]", extra_per_inst);
    printf("\n");
}

int main() {
    double* out;
    long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8);
    hipMalloc(&cyc, 256 * 8);
    for (int w : {1, 2, 4}) {
        run("mfma_f64_16x16x4", mfma_loop<4>, w, 4, 2048.0, 0, out, cyc);
        run("mfma_f64_16x16x4", mfma_loop<8>, w, 8, 2048.0, 0, out, cyc);
        run("mfma_f64_16x16x4", mfma_loop<16>, w, 16, 2048.0, 0, out, cyc);
    }
    for (int w : {1, 2, 4}) {
        run("v_fma_f64", valu_loop<0, 8>, w, 8, 128.0, 0, out, cyc);
        run("v_mul_f64", valu_loop<8, 8>, w, 8, 64.0, 0, out, cyc);
        run("v_fma_f32", valu_loop<1, 8>, w, 8, 128.0, 0, out, cyc);
        run("v_pk_fma_f32", valu_loop<2, 8>, w, 8, 256.0, 0, out, cyc);
        run("v_exp_f32", valu_loop<3, 8>, w, 8, 0, 1, out, cyc);
        run("v_log_f32", valu_loop<4, 8>, w, 8, 0, 1, out, cyc);
        run("v_rcp_f32", valu_loop<5, 8>, w, 8, 0, 1, out, cyc);
        run("v_rcp_f64", valu_loop<6, 8>, w, 8, 0, 1, out, cyc);
        run("v_sqrt_f64", valu_loop<7, 8>, w, 8, 0, 1, out, cyc);
    }
    return 0;
}
