// The node-value contraction of the sparse-grid surrogate (csrc/pem_surrogate.hip) -- out[point][o] += w[point][node] Y[node][o],
// the weights w generated per point on the fly -- on the fp64 matrix pipe against the VALU form the kernel uses, at the widths the
// field surrogate has (round 4): n_out = 3 scalars + r latents = 9 (r = 6 under the PEM-v0 priors) and the kernel's maximum 16.
// "measure the node-value contraction on v_mfma_f64_16x16x4_f64 against the VALU form, keep whichever wins".
//
//   VALU   lane = point: per node one weight (a 2-multiply recurrence standing in for the basis product) and NO FMAs against
//          the node's NO values, which are wave-uniform and come through the scalar cache;
//   MFMA   wave = 4 groups of 16 points: A[16 points][4 nodes] = the weights (lane (p, k) produces the weight of point p and
//          node k0 + k), B[4 nodes][16 outputs] = the node values (one per lane, from LDS), C[16][16] accumulates -- 4 MFMAs per
//          4 nodes cover the wave's 64 points; outputs beyond NO are padding.
// Both forms do the same useful work: 64 points x K nodes x NO multiply-adds per wave.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench/contraction_mfma_vs_valu.hip -o /tmp/cmv && /tmp/cmv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int K = 729;          // nodes of a 9 x 9 x 9 grid

template <int NO>
__global__ __launch_bounds__(256) void valu_kernel(const double* __restrict__ Y, double* __restrict__ out, int reps) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = 0.0;
    const double q = 1.0 - 1e-6 * (tid & 63);
    for (int r = 0; r < reps; ++r) {
        double w = 1.0 + 1e-9 * r, g = q;
        for (int k = 0; k < K; ++k) {
            w *= g;                      // the weight of (point, node): stands in for the running basis product
            g *= q;
#pragma unroll
            for (int o = 0; o < NO; ++o) acc[o] = __builtin_fma(w, Y[k * NO + o], acc[o]);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int o = 0; o < NO; ++o) s += acc[o];
    out[tid] = s;
}

template <int NO>
__global__ __launch_bounds__(256) void mfma_kernel(const double* __restrict__ Y, double* __restrict__ out, int reps) {
    __shared__ double ys[K * 16 + 64];                      // node values padded to 16 outputs
    for (int i = threadIdx.x; i < K * 16; i += 256) ys[i] = (i & 15) < NO ? Y[(i >> 4) * NO + (i & 15)] : 0.0;
    __syncthreads();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const int p = lane & 15, kk = lane >> 4;
    f64x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f64x4{0, 0, 0, 0};
    double q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) q[g] = 1.0 - 1e-6 * (16 * g + p);
    for (int r = 0; r < reps; ++r) {
        double w[4], gq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            w[g] = 1.0 + 1e-9 * r;
            gq[g] = q[g];
        }
        for (int k0 = 0; k0 + 4 <= K; k0 += 4) {
            const double b = ys[(k0 + kk) * 16 + p];        // B[k][n]: lane (n = lane % 16, k = lane / 16)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                w[g] *= gq[g];                              // lane (p, kk): the weight of point 16 g + p and node k0 + kk
                gq[g] *= q[g];
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(w[g], b, acc[g], 0, 0, 0);
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    out[tid] = s;
}

template <class Kern>
static double time_ms(Kern kern, const double* Y, double* out, int blocks, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, Y, out, reps);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (it > 0 && ms < best) best = ms;
    }
    return best;
}

template <int NO>
static void compare(const double* Y, double* out) {
    const int blocks = 256 * 8, reps = 40;                  // 8 workgroups of 4 waves per CU
    const double points = (double)blocks * 256, useful = points * reps * (K / 4 * 4) * NO * 2.0;
    const double tv = time_ms(valu_kernel<NO>, Y, out, blocks, reps), tm = time_ms(mfma_kernel<NO>, Y, out, blocks, reps);
    printf("n_out %2d: VALU %7.3f ms = %6.2f useful TFLOP/s | MFMA f64 16x16x4 %7.3f ms = %6.2f useful TFLOP/s (%5.2f issued) | VALU / MFMA time %.2f\n",
           NO, tv, useful / tv / 1e9, tm, useful / tm / 1e9, useful / NO * 16 / tm / 1e9, tv / tm);
}

int main() {
    std::vector<double> h((size_t)K * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
    double *Y, *out;
    hipMalloc(&Y, h.size() * sizeof(double));
    hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(double));
    hipMemcpy(Y, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
    printf("node-value contraction, 64 points x %d nodes x n_out outputs per wave, 524288 points x 40 repetitions, one MI355X\n", K);
    compare<3>(Y, out);
    compare<9>(Y, out);
    compare<12>(Y, out);
    compare<16>(Y, out);
    return 0;
}
