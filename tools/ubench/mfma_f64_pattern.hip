// micro-benchmark: v_mfma_f64_16x16x4_f64 in the full-stat pass's register pattern (2 row tiles x 8
// sub-tiles = 16 accumulators, A changes every 8 ops, B cycles through 8 registers), 8 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(double *out, int iters, double a0, double b0)
{
    d4 acc[2][8];
    for (int t = 0; t < 2; t++) for (int u = 0; u < 8; u++) acc[t][u] = d4{0, 0, 0, 0};
    double a[4][2], b[8];
    for (int s = 0; s < 4; s++) for (int t = 0; t < 2; t++) a[s][t] = a0 + threadIdx.x * 1e-9 + s + 0.5 * t;
    for (int u = 0; u < 8; u++) b[u] = b0 + u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int u = 0; u < 8; u++) acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][t], b[u], acc[t][u], 0, 0, 0);
        }
        asm volatile("" : "+v"(a[0][0]), "+v"(b[0]));
    }
    double s = 0;
    for (int t = 0; t < 2; t++) for (int u = 0; u < 8; u++) s += acc[t][u][0] + acc[t][u][1] + acc[t][u][2] + acc[t][u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    const int threads = 512, blocks = 256, iters = 4000;
    double *out; hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double mfma = (double)iters * 64 * (threads / 64) * blocks;
        printf("pattern: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", ms, mfma * 2048 / (ms * 1e-3) / 1e12,
               2.4e9 * (ms * 1e-3) / (mfma / (256.0 * 4)));
    }
    return 0;
}
