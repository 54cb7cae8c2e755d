// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (one number the roofline needs)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(double *out, int iters, double a0, double b0)
{
    d4 acc[NACC];
    for (int t = 0; t < NACC; t++) acc[t] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < NACC; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int threads, int blocks)
{
    double *out; hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)iters * NACC * (threads / 64) * blocks;
    double tf = mfma * 2048 / (ms * 1e-3) / 1e12;
    printf("NACC=%d threads=%d blocks=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD @2.4GHz\n", NACC, threads, blocks, ms, tf,
           2.4e9 * (ms * 1e-3) / (mfma / (256.0 * 4)));
    hipFree(out);
}
int main()
{
    run<1>(256, 256); run<4>(256, 256); run<5>(1024, 256); run<8>(256, 256); run<4>(1024, 256);
    return 0;
}
