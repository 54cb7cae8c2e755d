// micro-benchmark: v_mfma_f64_16x16x4_f64 issued as chains of 4 dependent ops per accumulator (the
// "tile outer, k-step inner" order) vs. the same ops interleaved over 4 accumulators; 8 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CHAIN>
__global__ __launch_bounds__(512) void k(double *out, int iters, double a0, double b0)
{
    d4 acc[16];
    for (int t = 0; t < 16; t++) acc[t] = d4{0, 0, 0, 0};
    double a[4], b[4];
    for (int s = 0; s < 4; s++) { a[s] = a0 + threadIdx.x * 1e-9 + s; b[s] = b0 + s; }
    for (int i = 0; i < iters; i++) {
        if (CHAIN) {
#pragma unroll
            for (int t = 0; t < 16; t++)
#pragma unroll
                for (int s = 0; s < 4; s++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int tg = 0; tg < 16; tg += 4)
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int t = tg; t < tg + 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc[t], 0, 0, 0);
        }
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    double s = 0;
    for (int t = 0; t < 16; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAIN> void run()
{
    const int threads = 512, blocks = 256, iters = 4000;
    double *out; hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAIN>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAIN>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)iters * 64 * (threads / 64) * blocks;
    printf("%s: %.1f cycles/MFMA/SIMD @2.4GHz (%.1f TFLOP/s)\n", CHAIN ? "chains of 4 dependent ops " : "interleaved over 4 tiles  ",
           2.4e9 * (ms * 1e-3) / (mfma / (256.0 * 4)), mfma * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() { run<1>(); run<0>(); run<1>(); run<0>(); return 0; }
