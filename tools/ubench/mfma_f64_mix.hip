// micro-benchmark: does other VALU work (address arithmetic, f64 multiplies, LDS reads) issued between
// v_mfma_f64_16x16x4_f64 groups slow the matrix pipe?  Same register pattern as mfma_f64_pattern.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NV, int LDS>
__global__ __launch_bounds__(512) void k(double *out, int iters, double a0, double b0, int ld)
{
    __shared__ double sh[16 * 144 * 2];
    for (int i = threadIdx.x; i < 16 * 144 * 2; i += blockDim.x) sh[i] = 1.0 + i * 1e-6;
    __syncthreads();
    d4 acc[2][8];
    for (int t = 0; t < 2; t++) for (int u = 0; u < 8; u++) acc[t][u] = d4{0, 0, 0, 0};
    double a[4][2], b[8];
    for (int s = 0; s < 4; s++) for (int t = 0; t < 2; t++) a[s][t] = a0 + threadIdx.x * 1e-9 + s + 0.5 * t;
    for (int u = 0; u < 8; u++) b[u] = b0 + u;
    unsigned long long addr = threadIdx.x;
    double junk = a0;
    const int l15 = threadIdx.x & 15, l4 = (threadIdx.x >> 4) & 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (LDS) {
#pragma unroll
                for (int u = 0; u < 8; u++) b[u] = sh[((i & 1) * 16 + s * 4 + l4) * 144 + u * 16 + l15];
            }
#pragma unroll
            for (int v = 0; v < NV; v++) {           // 64-bit integer mads + f64 multiplies, independent of the MFMAs
                addr = addr * (unsigned)ld + (unsigned)(v + s);
                junk = junk * 1.0000001;
            }
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int u = 0; u < 8; u++) acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][t], b[u], acc[t][u], 0, 0, 0);
        }
        asm volatile("" : "+v"(a[0][0]), "+v"(b[0]));
    }
    double s = junk + (double)addr;
    for (int t = 0; t < 2; t++) for (int u = 0; u < 8; u++) s += acc[t][u][0] + acc[t][u][1] + acc[t][u][2] + acc[t][u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int LDS> void run()
{
    const int threads = 512, blocks = 256, iters = 2000;
    double *out; hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, LDS>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0, 1.0, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, LDS>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1.0, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)iters * 64 * (threads / 64) * blocks;
    printf("VALU pairs per 16 MFMAs = %2d, LDS operand reads = %d: %.1f cycles/MFMA/SIMD @2.4GHz (%.1f TFLOP/s)\n", NV, LDS,
           2.4e9 * (ms * 1e-3) / (mfma / (256.0 * 4)), mfma * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main()
{
    run<0, 0>(); run<8, 0>(); run<16, 0>(); run<32, 0>(); run<0, 1>(); run<16, 1>();
    return 0;
}
