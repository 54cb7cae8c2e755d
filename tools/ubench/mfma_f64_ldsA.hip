// micro-benchmark: matrix ops whose A operand comes from LDS (one 512 B line per op), B held in registers;
// 16 tiles x 4 k-steps per "step", 8 waves per CU.  MODE 0: read a tile's 4 operands, then its 4 ops;
// MODE 1: operands of tile t+1 requested before the ops of tile t; MODE 2: groups of 4 tiles (16 reads, 16 ops).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, int iters, double b0, int nsteps, int sync)
{
    extern __shared__ double sh[];                      // 16 tiles x 4 x 64 doubles = 32 KB, x2 buffers
    for (int i = threadIdx.x; i < 2 * 16 * 256; i += blockDim.x) sh[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    d4 acc[16];
    for (int t = 0; t < 16; t++) acc[t] = d4{0, 0, 0, 0};
    double b[4];
    for (int s = 0; s < 4; s++) b[s] = b0 + s;
    for (int i = 0; i < iters; i++) {
        const double *A = sh + (i & 1) * (16 * 256);
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 16; t++) {
                double a[4];
#pragma unroll
                for (int s = 0; s < 4; s++) a[s] = A[(t * 4 + s) * 64 + lane];
#pragma unroll
                for (int s = 0; s < 4; s++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc[t], 0, 0, 0);
            }
        } else if (MODE == 1) {
            double ac[4], an[4];
#pragma unroll
            for (int s = 0; s < 4; s++) ac[s] = A[s * 64 + lane];
#pragma unroll
            for (int t = 0; t < 16; t++) {
                if (t + 1 < 16) {
#pragma unroll
                    for (int s = 0; s < 4; s++) an[s] = A[((t + 1) * 4 + s) * 64 + lane];
                }
#pragma unroll
                for (int s = 0; s < 4; s++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[s], b[s], acc[t], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 4; s++) { asm volatile("" : "+v"(an[s])); ac[s] = an[s]; }
            }
        } else if (MODE >= 4) {                          // mode 3 + the step's other LDS work: operand staging writes, B reads, fold
            constexpr bool FOLD_END = (MODE - 4) & 1, WRITE_FIRST = (MODE - 4) & 2, WIDE = (MODE - 4) & 4;
            typedef double d2 __attribute__((ext_vector_type(2)));
            double q = 0;
            for (int h = 0; h < nsteps; h++) {
                const double *Ah = sh + (h & 1) * (16 * 256);
                double *An = sh + ((h + 1) & 1) * (16 * 256);
                double *Bb = sh + 2 * 16 * 256;
                const int wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
                auto writes = [&]() {
#pragma unroll
                    for (int pi = 0; pi < 2; pi++) {
                        const int t = wave + 8 * pi;
                        if (t >= h + 1) {
                            if (WIDE) { for (int sp = 0; sp < 2; sp++) *(d2 *)&An[((t * 2 + sp) * 64 + lane) * 2] = d2{b0 * 1e-30 + sp, b0 * 1e-30}; }
                            else for (int s = 0; s < 4; s++) An[(t * 4 + s) * 64 + lane] = b0 * 1e-30 + s;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int e = 0; e < 2; e++) Bb[(wave + 8 * r) * 144 + e * 64 + lane + ((h + 1) & 1) * 16 * 144] = b0 * 1e-30 + e;
                };
                if (WRITE_FIRST) writes();
#pragma unroll
                for (int s = 0; s < 4; s++) b[s] = Bb[(4 * s + l4) * 144 + 16 * wave + l15 + (h & 1) * 16 * 144];
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    if (t >= h) {
                        double a[4];
                        if (WIDE) {
#pragma unroll
                            for (int sp = 0; sp < 2; sp++) { const d2 v = *(const d2 *)&Ah[((t * 2 + sp) * 64 + lane) * 2]; a[2 * sp] = v[0]; a[2 * sp + 1] = v[1]; }
                        } else {
#pragma unroll
                            for (int s = 0; s < 4; s++) a[s] = Ah[(t * 4 + s) * 64 + lane];
                        }
#pragma unroll
                        for (int s = 0; s < 4; s++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc[t], 0, 0, 0);
                        if (!FOLD_END && t == h) {
#pragma unroll
                            for (int r = 0; r < 4; r++) q += acc[t][r] * Bb[(l4 + 4 * r) * 144 + 16 * wave + l15 + (h & 1) * 16 * 144];
                            acc[t] = d4{0, 0, 0, 0};
                        }
                    }
                }
                if (!WRITE_FIRST) writes();
                if (FOLD_END) {
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        if (t == h) {
#pragma unroll
                            for (int r = 0; r < 4; r++) q += acc[t][r] * Bb[(l4 + 4 * r) * 144 + 16 * wave + l15 + (h & 1) * 16 * 144];
                            acc[t] = d4{0, 0, 0, 0};
                        }
                    }
                }
                if (sync) __syncthreads();
            }
            acc[0][0] += q;
        } else if (MODE == 3) {                          // the full-stat pass's triangular schedule: step h runs tiles t >= h
            for (int h = 0; h < nsteps; h++) {
                const double *Ah = sh + (h & 1) * (16 * 256);
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    if (t >= h) {
                        double a[4];
#pragma unroll
                        for (int s = 0; s < 4; s++) a[s] = Ah[(t * 4 + s) * 64 + lane];
#pragma unroll
                        for (int s = 0; s < 4; s++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc[t], 0, 0, 0);
                    }
                }
                if (sync) __syncthreads();
            }
        } else {
#pragma unroll
            for (int tg = 0; tg < 16; tg += 4) {
                double a[4][4];
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int s = 0; s < 4; s++) a[t][s] = A[((tg + t) * 4 + s) * 64 + lane];
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[tg + t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][s], b[s], acc[tg + t], 0, 0, 0);
            }
        }
        asm volatile("" : "+v"(b[0]));
    }
    double s = 0;
    for (int t = 0; t < 16; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int SYNC = 0> void run()
{
    const int threads = 512, blocks = 256, iters = MODE >= 3 ? 1000 : 4000;
    double *out; hipMalloc(&out, sizeof(double) * threads * blocks);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 110592);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 110592, 0, out, 10, 1.0, 16, SYNC);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 110592, 0, out, iters, 1.0, 16, SYNC);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)iters * (MODE >= 3 ? 136 * 4 : 64) * (threads / 64) * blocks;
    printf("mode %d sync %d: %.1f cycles/MFMA/SIMD @2.4GHz (%.1f TFLOP/s)\n", MODE, SYNC, 2.4e9 * (ms * 1e-3) / (mfma / (256.0 * 4)), mfma * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() { run<0>(); run<1>(); run<2>(); run<3, 0>(); run<3, 1>(); run<4, 1>(); run<8, 1>(); run<9, 1>(); return 0; }
