// Does v_mfma_f64_16x16x4_f64 round like the sequential chain  d = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c)))) ?
// One wave, random operands with mixed signs / magnitudes, many trials; counts bitwise agreement with the
// k-ascending chain, the k-descending chain and a pairwise tree.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, const double *C, int trials, unsigned long long *cnt)
{
    const int l = threadIdx.x, l15 = l & 15, l4 = l >> 4;
    unsigned long long asc = 0, desc = 0, tree = 0, tot = 0;
    for (int t = 0; t < trials; t++) {
        const double a = A[t * 64 + l], b = B[t * 64 + l];     // A[row=l15][k=l4], B[k=l4][col=l15]
        d4 c;
        for (int r = 0; r < 4; r++) c[r] = C[(t * 4 + r) * 64 + l];   // C[row=l4+4r][col=l15]
        const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; r++) {
            const int row = l4 + 4 * r;
            double av[4], bv[4];
            for (int kk = 0; kk < 4; kk++) { av[kk] = __shfl(a, kk * 16 + row, 64); bv[kk] = __shfl(b, kk * 16 + l15, 64); }
            double s1 = c[r]; for (int kk = 0; kk < 4; kk++) s1 = __builtin_fma(av[kk], bv[kk], s1);
            double s2 = c[r]; for (int kk = 3; kk >= 0; kk--) s2 = __builtin_fma(av[kk], bv[kk], s2);
            double s3 = (av[0] * bv[0] + av[1] * bv[1]) + (av[2] * bv[2] + av[3] * bv[3]) + c[r];
            asc += d[r] == s1; desc += d[r] == s2; tree += d[r] == s3; tot++;
        }
    }
    atomicAdd(cnt + 0, asc); atomicAdd(cnt + 1, desc); atomicAdd(cnt + 2, tree); atomicAdd(cnt + 3, tot);
}
int main()
{
    const int trials = 4096;
    double *hA = (double *)malloc(sizeof(double) * trials * 64), *hB = (double *)malloc(sizeof(double) * trials * 64),
           *hC = (double *)malloc(sizeof(double) * trials * 256);
    srand(12345);
    auto rnd = [&]() { double m = (double)rand() / RAND_MAX * 2 - 1; int e = rand() % 41 - 20; return ldexp(m, e); };
    for (int i = 0; i < trials * 64; i++) { hA[i] = rnd(); hB[i] = rnd(); }
    for (int i = 0; i < trials * 256; i++) hC[i] = rnd();
    double *A, *B, *C; unsigned long long *cnt, h[4];
    hipMalloc(&A, sizeof(double) * trials * 64); hipMalloc(&B, sizeof(double) * trials * 64); hipMalloc(&C, sizeof(double) * trials * 256);
    hipMalloc(&cnt, 32); hipMemset(cnt, 0, 32);
    hipMemcpy(A, hA, sizeof(double) * trials * 64, hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof(double) * trials * 64, hipMemcpyHostToDevice);
    hipMemcpy(C, hC, sizeof(double) * trials * 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, C, trials, cnt);
    hipMemcpy(h, cnt, 32, hipMemcpyDeviceToHost);
    printf("elements %llu: equal to k-ascending fma chain %llu, k-descending chain %llu, pairwise tree %llu\n", h[3], h[0], h[1], h[2]);
    return 0;
}
