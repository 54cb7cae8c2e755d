// Does the device's FP64 division / sqrt / log / exp round like the host's (glibc), on the value ranges the fit uses?
// Strict-order mode (PAREBEN_STRICT_ORDER) can only be bit-identical to a CPU run if they do where results feed the state:
// division and sqrt do feed it (new alpha = root of a quadratic); log only feeds comparisons.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o libm_bits tools/ubench/libm_bits.hip && ./libm_bits
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__global__ void k(const double *a, const double *b, double *q, double *s, double *l, double *e, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { q[i] = a[i] / b[i]; s[i] = sqrt(a[i]); l[i] = log(a[i]); e[i] = exp(-b[i]); }
}
int main()
{
    const int n = 1 << 22;
    double *a = (double *)malloc(8 * n), *b = (double *)malloc(8 * n), *o = (double *)malloc(8 * n * 4);
    srand48(12345);
    for (int i = 0; i < n; i++) { a[i] = exp(40 * drand48() - 20); b[i] = exp(20 * drand48() - 10) * (drand48() < 0.5 ? 1 : 1 + drand48()); }
    double *da, *db, *d;
    hipMalloc(&da, 8 * n); hipMalloc(&db, 8 * n); hipMalloc(&d, 8 * n * 4);
    hipMemcpy(da, a, 8 * n, hipMemcpyHostToDevice); hipMemcpy(db, b, 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, d, d + n, d + 2 * n, d + 3 * n, n);
    hipMemcpy(o, d, 8 * n * 4, hipMemcpyDeviceToHost);
    long bad[4] = {0, 0, 0, 0}; double worst[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        volatile double r[4] = {a[i] / b[i], sqrt(a[i]), log(a[i]), exp(-b[i])};
        for (int j = 0; j < 4; j++) {
            double g = o[j * (size_t)n + i];
            if (memcmp(&g, (const void *)&r[j], 8)) { bad[j]++; double rel = fabs(g - r[j]) / fabs(r[j]); if (rel > worst[j]) worst[j] = rel; }
        }
    }
    const char *nm[4] = {"div", "sqrt", "log", "exp"};
    for (int j = 0; j < 4; j++) printf("%s: %ld of %d differ from the host's bits (worst relative difference %.3g)\n", nm[j], bad[j], n, worst[j]);
    return 0;
}
