// micro-benchmark: what one CU (a 512-thread workgroup, 8 waves) gets out of the memory system for the load shapes the fit
// kernels use, with `blocks` workgroups running (8: one per XCD, or 256: the whole chip), each on its own region.
//   shape 0: 8 B per lane, 64 consecutive lanes = one 512-byte run per load            (a coalesced row)
//   shape 1: 8 B per lane, 4 runs of 128 B (16 lanes each) `stride` bytes apart         (a 16 x 16 FP64 tile: the inverse)
//   shape 2: 16 B per lane, 64 consecutive lanes = one 1 KB run per load                (the Gram-row sweeps)
//   shape 3: 16 B per lane, 8 runs of 128 B (8 lanes each) `stride` bytes apart         (a tile, two columns per lane)
// U loads are issued back to back before the first use.  hipcc --offload-arch=gfx950 -O3 -o cu_load_rate cu_load_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int SHAPE, int U>
__global__ __launch_bounds__(512) void k(const char *base, size_t region, int stride, int iters, double *out)
{
    const char *p = base + (size_t)blockIdx.x * region;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0;
    size_t off = (size_t)wave * 65536;
    for (int it = 0; it < iters; it++) {
        double v[U]; d2 w[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t o = (off + (size_t)u * 8192 * 8) % (region - 1048576);
            if (SHAPE == 0) v[u] = *(const double *)(p + o + lane * 8);
            if (SHAPE == 1) v[u] = *(const double *)(p + o + (size_t)(lane >> 4) * stride + (lane & 15) * 8);
            if (SHAPE == 2) w[u] = *(const d2 *)(p + o + lane * 16);
            if (SHAPE == 3) w[u] = *(const d2 *)(p + o + (size_t)(lane >> 3) * stride + (lane & 7) * 16);
        }
#pragma unroll
        for (int u = 0; u < U; u++) { if (SHAPE < 2) acc += v[u]; else acc += w[u][0] + w[u][1]; }
        off += (size_t)U * 8192 * 8 + 4096;
    }
    if (acc == 1.2345) out[0] = acc;
}
template <int SHAPE, int U>
static void run(const char *buf, size_t region, int blocks, int stride, double *out)
{
    const int iters = 4000 / U;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, U>), dim3(blocks), dim3(512), 0, 0, buf, region, stride, 10, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SHAPE, U>), dim3(blocks), dim3(512), 0, 0, buf, region, stride, iters, out);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)iters * U * 8 * 64 * (SHAPE < 2 ? 8 : 16);
    printf("blocks %3d shape %d loads in flight per wave %2d: %7.1f GB/s per CU\n", blocks, SHAPE, U, bytes / (ms * 1e-3) / 1e9);
}
int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    const size_t region = (size_t)(argc > 2 ? atoi(argv[2]) : 8) << 20;       // MB per workgroup
    const int stride = argc > 3 ? atoi(argv[3]) : 8064;
    char *buf; double *out;
    hipMalloc((void **)&buf, region * blocks); hipMalloc((void **)&out, 8);
    hipMemset(buf, 0, region * blocks);
    printf("region %zu MB per workgroup, tile row stride %d B\n", region >> 20, stride);
    run<0, 2>(buf, region, blocks, stride, out); run<0, 8>(buf, region, blocks, stride, out); run<0, 16>(buf, region, blocks, stride, out);
    run<1, 2>(buf, region, blocks, stride, out); run<1, 8>(buf, region, blocks, stride, out); run<1, 16>(buf, region, blocks, stride, out);
    run<2, 2>(buf, region, blocks, stride, out); run<2, 8>(buf, region, blocks, stride, out); run<2, 16>(buf, region, blocks, stride, out);
    run<3, 2>(buf, region, blocks, stride, out); run<3, 8>(buf, region, blocks, stride, out); run<3, 16>(buf, region, blocks, stride, out);
    return 0;
}
