// blk.h -- the small set of workgroup-wide primitives the fit kernels are written against:
// strided parallel loops, barriers, deterministic reductions (fixed tree: results do not depend
// on timing, so 1/2/4/8-GPU runs are bit-identical), arg-max and ordered stream compaction.
//
// Two builds of the same source:
//   * hipcc, gfx950: 64-wide wavefront shuffles + an LDS scratch area (the product);
//   * PAREBEN_HOST_EMUL: one "thread" per block executed by g++ on the CPU.  That build exists
//     only so tests can step the kernel's control flow against the oracle without a GPU; it is
//     never linked into the shipped library and nothing falls back to it.
#pragma once
#include <stdint.h>
#include <math.h>

// Reference parameters of the fit routines (Blk, the workspace and fold descriptors, the scalars) are marked NOALIAS: a non-inlined
// routine receives them as pointers to the caller's stack, and without the no-alias promise every store through any other
// pointer forces the fields (thread id, workspace pointers ...) to be loaded again -- inside loops, each time with a full wait.
#define NOALIAS __restrict__

#ifdef PAREBEN_HOST_EMUL
#define DEV static inline
#define DEVNI static
#define BLK_LANES 1          // lane stride of the 2-d (wave x lane) loops
struct Blk {
    int tid, nthr, lane, wave, nwave;
    double *red;     // reduction scratch (unused on the host)
    int *ired;
    double *pool;    // phase-local LDS pool (device only)
    int pool_n;
};
DEV void blk_sync(const Blk &) {}
DEV double blk_sum(const Blk &, double v) { return v; }
DEV int blk_or(const Blk &, int v) { return v; }
DEV int blk_isum(const Blk &, int v) { return v; }
DEV void blk_argmax(const Blk &, double v, int idx, double *vout, int *iout) { *vout = v; *iout = idx; }
// exclusive prefix of per-thread counts in thread order; *total = block total
DEV int blk_scan_excl(const Blk &, int v, int *total) { *total = v; return 0; }
#else
#include <hip/hip_runtime.h>
#define DEV __device__ __forceinline__
#define DEVNI __device__ __noinline__   // big phases: own register allocation, no spill spill-over
#define BLK_LANES 64
#define BLK_MAX_WAVES 16
struct Blk {
    int tid, nthr, lane, wave, nwave;
    double *red;     // LDS: >= 2*BLK_MAX_WAVES doubles
    int *ired;       // LDS: 4*BLK_MAX_WAVES ints (reductions use the first 2*BLK_MAX_WAVES)
    double *pool;    // LDS: pool_n doubles, owned by whichever phase is running
    int pool_n;
};
DEV void blk_sync(const Blk &) { __syncthreads(); }

DEV double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Eight wave-wide sums for the price of ten shuffles instead of forty-eight: at the 32/16/8 strides the two
// partner lanes split the columns between them (each keeps half and adds the partner's half), the last three
// strides finish the single column a lane is left with.  On return lane l with (l & 7) == 0 holds in v[0] the
// total of column l >> 3.  The tree is fixed, so results do not depend on timing.
DEV void wave_sum8(double (&v)[8], int lane)
{
    {
        const bool up = lane & 32;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const double send = up ? v[c] : v[c + 4], keep = up ? v[c + 4] : v[c];
            v[c] = keep + __shfl_xor(send, 32, 64);
        }
    }
    {
        const bool up = lane & 16;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const double send = up ? v[c] : v[c + 2], keep = up ? v[c + 2] : v[c];
            v[c] = keep + __shfl_xor(send, 16, 64);
        }
    }
    {
        const bool up = lane & 8;
        const double send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
        v[0] = keep + __shfl_xor(send, 8, 64);
    }
    v[0] += __shfl_xor(v[0], 4, 64);
    v[0] += __shfl_xor(v[0], 2, 64);
    v[0] += __shfl_xor(v[0], 1, 64);
}
// block-wide sum, identical in every thread.  xor-butterfly inside a wave (every lane ends with
// the same bits), then the per-wave partials are added in wave order by every thread.
DEV double blk_sum(const Blk &B, double v)
{
    v = wave_sum(v);
    __syncthreads();                 // protect scratch reuse
    if (B.lane == 0) B.red[B.wave] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < B.nwave; w++) s += B.red[w];
    return s;
}
DEV int blk_or(const Blk &B, int v)
{
    return __syncthreads_or(v);
}
DEV int blk_isum(const Blk &B, int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (B.lane == 0) B.ired[B.wave] = v;
    __syncthreads();
    int s = 0;
    for (int w = 0; w < B.nwave; w++) s += B.ired[w];
    return s;
}
// max value, lowest index among equal maxima; result identical in every thread
DEV void blk_argmax(const Blk &B, double v, int idx, double *vout, int *iout)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(v, o, 64);
        int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    __syncthreads();
    if (B.lane == 0) { B.red[B.wave] = v; B.ired[B.wave] = idx; }
    __syncthreads();
    double bv = B.red[0]; int bi = B.ired[0];
    for (int w = 1; w < B.nwave; w++) {
        double ov = B.red[w]; int oi = B.ired[w];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    *vout = bv; *iout = bi;
}
DEV int blk_scan_excl(const Blk &B, int v, int *total)
{
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (B.lane >= o) incl += t;
    }
    __syncthreads();
    if (B.lane == 63) B.ired[B.wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < B.nwave; w++) {
        int c = B.ired[w];
        if (w < B.wave) base += c;
        tot += c;
    }
    *total = tot;
    return base + incl - v;
}
#endif

#define PAR(i, n) for (int i = B.tid; i < (n); i += B.nthr)
