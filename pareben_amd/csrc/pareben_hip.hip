// pareben_hip.hip -- kernels and C ABI of libpareben_hip.so (gfx950 only).
//
// Data flow of one pareben_ctx_run():
//   BASIS, Target, fold ids (resident in HBM since ctx_create)
//     -> split_kernel      per fold: training / held-out rows compacted, column-major
//     -> colstats_kernel   per fold, per column: |x|, 1/|x|, x.y/|x|, x.1/|x|
//     -> ystats_kernel     per fold: mean and unbiased variance of the training target
//     -> gram_kernel       per fold: normalised Gram matrix G (K x K), LDS-tiled FP64
//     -> gm_cv_kernel      persistent workgroups pull (cell, fold) units, heaviest first, from
//                          an atomic queue; one workgroup = one fit + its held-out score
//     -> fold_err / status / counters copied back
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <string>
#include <thread>
#include <mutex>
#include <dlfcn.h>
#include <rccl/rccl.h>           // types and prototypes only: the library is bound at run time (rccl_load)

#ifndef FIT_THREADS
#define FIT_THREADS 512          // 8 wavefronts per fit workgroup: 256 VGPRs per lane for the matrix-core pass
#endif
#define FS_NWAVES (FIT_THREADS / 64)
#ifndef FIT_WAVES_PER_EU
#define FIT_WAVES_PER_EU 2       // wavefronts per SIMD the fit kernels are compiled for (register budget 512 / this)
#endif

#include "../../include/pareben_hip.h"
#include "types.h"
#include "blk.h"
#include "gm_fit.h"
#include "bm_fit.h"
#include "gm_strict.h"

// ------------------------------------------------------------------------------------------
// error plumbing
static thread_local std::string g_err;
static int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(PAREBEN_EHIP, #x, e_); } while (0)

extern "C" const char *pareben_version(void) { return "pareben-hip 0.1 (gfx950)"; }
extern "C" const char *pareben_last_error(void) { return g_err.c_str(); }
extern "C" int pareben_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// preparation kernels

// rows[] lists the source rows of this fold's training (or held-out) set; one workgroup per column
__global__ void split_kernel(const double *__restrict__ basis, int n, const int *__restrict__ rows,
                             int nr, double *__restrict__ out)
{
    const int j = blockIdx.x;
    const double *src = basis + (size_t)j * n;
    double *dst = out + (size_t)j * nr;
    for (int r = threadIdx.x; r < nr; r += blockDim.x) dst[r] = src[rows[r]];
}

// pairwise (epistasis) columns of one fold: Z[:, K + rank(i,j)] = x_i * x_j for i < j, in the
// reference's order (1,2),(1,3),..,(1,K),(2,3).. (elasticNetLinearNeFull2.c:115-134); the first K
// columns of Z are the main effects (copied by split_kernel).  One workgroup per pair column.
__global__ void expand_kernel(const double *__restrict__ X, int N, int K, double *__restrict__ Z)
{
    const long long q = blockIdx.x;                       // pair rank
    // invert rank -> (i, j): rank(i, j) = i*K - i*(i+1)/2 + (j - i - 1)
    int i = (int)((2.0 * K - 1.0 - sqrt((2.0 * K - 1.0) * (2.0 * K - 1.0) - 8.0 * (double)q)) * 0.5);
    while ((long long)(i + 1) * K - (long long)(i + 1) * (i + 2) / 2 <= q) i++;
    while ((long long)i * K - (long long)i * (i + 1) / 2 > q) i--;
    const int j = (int)(q - ((long long)i * K - (long long)i * (i + 1) / 2)) + i + 1;
    const double *xi = X + (size_t)i * N, *xj = X + (size_t)j * N;
    double *z = Z + ((size_t)K + q) * N;
    for (int h = threadIdx.x; h < N; h += blockDim.x) z[h] = xi[h] * xj[h];
}

__global__ void gather_kernel(const double *__restrict__ y, const int *__restrict__ rows, int nr,
                              double *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nr) out[r] = y[rows[r]];
}

__device__ __forceinline__ double block_sum_256(double v, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += sh[w];
    return s;
}

// per column: scale = |x| (1 when 0), rscale, bt0 = x.y/scale, cs = x.1/scale
// (elasticNetLinearNeMainEff.c:87-99 and the y- and 1-parts of :1171-1177)
// phi (optional): the normalised column, the u-operand of gram_kernel, formed as the reference forms its PHI: a main-effect
// column is scaled by the reciprocal of its norm (dscal with 1/Scales, :517-520, :1062-1065; Full2.c:530-535), a pair column is
// divided by it (Full2.c:544, :913)
__global__ void colstats_kernel(const double *__restrict__ X, const double *__restrict__ y, int N,
                                double *__restrict__ scale, double *__restrict__ rscale,
                                double *__restrict__ bt0, double *__restrict__ cs, double *__restrict__ phi, int n_main)
{
    __shared__ double sh[16];
    const int j = blockIdx.x;
    const double *x = X + (size_t)j * N;
    double q = 0, xy = 0, x1 = 0;
    for (int h = threadIdx.x; h < N; h += blockDim.x) {
        const double v = x[h];
        q += v * v; xy += v * y[h]; x1 += v;
    }
    q = block_sum_256(q, sh);
    xy = block_sum_256(xy, sh);
    x1 = block_sum_256(x1, sh);
    if (threadIdx.x == 0) {
        if (q == 0) q = 1;
        const double s = sqrt(q);
        scale[j] = s; rscale[j] = 1 / s; bt0[j] = xy / s; cs[j] = x1 / s;
    }
    if (phi) {
        const double s = sqrt(q == 0 ? 1.0 : q), r = 1 / s;     // q is the same in every thread (block_sum_256)
        if (j < n_main) for (int h = threadIdx.x; h < N; h += blockDim.x) phi[(size_t)j * N + h] = x[h] * r;
        else for (int h = threadIdx.x; h < N; h += blockDim.x) phi[(size_t)j * N + h] = x[h] / s;
    }
}

// out[0] = sum(y)/N, out[1] = unbiased variance (:145-152, :1826-1838)
__global__ void ystats_kernel(const double *__restrict__ y, int N, double *__restrict__ out)
{
    __shared__ double sh[16];
    double s = 0;
    for (int h = threadIdx.x; h < N; h += blockDim.x) s += y[h];
    s = block_sum_256(s, sh);
    const double m = s / N;
    double v = 0;
    for (int h = threadIdx.x; h < N; h += blockDim.x) { const double d = y[h] - m; v += d * d; }
    v = block_sum_256(v, sh);
    if (threadIdx.x == 0) { out[0] = m; out[1] = v / (N - 1); }
}

// Normalised Gram matrix of one fold: G[u*K + i] = ( sum_h X[h,i] * (X[h,u] / scale[u]) ) / scale[i], i.e. the
// reference's BASIS_PHI row of basis u (PHI = x_u/|x_u| dotted with every column, then divided by that
// column's norm: elasticNetLinearNeMainEff.c:1171-1177, :1608-1630) for every u at once.
//   * Same operations in the same order as the reference's sequential dot product: the u-operand is the column
//     divided by its norm (colstats_kernel writes that copy, Phi: a division in this kernel's staging loop took a
//     quarter of the matrix pipe's time -- FP64 vector ops take turns with it), and a 16 x 16 accumulator tile of v_mfma_f64_16x16x4_f64 is ONE fma chain
//     over the samples h in ascending order (the matrix op rounds like four chained fmas,
//     tools/ubench/mfma_f64_order.hip).  For integer-coded designs (genotypes: -1/0/1) every product is exact,
//     so G equals the reference's values bit for bit -- which is what keeps the long add/delete trajectories of
//     the real-R table on the reference's path ("Parity on chaotic fits").  The price: x_u/|x_u| on
//     one side only makes the two triangles round differently, so all K^2 entries are computed (a
//     symmetric-half version was 2x faster and lost that parity; profiles/r02).
//   * FP64 matrix cores: a 128 x 128 block per 256-thread workgroup, 64 x 64 per wave = 4 x 4 accumulator
//     tiles (A rows <-> u, B columns <-> i, k <-> sample h).
//   * Operands: 16-sample slabs of the two 128-column panels, loaded with 16 lanes along h (columns are
//     contiguous in h: whole 128-byte lines), staged in LDS as [column][h] with pitch 17 and double
//     buffered: the loads of slab s+1 are in flight while slab s feeds the matrix cores; one barrier per slab.
//     Per slab and workgroup: 32 KB loaded for 0.5 MFLOP -> far from any memory bound once the panels of
//     concurrently running blocks come out of L2.
//   * XCD-aware block order: workgroup b runs on XCD b % 8, which is given a contiguous range of the
//     row-major block list, so the 32 CUs of an XCD work on consecutive blocks of one block row at any time
//     (one shared u-panel, neighbouring i-panels) and their panels meet in that XCD's L2.
#define GB 128
#define GS 16
#define GP 17
typedef double gd4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void gram_kernel(const double *__restrict__ X, const double *__restrict__ Phi, int N, int K,
                                                      const double *__restrict__ scale,
                                                      double *__restrict__ G, int nb, int n_blocks)
{
    __shared__ double sm[2][2][GB * GP];
    const int per = (n_blocks + 7) / 8;
    const int lin = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (lin >= n_blocks) return;                                  // uniform per workgroup, before any barrier
    const int bu = lin / nb, bi = lin - bu * nb;
    const int u0 = bu * GB, i0 = bi * GB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int wu = wave >> 1, wi = wave & 1;
    // staging: thread -> sample h = tid & 15 of columns (tid >> 4) + 16 q
    const int sh = tid & 15, sc = tid >> 4;
    gd4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = gd4{0, 0, 0, 0};
    double ra[8], rb[8];
    auto fetch = [&](int h0) {
        const int hh = h0 + sh;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int cu = u0 + sc + 16 * q, ci = i0 + sc + 16 * q;
            ra[q] = (hh < N && cu < K) ? Phi[(size_t)cu * N + hh] : 0.0;
            rb[q] = (hh < N && ci < K) ? X[(size_t)ci * N + hh] : 0.0;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            sm[buf][0][(sc + 16 * q) * GP + sh] = ra[q];             // PHI = x_u / |x_u|, formed by colstats_kernel
            sm[buf][1][(sc + 16 * q) * GP + sh] = rb[q];
        }
    };
    const int n_slab = (N + GS - 1) / GS;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int s = 0; s < n_slab; s++) {
        const int buf = s & 1;
        if (s + 1 < n_slab) fetch((s + 1) * GS);
        const double *pa = &sm[buf][0][(wu * 64 + l15) * GP + l4];
        const double *pb = &sm[buf][1][(wi * 64 + l15) * GP + l4];
#pragma unroll
        for (int ks = 0; ks < GS / 4; ks++) {
            double av[4], bv[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { av[t] = pa[t * 16 * GP + ks * 4]; bv[t] = pb[t * 16 * GP + ks * 4]; }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (s + 1 < n_slab) stash(buf ^ 1);
        __syncthreads();
    }
    // rows u, 16 consecutive i per 16 lanes: 128-byte row segments
    double sci[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int i = i0 + wi * 64 + b * 16 + l15;
        sci[b] = i < K ? scale[i] : 1.0;
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int u = u0 + wu * 64 + a * 16 + l4 + 4 * r;
            if (u >= K) continue;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int i = i0 + wi * 64 + b * 16 + l15;
                if (i < K) G[(size_t)u * K + i] = acc[a][b][r] / sci[b];
            }
        }
}

// ------------------------------------------------------------------------------------------
// workspace carving

struct WsLayout { size_t bytes; int cap, ld; size_t offK, offSig, offM, offX; int nmax; };   // offX: the strict-order mode's extra arrays (0 = none)

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static WsLayout ws_layout(int K, int cap, int strict_nmax = 0)
{
    WsLayout L;
    L.cap = cap; L.ld = (cap + 15) / 16 * 16;      // whole 16-row blocks: the matrix-core passes index rows without clamps
    size_t o = 0;
    L.offK = o;   o += align_up((size_t)K * (7 * sizeof(double) + 2 * sizeof(int) + 1), 256);
    L.offSig = o; o += align_up(((size_t)3 * L.ld * L.ld + (size_t)L.ld * INV_TP) * sizeof(double), 256);      // Sigma, H, Gram block cache, inverse panel
    L.offM = o;   o += align_up((size_t)(cap + 2) * ((7 + ADD_TB) * sizeof(double) + 3 * sizeof(int)) + 4 * ADD_TB * sizeof(double), 256);
    L.offX = 0; L.nmax = 0;
    if (strict_nmax > 0) {                          // gm_strict.h: t, e, phi (max(N, cap + 2) each), w1, w2 (cap + 2 each), PHI'PHI (ld x ld), BASIS_PHI (cap x K)
        L.nmax = std::max(strict_nmax, cap + 2);
        L.offX = o;
        o += align_up(((size_t)3 * L.nmax + 2 * (size_t)(cap + 2) + (size_t)L.ld * L.ld + (size_t)cap * K) * sizeof(double), 256);
    }
    L.bytes = align_up(o, 4096);
    return L;
}

__device__ inline GsExtra gs_carve(char *base, const GmWork &W, size_t offX, int nmax, int K)
{
    GsExtra X;
    double *d = (double *)(base + offX);
    X.t = d; d += nmax; X.e = d; d += nmax; X.phi = d; d += nmax;
    X.w1 = d; d += W.cap + 2; X.w2 = d; d += W.cap + 2;
    X.D = d; d += (size_t)W.ld * W.ld;
    X.BP = d;
    X.SigNew = W.Gc;
    (void)K;
    return X;
}

__device__ inline GmWork ws_carve(char *base, int K, int cap, size_t offK, size_t offSig, size_t offM)
{
    GmWork W;
    double *d = (double *)(base + offK);
    W.Sin = d; d += K; W.Qin = d; d += K; W.Sout = d; d += K; W.Qout = d; d += K;
    W.dml = d; d += K; W.aroot = d; d += K; W.bt = d; d += K;
    int *ip = (int *)d;
    W.upos = ip; ip += K; W.todo = ip; ip += K;
    W.act = (signed char *)ip;
    d = (double *)(base + offSig);
    const int ldp = (cap + 15) / 16 * 16;
    W.Sig = d; d += (size_t)ldp * ldp; W.H = d; d += (size_t)ldp * ldp; W.Gc = d; d += (size_t)ldp * ldp; W.Tn = d;
    d = (double *)(base + offM);
    const int c1 = cap + 1;
    W.A = d; d += c1; W.mu = d; d += c1; W.gam = d; d += c1;
    W.v1 = d; d += c1; W.v2 = d; d += c1; W.v3 = d; d += c1; W.v4 = d; d += c1;
    W.vb = d; d += (size_t)ADD_TB * (cap + 2); W.bsc = d; d += 4 * ADD_TB;
    W.used = (int *)d;
    W.rowid = W.used + c1;
    W.pfree = W.rowid + c1;
    W.priv_base = 0; W.priv_rows = 0;
    W.e = nullptr;
    W.cap = cap; W.ld = ldp; W.cap_flag = cap;
    return W;
}

// binomial workspace: K-vectors, (ld x ld) Sigma and H, ld-vectors, N-vectors, K x ld weighted rows
struct BmLayout { size_t bytes; int cap, ld, nmax; size_t offK, offSig, offM, offN, offBP; };

static BmLayout bm_layout(int K, int nmax, int bmax)
{
    BmLayout L;
    int cap = std::min(K, bmax) + 1; if (cap > 1024) cap = 1024;      // model columns incl. the intercept
    L.cap = cap; L.ld = cap + 1; L.nmax = nmax;
    size_t o = 0;
    L.offK = o;   o += align_up((size_t)K * (7 * sizeof(double) + 2 * sizeof(int) + 1), 256);
    L.offSig = o; o += align_up((size_t)2 * L.ld * L.ld * sizeof(double), 256);
    L.offM = o;   o += align_up((size_t)L.ld * (9 * sizeof(double) + sizeof(int)), 256);
    L.offN = o;   o += align_up((size_t)nmax * 5 * sizeof(double), 256);
    L.offBP = o;  o += align_up((size_t)K * L.ld * sizeof(double), 256);
    L.bytes = align_up(o, 4096);
    return L;
}

__device__ inline BmWork bm_carve(char *base, int K, const BmLayout &L)
{
    BmWork W;
    double *d = (double *)(base + L.offK);
    W.Sin = d; d += K; W.Qin = d; d += K; W.Sout = d; d += K; W.Qout = d; d += K;
    W.dml = d; d += K; W.aroot = d; d += K; W.bb = d; d += K;
    int *ip = (int *)d;
    W.upos = ip; ip += K; W.todo = ip; ip += K;
    W.act = (signed char *)ip;
    d = (double *)(base + L.offSig);
    W.Sig = d; d += (size_t)L.ld * L.ld; W.H = d;
    d = (double *)(base + L.offM);
    W.A = d; d += L.ld; W.mu = d; d += L.ld; W.g = d; d += L.ld; W.dmu = d; d += L.ld; W.mnew = d; d += L.ld;
    W.tmp = d; d += L.ld; W.tp = d; d += L.ld; W.v3 = d; d += L.ld; W.v4 = d; d += L.ld;
    W.used = (int *)d;
    d = (double *)(base + L.offN);
    W.w = d; d += L.nmax; W.pm = d; d += L.nmax; W.yv = d; d += L.nmax; W.e = d; d += L.nmax; W.bphi = d;
    W.BP = (double *)(base + L.offBP);
    W.cap = L.cap; W.ld = L.ld; W.phi_div = 0; W.bmax = L.cap;
    return W;
}

// ------------------------------------------------------------------------------------------
// fit kernels


struct CvParams {
    const FoldDev *folds;
    const double *alpha, *lambda;     // per cell
    const int *order;                 // unit ids (cell * n_folds + fold), heaviest first
    int *queue;                       // head of the work queue
    double *fold_err;
    int *status;
    long long *counters;              // may be null
    long long *phase;                 // [n_units x PH_N] diagnostic ticks, may be null
    char *ws;
    size_t ws_stride, offK, offSig, offM;
    int K, cap, cap_flag, n_folds, n_units;
    int priv_base0, priv_rows;         // lazy Gram mode: private rows of workgroup b start at priv_base0 + b * priv_rows
    FsJob *jobs;                       // shared phases (gm_fit.h); null = off
    int *active;
    int early;                         // share from the start (few fits per workgroup)
    int heavy_m;                       // active-set size from which a fit shares its phases from the start
    int defer;                         // hold back the sweep of a block's last unit (gm_inner); PAREBEN_DEFER=0: off
    int inv_pair;                      // forms of the blocked inverse (gm_dev.h): bit 0 two pivot blocks per trip, bit 1 register-resident; PAREBEN_INV_PAIR=<bits>, default 3
    int queue_sys;                     // the queue head is shared by several GPUs (pinned host memory, pareben_cv_grid_multi)
    size_t offX; int nmax;             // strict-order mode (gm_cv_strict_kernel): the extra arrays of gm_strict.h
    GmVariant v;
};

// LDS carve of one fit workgroup (dynamic shared memory): a phase-local pool (full-stat Gram
// k-blocks and Sigma panels / inverse panels / action vectors) and the small reduction scratch.
#ifndef LDS_POOL_DOUBLES
#define LDS_POOL_DOUBLES 19456    // 152 KB of the CU's 160 KB: the blocked inverse keeps its M x 16 panel (pitch 18) in it up to M = 1040 (beyond: GmWork::Tn)
#endif
#define LDS_FIT_BYTES ((LDS_POOL_DOUBLES + 2 * BLK_MAX_WAVES) * 8 + 4 * BLK_MAX_WAVES * 4)
#define BM_POOL_DOUBLES_HALF 9856  // two binomial workgroups per CU: 77 KB pool + scratch + static LDS each, under half of 160 KB
extern __shared__ double lds_dyn[];

#define LDS_BYTES_FOR(pool) (((pool) + 2 * BLK_MAX_WAVES) * 8 + 4 * BLK_MAX_WAVES * 4)

__device__ inline Blk make_blk(int pool_n = LDS_POOL_DOUBLES)
{
    Blk B;
    B.tid = threadIdx.x; B.nthr = blockDim.x;
    B.lane = threadIdx.x & 63;
    B.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    B.nwave = blockDim.x >> 6;
    B.pool = lds_dyn; B.pool_n = pool_n;
    B.red = lds_dyn + pool_n;
    B.ired = (int *)(B.red + 2 * BLK_MAX_WAVES);
    return B;
}

__device__ inline void store_counters(long long *dst, const FitCounters &c)
{
    dst[0] = c.n_outer; dst[1] = c.n_inner; dst[2] = c.n_add; dst[3] = c.n_del; dst[4] = c.n_reest;
    dst[5] = c.n_fullstat; dst[6] = c.sum_m_action; dst[7] = c.sum_m_full; dst[8] = c.sum_m2_full;
    dst[9] = c.m_final; dst[10] = c.m_max; dst[11] = c.status; dst[12] = c.mfma_tiles; dst[13] = c.sum_m_swept;
}

// A workgroup that found the work queue empty helps the fits still running: it scans the job board (64
// owners per load with wave 0), claims a chunk of feature tiles of an open full-stat pass and runs it on
// the owner's state (gm_fit.h, "shared full-stat passes").  Leaves when no workgroup owns a fit any more.
// `until_all_done`: the tail of the launch (this workgroup owns nothing any more: stay until no workgroup
// owns a fit).  Otherwise: between two fits, help as long as some open job has chunks left, then return.
__device__ void fs_help_loop(const Blk &B, const FsShare &sh, int K, bool until_all_done)
{
    __shared__ int s_pick[4];
    if (!sh.jobs) return;
    __syncthreads();
    for (;;) {
        __syncthreads();
        if (B.wave == 0) {
            int owner = -1, first = -1;
            // leave when nobody owns a fit and none is left to pull.  Leaving early is always safe: an owner only
            // ever waits for chunks that a running helper has claimed, and takes the unclaimed ones itself -- so a
            // launch that is not fully resident (workgroups that start late, or never) cannot stall anybody.
            const int quit = until_all_done && AT_LOAD(sh.queue) >= sh.n_units && AT_LOAD(sh.active) <= 0;
            for (int base = 0; !quit && owner < 0 && base < sh.n_blocks; base += 64) {
                const int b = base + B.lane;
                unsigned long long w = b < sh.n_blocks ? AT_LOAD(&sh.jobs[b].word) : 0ull;
                const int nt = (b < sh.n_blocks && fs_epoch_open(w)) ? AT_LOAD(&sh.jobs[b].n_tiles) : 0;
                const int chunk = (b < sh.n_blocks && fs_epoch_open(w)) ? job_chunk(AT_LOAD(&sh.jobs[b].kind), AT_LOAD(&sh.jobs[b].M)) : FS_CHUNK;
                unsigned long long cand = __ballot(fs_epoch_open(w) && (int)(unsigned)w < nt);
                while (cand && owner < 0) {
                    const int l = __ffsll((long long)cand) - 1;
                    cand &= cand - 1;
                    int got = -1;
                    if (B.lane == l) {
                        unsigned long long e = w;
                        if (__hip_atomic_compare_exchange_strong(&sh.jobs[b].word, &e, w + chunk, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT)) got = (int)(unsigned)w;
                    }
                    got = __shfl(got, l, 64);
                    if (got >= 0) { owner = base + l; first = got; }
                }
            }
            if (B.lane == 0) { s_pick[0] = owner; s_pick[1] = first; s_pick[2] = quit; }
        }
        __syncthreads();
        const int owner = s_pick[0], first = s_pick[1], quit = s_pick[2];
        if (quit) break;
        if (owner < 0) {
            if (!until_all_done) break;
            __builtin_amdgcn_s_sleep(127);
            continue;
        }
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // the owner's Sigma, mu, row ids, bt
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        FsJob *job = sh.jobs + owner;
        const int M = AT_LOAD(&job->M), fold = AT_LOAD(&job->fold), n_tiles = AT_LOAD(&job->n_tiles);
        const int kind = AT_LOAD(&job->kind), mode = AT_LOAD(&job->mode), rid = AT_LOAD(&job->rid), aux = AT_LOAD(&job->pad);
        const double beta = __hip_atomic_load(&job->beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double c1 = __hip_atomic_load(&job->c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double c2 = __hip_atomic_load(&job->c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const GmWork Wo = ws_carve(sh.ws + (size_t)owner * sh.ws_stride, K, sh.cap, sh.offK, sh.offSig, sh.offM);
        const FoldDev Fo = sh.folds[fold];
        const int last = first + job_chunk(kind, M) < n_tiles ? first + job_chunk(kind, M) : n_tiles;
        if (kind == JOB_SQB) {
            const int Kp = K & ~1;
            gm_sq_batch_range(B, Fo, Wo, K, M, mode, beta, first * SQ_FT, last * SQ_FT < Kp ? last * SQ_FT : Kp);
        } else if (kind == JOB_SQ)
            gm_sq_tiles(B, Fo, Wo, K, M, Wo.v2, mode, beta, c1, c2, (mode == 1 && rid >= 0) ? Fo.G + (size_t)rid * K : nullptr, first, last, true,
                        mode == 2 ? aux : -1, mode == 2 ? rid : -1);     // a held-back delete: the freed slot and its Gram row (gm_sq_stage)
        else
            gm_fullstat_features(B, Fo, Wo, K, M, beta, first, last);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every wave drains its S_in / Q_in stores
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            AT_ADD(&job->done, 1);
        }
    }
}

__global__ __launch_bounds__(FIT_THREADS, FIT_WAVES_PER_EU) void gm_cv_kernel(CvParams P)
{
    __shared__ int s_unit;
    __shared__ FitCounters s_cnt;
    __shared__ long long s_ph[PH_N];
    const Blk B = make_blk();
    GmWork W = ws_carve(P.ws + (size_t)blockIdx.x * P.ws_stride, P.K, P.cap, P.offK, P.offSig, P.offM);
    W.priv_rows = P.priv_rows;
    W.priv_base = P.priv_base0 + (int)blockIdx.x * P.priv_rows;
    W.cap_flag = P.cap_flag;
    FsShare sh;
    sh.jobs = P.jobs; sh.active = P.active; sh.queue = P.queue; sh.n_units = P.n_units; sh.n_blocks = gridDim.x;
    sh.self = blockIdx.x; sh.ws = P.ws; sh.ws_stride = P.ws_stride; sh.offK = P.offK; sh.offSig = P.offSig; sh.offM = P.offM;
    sh.folds = P.folds; sh.cap = P.cap; sh.early = P.early; sh.heavy_m = P.heavy_m;
    for (int n_done = 0;; n_done++) {
        if (n_done > 0) fs_help_loop(B, sh, P.K, false);            // between fits: lend a hand to the long ones
        __syncthreads();
        // one queue head per launch in HBM, or one for all GPUs of a multi-GPU call in pinned host memory (system scope)
        if (threadIdx.x == 0) s_unit = P.queue_sys ? __hip_atomic_fetch_add(P.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : atomicAdd(P.queue, 1);
        __syncthreads();
        const int q = s_unit;
        if (q >= P.n_units) break;                 // every wave of every workgroup reaches this
        if (threadIdx.x == 0 && P.active) AT_ADD(P.active, 1);      // this workgroup owns a fit (see fs_help_loop)
        const int unit = P.order[q];
        const int cell = unit / P.n_folds, f = unit % P.n_folds;
        const FoldDev F = P.folds[f];
        GmScalars S;
        S.c = &s_cnt;
        S.ph = s_ph;
        S.v = P.v;
        S.share = &sh;
        S.fold = f;
        S.defer = P.defer; S.inv_pair = P.inv_pair;
#ifdef PAREBEN_PHASE_TIMERS
        if (threadIdx.x < PH_N) s_ph[threadIdx.x] = 0;
        __syncthreads();
        const long long t_fit0 = wall_clock64();
#endif
        gm_fit(B, F, W, P.K, P.lambda[cell], P.alpha[cell], S);
        const double sse = gm_fold_sse(B, F, W, S);
#ifdef PAREBEN_PHASE_TIMERS
        if (threadIdx.x == 0 && P.phase) {
            s_ph[PH_TOTAL] = wall_clock64() - t_fit0;
            for (int k = 0; k < PH_N; k++) P.phase[(size_t)unit * PH_N + k] = s_ph[k];
        }
#endif
        if (threadIdx.x == 0) {
            P.fold_err[unit] = (S.status & ST_ABORT) ? __builtin_nan("") : sse;   // an aborted fit has no score
            P.status[unit] = S.status;
            if (P.counters) store_counters(P.counters + (size_t)unit * PAREBEN_NCOUNTERS, s_cnt);
            if (P.active) AT_ADD(P.active, -1);
        }
    }
    fs_help_loop(B, sh, P.K, true);
}

struct BmCvParams {
    const FoldDev *folds;
    const double *alpha, *lambda;
    const int *order;
    int *queue;
    double *fold_err;
    int *status;
    long long *counters;
    char *ws;
    BmLayout L;
    int K, n_folds, n_units;
    int epis, bmax;        // epistasis: NeFull.c rule set on the expanded design, at most bmax bases per model
    long long *phase;      // [n_units x PH_N] diagnostic ticks, may be null
    int pool_n;            // doubles of the dynamic LDS pool this launch was given (two 256-thread workgroups per CU: half the CU's LDS each)
    int queue_sys;         // see CvParams
};

#ifndef BM_WAVES_PER_EU
#define BM_WAVES_PER_EU FIT_WAVES_PER_EU
#endif
__global__ __launch_bounds__(FIT_THREADS, BM_WAVES_PER_EU) void bm_cv_kernel(BmCvParams P)
{
    __shared__ int s_unit;
    __shared__ FitCounters s_cnt;
    __shared__ long long s_ph[PH_N];
    const Blk B = make_blk(P.pool_n);
    BmWork W = bm_carve(P.ws + (size_t)blockIdx.x * P.L.bytes, P.K, P.L);
    W.phi_div = P.epis; W.bmax = P.bmax;
    for (;;) {
        __syncthreads();
        // one queue head per launch in HBM, or one for all GPUs of a multi-GPU call in pinned host memory (system scope)
        if (threadIdx.x == 0) s_unit = P.queue_sys ? __hip_atomic_fetch_add(P.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : atomicAdd(P.queue, 1);
        __syncthreads();
        const int q = s_unit;
        if (q >= P.n_units) break;
        const int unit = P.order[q];
        const int cell = unit / P.n_folds, f = unit % P.n_folds;
        const FoldDev F = P.folds[f];
        GmScalars S;
        S.c = &s_cnt; S.ph = s_ph;
        S.v.epis = P.epis;
#ifdef PAREBEN_PHASE_TIMERS
        if (threadIdx.x < PH_N) s_ph[threadIdx.x] = 0;
        __syncthreads();
        const long long t_fit0 = wall_clock64();
#endif
        double ll;
        bm_fit(B, F, W, P.K, P.lambda[cell], P.alpha[cell], S, &ll);
        const double score = bm_fold_loglik(B, F, W, S);
#ifdef PAREBEN_PHASE_TIMERS
        if (threadIdx.x == 0 && P.phase) {
            s_ph[PH_TOTAL] = wall_clock64() - t_fit0;
            for (int k = 0; k < PH_N; k++) P.phase[(size_t)unit * PH_N + k] = s_ph[k];
        }
#endif
        if (threadIdx.x == 0) {
            P.fold_err[unit] = (S.status & ST_ABORT) ? __builtin_nan("") : score;
            P.status[unit] = S.status;
            if (P.counters) store_counters(P.counters + (size_t)unit * PAREBEN_NCOUNTERS, s_cnt);
        }
    }
}

struct FitParams {
    FoldDev F;
    double lambda, alpha;
    double *Beta;        // K x 4 (main effects) or K x 5 (epistasis, K = p(p+1)/2)
    double *scalars;     // wald, intercept, residual
    int *status;
    long long *counters;
    char *ws;
    size_t offK, offSig, offM;
    int K, cap, cap_flag;
    int p;               // design columns (K = p without epistasis)
    GmVariant v;
    // helper workgroups (blocks 1..): the job board of the shared phases; null = the fit runs alone
    FsJob *jobs; int *active; int *queue; const FoldDev *folds; size_t ws_stride;
    unsigned long long *trace; long long trace_cap;    // decision trace (pareben_set_trace); null = off
    double *outer_log;                                 // verbose > 2: 3 doubles per outer iteration; null = off
    int defer, inv_pair;
    size_t offX; int nmax;                             // strict-order mode (gm_fit_strict_kernel)
};

// single Gaussian fit with the reference's .C outputs: elasticNetLinearNeMainEff.c:199-227 (Beta K x 4:
// locus, locus, effect, posterior variance) or elasticNetLinearNeFull2.c:115-134, :232-238 (Beta
// M_full x 5: locus1, locus2, effect, variance, 1-based column id of used columns)
__global__ __launch_bounds__(FIT_THREADS, FIT_WAVES_PER_EU) void gm_fit_kernel(FitParams P)
{
    __shared__ FitCounters s_cnt;
    __shared__ long long s_ph[PH_N];
    const Blk B = make_blk();
    // One fit = one workgroup, as in the CV kernel; the other workgroups of the launch only lend a hand with its
    // feature-parallel phases (full-stat passes, action sweeps) through the same job board and leave when the
    // owner is done (`active`, set to 1 by the host, drops to 0).  Bit-identical with or without them.
    FsShare sh;
    sh.jobs = P.jobs; sh.active = P.active; sh.queue = P.queue; sh.n_units = 1; sh.n_blocks = gridDim.x;
    sh.self = blockIdx.x; sh.ws = P.ws; sh.ws_stride = P.ws_stride; sh.offK = P.offK; sh.offSig = P.offSig; sh.offM = P.offM;
    sh.folds = P.folds; sh.cap = P.cap; sh.early = 1; sh.heavy_m = 0;
    if (blockIdx.x > 0) { fs_help_loop(B, sh, P.K, true); return; }
    GmWork W = ws_carve(P.ws, P.K, P.cap, P.offK, P.offSig, P.offM);
    W.cap_flag = P.cap_flag;
    const int K = P.K, p = P.p;
    const int ncol = P.v.epis ? 5 : 4;
    PAR(i, p) { P.Beta[i] = i + 1; P.Beta[(size_t)K + i] = i + 1; }
    if (P.v.epis) {
        PAR(i, p - 1) {                                       // pairs (i, j > i) follow the main effects
            size_t kk = (size_t)p + (size_t)i * (2 * (size_t)p - i - 1) / 2;
            for (int j = i + 1; j < p; j++, kk++) { P.Beta[kk] = i + 1; P.Beta[(size_t)K + kk] = j + 1; }
        }
    }
    for (int c = 2; c < ncol; c++) PAR(i, K) P.Beta[(size_t)c * K + i] = 0;
    GmScalars S;
    S.c = &s_cnt;
    S.ph = s_ph;
    S.v = P.v;
    if (P.jobs) { S.share = &sh; S.fold = 0; }
    S.trace = P.trace; S.trace_cap = P.trace_cap; S.outer_log = P.outer_log; S.defer = P.defer; S.inv_pair = P.inv_pair;
    gm_fit(B, P.F, W, K, P.lambda, P.alpha, S);
    if (threadIdx.x == 0 && P.active) AT_ADD(P.active, -1);        // the helpers may go (every path of the owner gets here)
    const int M = S.M, ld = W.ld;
    PAR(i, M) {
        const int f = W.used[i];
        const double sc = P.F.scale[f];
        P.Beta[2 * (size_t)K + f] = W.mu[i] / sc;
        P.Beta[3 * (size_t)K + f] = W.Sig[(size_t)i * ld + i] / (sc * sc);
        if (P.v.epis) P.Beta[4 * (size_t)K + f] = f + 1;
    }
    // Wald score mu' H mu with the H of the last final update (:199-215); H is held as its triangle [j][i >= j]
    double part = 0;
    PAR(i, M) {
        double a = 0;
        for (int j = 0; j < M; j++) a += W.mu[j] * W.H[i < j ? (size_t)i * ld + j : (size_t)j * ld + i];
        part += a * W.mu[i];
    }
    const double wald = blk_sum(B, part);
    if (threadIdx.x == 0) {
        P.scalars[0] = wald;
        P.scalars[1] = S.b;
        P.scalars[2] = 1 / (S.beta + 1e-10);
        P.status[0] = S.status;
        if (P.counters) store_counters(P.counters, s_cnt);
    }
}

// ---- PAREBEN_STRICT_ORDER=1 (diagnostic): the same launches with gm_strict.h's fit -- the reference's formulation and
// operation order, bit-identical to the netlib-order CPU oracle -- instead of the production fit.  One workgroup per fit,
// no shared phases.
__global__ __launch_bounds__(FIT_THREADS, FIT_WAVES_PER_EU) void gm_cv_strict_kernel(CvParams P)
{
    __shared__ int s_unit;
    __shared__ FitCounters s_cnt;
    const Blk B = make_blk();
    char *base = P.ws + (size_t)blockIdx.x * P.ws_stride;
    GmWork W = ws_carve(base, P.K, P.cap, P.offK, P.offSig, P.offM);
    W.cap_flag = P.cap_flag;
    const GsExtra X = gs_carve(base, W, P.offX, P.nmax, P.K);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_unit = P.queue_sys ? __hip_atomic_fetch_add(P.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : atomicAdd(P.queue, 1);
        __syncthreads();
        const int q = s_unit;
        if (q >= P.n_units) break;
        const int unit = P.order[q];
        const int cell = unit / P.n_folds, f = unit % P.n_folds;
        const FoldDev F = P.folds[f];
        GmScalars S;
        S.c = &s_cnt; S.ph = nullptr; S.v = P.v;
        gs_fit(B, F, W, X, P.K, P.lambda[cell], P.alpha[cell], S);
        const double sse = gs_fold_sse(B, F, W, X, S, P.K);
        if (threadIdx.x == 0) {
            P.fold_err[unit] = (S.status & ST_ABORT) ? __builtin_nan("") : sse;
            P.status[unit] = S.status;
            if (P.counters) store_counters(P.counters + (size_t)unit * PAREBEN_NCOUNTERS, s_cnt);
        }
    }
}

__global__ __launch_bounds__(FIT_THREADS, FIT_WAVES_PER_EU) void gm_fit_strict_kernel(FitParams P)
{
    __shared__ FitCounters s_cnt;
    const Blk B = make_blk();
    GmWork W = ws_carve(P.ws, P.K, P.cap, P.offK, P.offSig, P.offM);
    W.cap_flag = P.cap_flag;
    const GsExtra X = gs_carve(P.ws, W, P.offX, P.nmax, P.K);
    const int K = P.K;
    PAR(i, K) { P.Beta[i] = i + 1; P.Beta[(size_t)K + i] = i + 1; P.Beta[2 * (size_t)K + i] = 0; P.Beta[3 * (size_t)K + i] = 0; }
    GmScalars S;
    S.c = &s_cnt; S.ph = nullptr; S.v = P.v;
    S.trace = P.trace; S.trace_cap = P.trace_cap; S.outer_log = P.outer_log;
#ifdef GS_TIMING
    __shared__ long long s_tim[8];
    if (threadIdx.x < 8) s_tim[threadIdx.x] = 0;
    __syncthreads();
    const_cast<GsExtra &>(X).tim = s_tim;
    const long long t_all0 = wall_clock64();
#endif
    gs_fit(B, P.F, W, X, K, P.lambda, P.alpha, S);
#ifdef GS_TIMING
    if (threadIdx.x == 0) { s_cnt.mfma_tiles = s_tim[0]; s_cnt.sum_m_swept = s_tim[1]; s_cnt.sum_m_action = s_tim[2]; s_cnt.sum_m_full = s_tim[3];
                            s_cnt.sum_m2_full = s_tim[4]; s_cnt.n_fullstat = wall_clock64() - t_all0;
                            s_cnt.n_add = s_tim[5]; s_cnt.n_del = s_tim[6]; s_cnt.n_reest = s_tim[7]; }
    __syncthreads();
#endif
    const int M = S.M, ld = W.ld;
    __syncthreads();
    PAR(i, M) {
        const int f = W.used[i];
        const double sc = P.F.scale[f];
        P.Beta[2 * (size_t)K + f] = W.mu[i] / sc;
        P.Beta[3 * (size_t)K + f] = W.Sig[(size_t)i * ld + i] / (sc * sc);
    }
    if (threadIdx.x == 0) {                                       // Wald score mu' H mu (:199-215), ddot order
        double wald = 0;
        for (int i = 0; i < M; i++) {
            double a = 0;
            for (int j = 0; j < M; j++) a = a + W.mu[j] * W.H[(size_t)i * ld + j];
            wald = wald + a * W.mu[i];
        }
        P.scalars[0] = wald;
        P.scalars[1] = S.b;
        P.scalars[2] = 1 / (S.beta + 1e-10);
        P.status[0] = S.status;
        if (P.counters) store_counters(P.counters, s_cnt);
    }
}

// sample-major copy of a fold's training design for the strict-order mode: Xt[h*K + i] = X[i*N + h] (32 x 32 tiles through LDS)
__global__ void transpose_kernel(const double *__restrict__ X, int N, int K, double *__restrict__ Xt)
{
    __shared__ double tile[32][33];
    const int i0 = blockIdx.x * 32, h0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) { const int i = i0 + r, h = h0 + tx; tile[r][tx] = (i < K && h < N) ? X[(size_t)i * N + h] : 0.0; }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) { const int h = h0 + r, i = i0 + tx; if (h < N && i < K) Xt[(size_t)h * K + i] = tile[tx][r]; }
}

// column norms in the reference's sequential order (:87-99) for the strict-order mode: scale = sqrt(sum x^2) (1 when 0), 1/scale
__global__ void strict_scale_kernel(const double *__restrict__ X, int N, int K, double *__restrict__ scale, double *__restrict__ rscale)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= K) return;
    const double *x = X + (size_t)j * N;
    double q = 0;
    for (int h = 0; h < N; h++) q = __dadd_rn(q, __dmul_rn(x[h], x[h]));
    if (q == 0) q = 1;
    const double sc = sqrt(q);
    scale[j] = sc; rscale[j] = 1 / sc;
}

struct BmFitParams {
    FoldDev F;
    double lambda, alpha;
    double *Beta;        // K x 4 (main effects) or bmax x 4 (epistasis: the used bases in model order)
    double *scalars;     // logLikelihood, wald, intercept, Sigma[0,0]
    int *status;
    long long *counters;
    char *ws;
    BmLayout L;
    int K;
    int epis, p, bmax;   // epistasis: K = p(p+1)/2 columns of the expanded design
    double *outer_log;   // verbose > 2: 3 doubles per outer iteration; null = off
};

// single binomial fit with the reference's .C outputs: ElasticNetBinaryNEmainEff.c:346-389 (Beta K x 4 indexed by
// column) or ElasticNetBinaryNeFull.c:154-211 (Beta bMax x 4: the used bases in model order with their loci)
__global__ __launch_bounds__(FIT_THREADS, FIT_WAVES_PER_EU) void bm_fit_kernel(BmFitParams P)
{
    __shared__ FitCounters s_cnt;
    __shared__ long long s_ph[PH_N];
    const Blk B = make_blk();
    BmWork W = bm_carve(P.ws, P.K, P.L);
    W.phi_div = P.epis; W.bmax = P.bmax;
    const int K = P.K;
    if (!P.epis) { PAR(i, K) { P.Beta[i] = i + 1; P.Beta[(size_t)K + i] = i + 1; P.Beta[2 * (size_t)K + i] = 0; P.Beta[3 * (size_t)K + i] = 0; } }
    else { PAR(i, 4 * P.bmax) P.Beta[i] = 0; }
    GmScalars S;
    S.c = &s_cnt; S.ph = s_ph;
    S.v.epis = P.epis;
    S.outer_log = P.outer_log;
    double ll;
    bm_fit(B, P.F, W, K, P.lambda, P.alpha, S, &ll);
    const int M = S.M, ld = W.ld;
    if (!P.epis) {
        for (int i = 1 + threadIdx.x; i < M; i += blockDim.x) {
            const int f = W.used[i - 1];
            const double sc = P.F.scale[f];
            P.Beta[2 * (size_t)K + f] = W.mu[i] / sc;
            P.Beta[3 * (size_t)K + f] = W.Sig[(size_t)i * ld + i] / (sc * sc);
        }
    } else {
        const int p = P.p, bm = P.bmax;
        const int meff = M - 1 < bm ? M - 1 : bm;                  // NeFull.c:162-167
        PAR(i, meff) {
            const int f = W.used[i];
            int l1 = f, l2 = f;
            if (f >= p) {                                          // pair rank -> (i, j), the order of NeFull.c:90-105
                const long long q = f - p;
                int a = (int)((2.0 * p - 1.0 - sqrt((2.0 * p - 1.0) * (2.0 * p - 1.0) - 8.0 * (double)q)) * 0.5);
                while ((long long)(a + 1) * p - (long long)(a + 1) * (a + 2) / 2 <= q) a++;
                while ((long long)a * p - (long long)a * (a + 1) / 2 > q) a--;
                l1 = a; l2 = (int)(q - ((long long)a * p - (long long)a * (a + 1) / 2)) + a + 1;
            }
            const double sc = P.F.scale[f];
            P.Beta[i] = l1 + 1; P.Beta[(size_t)bm + i] = l2 + 1;
            P.Beta[2 * (size_t)bm + i] = W.mu[i + 1] / sc;
            P.Beta[3 * (size_t)bm + i] = W.Sig[(size_t)(i + 1) * ld + i + 1] / (sc * sc);
        }
    }
    double part = 0;
    PAR(i, M) {
        double a = 0;
        for (int j = 0; j < M; j++) a += W.H[(size_t)i * ld + j] * W.mu[j];
        part += a * W.mu[i];
    }
    const double wald = blk_sum(B, part);
    if (threadIdx.x == 0) {
        P.scalars[0] = ll;
        P.scalars[1] = wald;
        P.scalars[2] = W.mu[0];
        P.scalars[3] = W.Sig[0];
        P.status[0] = S.status;
        if (P.counters) store_counters(P.counters, s_cnt);
    }
}

// ------------------------------------------------------------------------------------------
// host side

struct FoldHost {
    int N = 0, nte = 0;
    int *d_tr = nullptr, *d_te = nullptr;
    double *X = nullptr, *y = nullptr, *Xte = nullptr, *yte = nullptr;
    double *scale = nullptr, *rscale = nullptr, *bt0 = nullptr, *cs = nullptr, *G = nullptr;
    double *ystat = nullptr;
    double *Xt = nullptr;      // strict-order mode: sample-major copy of X
};

struct pareben_ctx {
    int device = 0, n = 0, p = 0, n_folds = 0, prior = 0, epis = 0;
    int cap = 0;            // active-set capacity of the fit workspaces
    int cap_ref = 0;        // the reference's basisMax (<= cap): fits whose active set grows past it are flagged
    int kfull = 0;          // columns the fit sees: p, or p(p+1)/2 with epistasis
    GmVariant variant{};
    double *d_basis = nullptr, *d_y = nullptr;
    double *d_phi = nullptr;   // max(N_train) x K scratch: the normalised design of the fold whose Gram matrix is being built
    std::vector<FoldHost> folds;
    FoldDev *d_folds = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    char *d_ws = nullptr; size_t ws_bytes = 0; int ws_blocks = 0;
    WsLayout L{};
    BmLayout BL{};
    double last_ms[3] = {0, 0, 0};
    int64_t launch_info[5] = {0, 0, 0, 0, 0};
    int n_cu = 0;
    int strict = 0;         // PAREBEN_STRICT_ORDER=1 at context creation: gm_strict.h's fit (Gaussian main effects only)
    // on-demand Gram rows (K x K per fold does not fit): [lazy_hdr ints of pool counters][n_folds x K slots]
    int lazy = 0, pool_rows = 0, priv_rows = 0, max_blocks = 0, lazy_hdr = 0;
    int *d_lazy = nullptr;
    double *d_rows = nullptr;   // lazy mode: n_folds pools of pool_rows rows, then max_blocks x priv_rows private rows
};

// Active-set capacities.  The reference sizes its per-fit arrays for basisMax = min(K, 1e7/K) columns
// (elasticNetLinearNeMainEff.c:68-69; elasticNetLinearNeFull2.c:67-80 for epistasis) and, when a fit grows
// past that, prints "out of Memory" and keeps writing (:605-611) -- undefined behaviour, at K = 50 000 already
// from 200 columns on.  Policy here: `cap_ref` is that number and only FLAGS a fit (PAREBEN_ST_OVERFLOW);
// the fit goes on in a workspace of `cap` >= cap_ref columns, cap = max(cap_ref, min(N_train, 2048)) bounded
// by K and by what the matrix-core passes are laid out for (FS_MAX_M), and is stopped (ST_OVERFLOW | ST_ABORT,
// score NaN) only there.  An active set larger than the number of training rows is not a model the algorithm
// keeps (delete priority from M >= N on), so min(N_train, .) costs nothing.  The workspace is 3 x cap^2 doubles per
// resident fit (100 MB at 2048 columns, 26 GB for 256 workgroups: the Gram planner in ctx_create counts it).
// Real data needs the room: on the reference's own 3802 x 19 871 design every fit of one fold passes 1024 columns
// within seven inner iterations (batch adds) and peaks at 1067 ... 1451 before the deletes bring it back to ~200.
// max_active > 0 (ctx_create) lowers both; PAREBEN_WS_CAP=<cols> lowers the workspace bound and PAREBEN_REF_CAP=<n>
// cap_ref alone (tests: the stop and the flag-and-continue paths at sizes the oracle can follow).
static void capacities(int K, int ref_rule, int n_train_max, int max_active, int *cap_ref_out, int *cap_out)
{
    long ref = ref_rule;
    if (ref > K) ref = K;
    long ws = FS_MAX_M;
    if (const char *e = getenv("PAREBEN_WS_CAP")) { const long v = atol(e); if (v >= 2 && v < ws) ws = v; }
    long cap = std::max<long>(ref, std::min<long>(n_train_max, ws));
    if (cap > K) cap = K;
    if (cap > FS_MAX_M) cap = FS_MAX_M;
    if (max_active > 0 && cap > max_active) cap = max_active;
    if (cap < 2) cap = 2;
    if (const char *e = getenv("PAREBEN_REF_CAP")) { const long v = atol(e); if (v > 0 && v < ref) ref = v; }
    if (ref > cap) ref = cap;
    if (ref < 1) ref = 1;
    *cap_ref_out = (int)ref; *cap_out = (int)cap;
}

template <class T> static hipError_t dmalloc(T **p, size_t count)
{
    return hipMalloc((void **)p, count ? count * sizeof(T) : sizeof(T));
}

extern "C" int pareben_ctx_destroy(pareben_ctx *c)
{
    if (!c) return PAREBEN_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &f : c->folds) {
        hipFree(f.d_tr); hipFree(f.d_te); hipFree(f.X); hipFree(f.y); hipFree(f.Xte); hipFree(f.yte);
        hipFree(f.scale); hipFree(f.rscale); hipFree(f.bt0); hipFree(f.cs); hipFree(f.G); hipFree(f.ystat); hipFree(f.Xt);
    }
    hipFree(c->d_basis); hipFree(c->d_y); hipFree(c->d_phi); hipFree(c->d_folds); hipFree(c->d_ws); hipFree(c->d_lazy); hipFree(c->d_rows);
    for (auto &e : c->ev) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return PAREBEN_OK;
}

// rows_tr[f] / rows_te[f]: source rows of fold f's training / held-out set
static int ctx_create_impl(pareben_ctx **out, int device, const double *basis, int n, int p,
                           const double *target, const std::vector<std::vector<int>> &rows_tr,
                           const std::vector<std::vector<int>> &rows_te, int prior, int epis, int max_active)
{
    const int n_folds = (int)rows_tr.size();
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(PAREBEN_EINVAL, "no such device");
    HIPCHK(hipSetDevice(device));
    pareben_ctx *c = new pareben_ctx();
    c->device = device; c->n = n; c->p = p; c->n_folds = n_folds; c->prior = prior; c->epis = epis;
    c->kfull = epis ? (int)((long long)p * (p + 1) / 2) : p;
    { const char *so = getenv("PAREBEN_STRICT_ORDER"); c->strict = (so && atoi(so) != 0 && prior == PAREBEN_PRIOR_GAUSSIAN && !epis) ? 1 : 0; }
    if (epis && (long long)p * (p + 1) / 2 > 2000000000LL) { delete c; return fail(PAREBEN_EINVAL, "too many pairwise columns"); }
    int n_train_max = 2;
    for (auto &tr : rows_tr) n_train_max = std::max(n_train_max, (int)tr.size());
    if (!epis) {
        capacities(c->kfull, (int)std::min<double>(1e7 / p, 2e9), n_train_max, max_active, &c->cap_ref, &c->cap);
        c->variant = GmVariant{0, 0.9, 0.001, 1e-3, 1e2, 1e-10};
    } else {                                       // basisMax of elasticNetLinearNeFull2.c:67-80
        int ref = 0;
        for (auto &tr : rows_tr) {
            const int N = (int)tr.size();
            ref = std::max(ref, N > p ? 2 * p : (N < 200 ? 4 * p : p));
        }
        capacities(c->kfull, ref, n_train_max, max_active, &c->cap_ref, &c->cap);
        c->variant = GmVariant{1, 0.99, 0.01, 0.1, 1e3, 0.0};
    }
    const size_t KF = (size_t)c->kfull;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { pareben_ctx_destroy(c); return fail(e_ == hipErrorOutOfMemory ? PAREBEN_ENOMEM : PAREBEN_EHIP, #x, e_); } } while (0)
    CK(hipStreamCreate(&c->stream));
    for (auto &e : c->ev) CK(hipEventCreate(&e));
    CK(dmalloc(&c->d_basis, (size_t)n * p));
    CK(dmalloc(&c->d_y, (size_t)n));
    CK(hipMemcpy(c->d_basis, basis, sizeof(double) * (size_t)n * p, hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_y, target, sizeof(double) * n, hipMemcpyHostToDevice));
    c->folds.resize(n_folds);
    std::vector<FoldDev> fd(n_folds);
    for (int f = 0; f < n_folds; f++) {
        FoldHost &H = c->folds[f];
        const std::vector<int> &tr = rows_tr[f], &te = rows_te[f];
        H.N = (int)tr.size(); H.nte = (int)te.size();
        if (H.N < 2) { pareben_ctx_destroy(c); return fail(PAREBEN_EINVAL, "a fold leaves fewer than 2 training rows"); }
        CK(dmalloc(&H.d_tr, tr.size())); CK(dmalloc(&H.d_te, te.size()));
        CK(hipMemcpy(H.d_tr, tr.data(), sizeof(int) * tr.size(), hipMemcpyHostToDevice));
        if (!te.empty()) CK(hipMemcpy(H.d_te, te.data(), sizeof(int) * te.size(), hipMemcpyHostToDevice));
        CK(dmalloc(&H.X, (size_t)H.N * KF)); CK(dmalloc(&H.y, (size_t)H.N));
        CK(dmalloc(&H.Xte, (size_t)H.nte * KF)); CK(dmalloc(&H.yte, (size_t)H.nte));
        CK(dmalloc(&H.scale, KF)); CK(dmalloc(&H.rscale, KF));
        CK(dmalloc(&H.bt0, KF)); CK(dmalloc(&H.cs, KF));
        CK(dmalloc(&H.ystat, (size_t)2));
        if (c->strict || prior == PAREBEN_PRIOR_BINOMIAL) CK(dmalloc(&H.Xt, (size_t)H.N * KF));   // sample-major copy: strict-order mode; the binomial weighted-rows pass (bm_dev.h)
    }
    if (prior == PAREBEN_PRIOR_GAUSSIAN) {         // binomial: no Gram matrix (the weights change)
        // Full per-fold Gram matrices when they fit beside the fit workspaces; else one buffer of
        // Gram rows filled on demand by the fit kernel (gm_row): a shared pool per fold plus a few
        // private rows per workgroup for when a pool runs out.  PAREBEN_GRAM_ROWS=<rows per fold>
        // forces the second mode (diagnostics / tests).
        size_t free_b = 0, total_b = 0;
        CK(hipMemGetInfo(&free_b, &total_b));
        int occ = 1;
        CK(hipFuncSetAttribute((const void *)gm_cv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FIT_BYTES));
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gm_cv_kernel, FIT_THREADS, LDS_FIT_BYTES));
        c->max_blocks = c->n_cu * std::max(occ, 1);
        const size_t ws_guess = (size_t)c->max_blocks * ws_layout(c->kfull, c->cap).bytes + ((size_t)1 << 30);
        const size_t avail = (free_b > ws_guess ? free_b - ws_guess : 0) / 10 * 9;
        const size_t row_b = KF * sizeof(double);
        const char *force = getenv("PAREBEN_GRAM_ROWS");
        const long long forced = force ? atoll(force) : 0;
        const size_t phi_b = (size_t)n_train_max * row_b;      // scratch of gram_kernel's normalised operand
        if (forced > 0 || (size_t)n_folds * KF * row_b + phi_b > avail) {
            c->lazy = 1;
            long long priv = std::min<long long>(c->cap, (long long)(avail / 4 / ((size_t)c->max_blocks * row_b)));
            long long rows = (long long)((avail - (size_t)priv * c->max_blocks * row_b) / ((size_t)n_folds * row_b));
            if (forced > 0) { rows = forced; priv = c->cap; }
            rows = std::min<long long>(rows, (long long)KF);
            if ((rows + priv < 1) || (rows * n_folds + priv * c->max_blocks) > 2000000000LL) {
                pareben_ctx_destroy(c);
                return fail(PAREBEN_ENOMEM, "no room for Gram rows");
            }
            c->pool_rows = (int)rows; c->priv_rows = (int)priv;
            CK(dmalloc(&c->d_rows, ((size_t)rows * n_folds + (size_t)priv * c->max_blocks) * KF));
            c->lazy_hdr = (n_folds + 63) / 64 * 64;
            CK(dmalloc(&c->d_lazy, (size_t)c->lazy_hdr + (size_t)n_folds * KF));
        } else {
            for (int f = 0; f < n_folds; f++) CK(dmalloc(&c->folds[f].G, KF * KF));
            CK(dmalloc(&c->d_phi, (size_t)n_train_max * KF));
        }
    }
    for (int f = 0; f < n_folds; f++) {
        FoldHost &H = c->folds[f];
        FoldDev &D = fd[f];
        D.X = H.X; D.y = H.y; D.Xte = H.Xte; D.yte = H.yte; D.scale = H.scale; D.rscale = H.rscale;
        D.bt0 = H.bt0; D.cs = H.cs; D.G = c->lazy ? c->d_rows : H.G; D.ymean = 0; D.varY = 0; D.N = H.N; D.nte = H.nte;
        D.lazy = c->lazy; D.pool_rows = c->pool_rows; D.pool_base = f * c->pool_rows; D.n_main = p; D.Xt = H.Xt;
        D.slot_of = c->lazy ? c->d_lazy + c->lazy_hdr + (size_t)f * KF : nullptr;
        D.pool_next = c->lazy ? c->d_lazy + f : nullptr;
    }
    CK(dmalloc(&c->d_folds, (size_t)n_folds));
    CK(hipMemcpy(c->d_folds, fd.data(), sizeof(FoldDev) * n_folds, hipMemcpyHostToDevice));
#undef CK
    *out = c;
    return PAREBEN_OK;
}

extern "C" int pareben_ctx_create(pareben_ctx **out, int device, const double *basis, int n, int p,
                                  const double *target, const int32_t *fold_id, int n_folds,
                                  int prior, int epis, int max_active)
{
    if (!out || !basis || !target || !fold_id || n < 2 || p < 1 || n_folds < 1) return fail(PAREBEN_EINVAL, "bad argument");
    if (prior != PAREBEN_PRIOR_GAUSSIAN && prior != PAREBEN_PRIOR_BINOMIAL) return fail(PAREBEN_EINVAL, "unknown prior");
    for (int i = 0; i < n; i++) if (fold_id[i] < 1 || fold_id[i] > n_folds) return fail(PAREBEN_EINVAL, "fold_id out of 1..n_folds");
    std::vector<std::vector<int>> tr(n_folds), te(n_folds);
    for (int f = 0; f < n_folds; f++)
        for (int i = 0; i < n; i++) (fold_id[i] == f + 1 ? te[f] : tr[f]).push_back(i);
    return ctx_create_impl(out, device, basis, n, p, target, tr, te, prior, epis, max_active);
}

// launch the per-fold preparation on the context's stream and patch ymean/varY into d_folds
static int prepare_folds(pareben_ctx *c)
{
    const int p = c->p, n = c->n, kf = c->kfull;
    for (int f = 0; f < c->n_folds; f++) {
        FoldHost &H = c->folds[f];
        hipLaunchKernelGGL(split_kernel, dim3(p), dim3(256), 0, c->stream, c->d_basis, n, H.d_tr, H.N, H.X);
        if (H.nte) hipLaunchKernelGGL(split_kernel, dim3(p), dim3(256), 0, c->stream, c->d_basis, n, H.d_te, H.nte, H.Xte);
        if (kf > p) {                              // pairwise columns behind the p main-effect columns
            hipLaunchKernelGGL(expand_kernel, dim3(kf - p), dim3(256), 0, c->stream, H.X, H.N, p, H.X);
            if (H.nte) hipLaunchKernelGGL(expand_kernel, dim3(kf - p), dim3(256), 0, c->stream, H.Xte, H.nte, p, H.Xte);
        }
        hipLaunchKernelGGL(gather_kernel, dim3((H.N + 255) / 256), dim3(256), 0, c->stream, c->d_y, H.d_tr, H.N, H.y);
        if (H.nte) hipLaunchKernelGGL(gather_kernel, dim3((H.nte + 255) / 256), dim3(256), 0, c->stream, c->d_y, H.d_te, H.nte, H.yte);
        hipLaunchKernelGGL(colstats_kernel, dim3(kf), dim3(256), 0, c->stream, H.X, H.y, H.N, H.scale, H.rscale, H.bt0, H.cs,
                           H.G ? c->d_phi : (double *)nullptr, p);
        if (H.Xt)                                    // the design sample-major (gm_strict.h; bm_weighted_rows)
            hipLaunchKernelGGL(transpose_kernel, dim3((kf + 31) / 32, (H.N + 31) / 32), dim3(256), 0, c->stream, H.X, H.N, kf, H.Xt);
        if (c->strict)
            hipLaunchKernelGGL(strict_scale_kernel, dim3((kf + 255) / 256), dim3(256), 0, c->stream, H.X, H.N, kf, H.scale, H.rscale);
        hipLaunchKernelGGL(ystats_kernel, dim3(1), dim3(256), 0, c->stream, H.y, H.N, H.ystat);
        if (H.G) {                                  // 128 x 128 blocks, dealt to the XCDs in contiguous ranges of the row-major list
            const int nb = (kf + GB - 1) / GB;
            const long long nbl = (long long)nb * nb;
            const int per = (int)((nbl + 7) / 8);
            hipLaunchKernelGGL(gram_kernel, dim3(per * 8), dim3(256), 0, c->stream, H.X, c->d_phi, H.N, kf, H.scale, H.G, nb, (int)nbl);
        }
        // ymean / varY live inside the FoldDev record: copy the two doubles device-to-device
        HIPCHK(hipMemcpyAsync((char *)(c->d_folds + f) + offsetof(FoldDev, ymean), H.ystat, 2 * sizeof(double),
                              hipMemcpyDeviceToDevice, c->stream));
    }
    if (c->lazy) {                                 // empty row pools: every run recomputes what it needs
        HIPCHK(hipMemsetAsync(c->d_lazy, 0, sizeof(int) * c->lazy_hdr, c->stream));
        HIPCHK(hipMemsetAsync(c->d_lazy + c->lazy_hdr, 0xFF, sizeof(int) * (size_t)c->n_folds * kf, c->stream));
    }
    HIPCHK(hipGetLastError());
    return PAREBEN_OK;
}

static int ensure_workspace(pareben_ctx *c, int blocks)
{
    int strict_nmax = 0;
    if (c->strict) for (auto &f : c->folds) strict_nmax = std::max(strict_nmax, std::max(f.N, f.nte));
    c->L = ws_layout(c->kfull, c->cap, strict_nmax);
    size_t per = c->L.bytes;
    if (c->prior == PAREBEN_PRIOR_BINOMIAL) {
        int nmax = 1;
        for (auto &f : c->folds) nmax = std::max(nmax, std::max(f.N, f.nte));
        c->BL = bm_layout(c->kfull, nmax, c->epis ? 2 * c->p : c->kfull);     // NeFull.c: the R wrapper's bMax = 2K bases
        per = c->BL.bytes;
    }
    const size_t need = per * (size_t)blocks;
    if (need > c->ws_bytes) {
        if (c->d_ws) { hipFree(c->d_ws); c->d_ws = nullptr; c->ws_bytes = 0; }
        hipError_t e = hipMalloc((void **)&c->d_ws, need);
        if (e != hipSuccess) return fail(PAREBEN_ENOMEM, "workspace hipMalloc", e);
        // Every fit initialises what it reads (the matrix-core passes mask the ragged 16-blocks beyond the
        // active set), so the fill value is irrelevant to the results; zero for reproducible memory dumps.
        // PAREBEN_WS_POISON=1 fills with 0xFF bytes (NaN doubles, -1 ints) instead:
        // tests/test_hip_parity.py::test_poisoned_workspace_is_bit_identical proves that independence.
        const char *poison = getenv("PAREBEN_WS_POISON");
        e = hipMemset(c->d_ws, (poison && atoi(poison)) ? 0xFF : 0, need);
        if (e != hipSuccess) return fail(PAREBEN_EHIP, "workspace hipMemset", e);
        c->ws_bytes = need;
    }
    c->ws_blocks = blocks;
    return PAREBEN_OK;
}

// Device-side state of one launch: everything run_enqueue() puts on the context's stream; the results stay in
// HBM until the caller copies them (pareben_ctx_run) or hands them to the all-gather (pareben_cv_grid_multi).
struct RunDev {
    double *d_alpha = nullptr, *d_lambda = nullptr, *d_err = nullptr;
    int *d_order = nullptr, *d_queue = nullptr, *d_status = nullptr;
    long long *d_cnt = nullptr, *d_phase = nullptr;
    FsJob *d_jobs = nullptr; int *d_active = nullptr;
    std::vector<int> order;                  // host copies the async H2D transfers read: alive until the stream is synchronised
    int act_host[4] = {0, 0, 0, 0};
    int n_units = 0, blocks = 0;
    void release()
    {
        hipFree(d_jobs); hipFree(d_active); hipFree(d_phase); hipFree(d_alpha); hipFree(d_lambda); hipFree(d_err);
        hipFree(d_order); hipFree(d_queue); hipFree(d_status); hipFree(d_cnt);
        d_jobs = nullptr; d_active = nullptr; d_phase = nullptr; d_alpha = d_lambda = d_err = nullptr;
        d_order = d_queue = d_status = nullptr; d_cnt = nullptr;
    }
};

// Enqueue one grid evaluation on the context's stream: parameter upload, per-fold preparation, the persistent fit
// kernel.  Events ev[0..2] bracket preparation and fit.  Nothing is synchronised here.
// shared_queue: null = the launch has its own queue head; else the pinned-host head all GPUs of a multi-GPU call pull
// from (zeroed by the caller) and n_sharers = how many launches share it
static int run_enqueue(pareben_ctx *c, int n_cells, const double *alpha, const double *lambda, bool want_counters, RunDev &D,
                       int *shared_queue = nullptr, int n_sharers = 1)
{
    HIPCHK(hipSetDevice(c->device));
    const int nF = c->n_folds, n_units = n_cells * nF;
    // Queue order.  Cost is far from monotone in lambda: nothing happens above the lambda where the first
    // features enter (a third of the grid costs nothing), the heaviest fits -- seconds each, against a mean of
    // 0.2 s -- sit right below it at the sparse-to-dense transition (its position moves with alpha), and from there
    // down the cost is flat.  So the queue simply runs from the large-lambda end: the free cells fly by, the heavy
    // ridge starts within the first milliseconds, and the uniform plateau packs the tail.  (Earlier builds took
    // one cell in four from the small-lambda end instead; measured on rank shares of config 2 the plain
    // descending order is as fast on a full grid -- 7.98 vs 7.99 s -- and better when a launch holds few fits per
    // workgroup, where a late heavy fit is the critical path: 4.40 -> 4.20 s for a half grid, 2.88 -> 2.67 s for a
    // quarter.  PAREBEN_QUEUE_MIX=<n> brings the interleave back for A/B runs.)
    std::vector<int> sorted(n_cells), cells(n_cells);
    std::iota(sorted.begin(), sorted.end(), 0);
    std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) {
        if (lambda[a] != lambda[b]) return lambda[a] > lambda[b];
        return alpha[a] < alpha[b];
    });
    // binomial fits: cost grows steadily towards small lambda (larger models, more Newton steps), so the queue simply runs
    // from the small-lambda end -- longest first, the short ones pack the tail (config 3: 8 % shorter than outside-in)
    const char *qenv = getenv("PAREBEN_QUEUE");                 // A/B: "outside-in" | "small-first"
    const bool small_first = qenv ? !strcmp(qenv, "small-first") : c->prior == PAREBEN_PRIOR_BINOMIAL;
    const char *menv = getenv("PAREBEN_QUEUE_MIX");             // A/B: one cell from the small-lambda end per <n> (default: none)
    const int qmix = (menv && atoi(menv) >= 2) ? atoi(menv) : 0;       // 0: none from the small end
    for (int k = 0, lo = 0, hi = n_cells - 1; k < n_cells; k++)
        cells[k] = small_first ? sorted[n_cells - 1 - k] : ((qmix && (k % qmix) == qmix - 1) ? sorted[hi--] : sorted[lo++]);
    D.order.resize(n_units);
    for (int k = 0; k < n_cells; k++) for (int f = 0; f < nF; f++) D.order[k * nF + f] = cells[k] * nF + f;

    int occ = 1;
    const bool binom = c->prior == PAREBEN_PRIOR_BINOMIAL;
    // Binomial launch shape.  Two fits per CU as 256-thread workgroups with half the LDS pool each when the launch has
    // fits enough for that and a fold's samples allow the matrix-core paths in half a pool (N <= 512); else one 512-thread
    // workgroup per CU with the whole pool.  A binomial fit is a chain of short phases in which three or four of the eight
    // waves work while the rest stand at a barrier (63 % of the wave cycles wait: profiles/r03/pmc_extra_config3.txt); a second
    // resident fit fills those gaps: config 3 205 -> 192 ms per grid.  (Measured early in round 3, when every phase still
    // paid a memory round trip per element, the same shape was slower: 282 vs 273 ms.)  PAREBEN_BM_THREADS=512 | 256 forces a
    // shape; PAREBEN_BM_POOL=<doubles> sets the pool (a build with -DBM_WAVES_PER_EU=4 needs two 512-thread workgroups to fit
    // in one CU's LDS).  The reductions of a fit run over its own threads, so the last bits of a binomial result depend on the
    // shape (not on anything else: same shape, same bits).
    int bm_threads = 512, bm_pool = LDS_POOL_DOUBLES;
    if (binom && n_units >= 2 * c->n_cu && c->BL.nmax <= 512) { bm_threads = 256; bm_pool = BM_POOL_DOUBLES_HALF; }
    if (const char *e = getenv("PAREBEN_BM_THREADS")) {
        if (atoi(e) == 256) { bm_threads = 256; bm_pool = BM_POOL_DOUBLES_HALF; }
        else if (atoi(e) == 512) { bm_threads = 512; bm_pool = LDS_POOL_DOUBLES; }
    }
    if (const char *e = getenv("PAREBEN_BM_POOL")) { const int v = atoi(e); if (v >= 4096 && v <= LDS_POOL_DOUBLES) bm_pool = v; }
    if (binom) {
        HIPCHK(hipFuncSetAttribute((const void *)bm_cv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_FOR(bm_pool)));
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bm_cv_kernel, bm_threads, LDS_BYTES_FOR(bm_pool)));
    } else {
        HIPCHK(hipFuncSetAttribute((const void *)gm_cv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FIT_BYTES));
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gm_cv_kernel, FIT_THREADS, LDS_FIT_BYTES));
    }
    if (occ < 1) occ = 1;
    int blocks = std::min(n_units, c->n_cu * occ);
    if (c->lazy && blocks > c->max_blocks) blocks = c->max_blocks;
    int rc = ensure_workspace(c, blocks);
    if (rc) return rc;
    D.n_units = n_units; D.blocks = blocks;

    const char *phase_path = getenv("PAREBEN_PHASE_DUMP");     // diagnostic build only
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { D.release(); return fail(PAREBEN_EHIP, #x, e_); } } while (0)
    CK(dmalloc(&D.d_alpha, (size_t)n_cells)); CK(dmalloc(&D.d_lambda, (size_t)n_cells));
    CK(dmalloc(&D.d_err, (size_t)n_units)); CK(dmalloc(&D.d_order, (size_t)n_units));
    CK(dmalloc(&D.d_queue, (size_t)1)); CK(dmalloc(&D.d_status, (size_t)n_units));
    if (want_counters) CK(dmalloc(&D.d_cnt, (size_t)n_units * PAREBEN_NCOUNTERS));
#ifdef PAREBEN_PHASE_TIMERS
    if (phase_path) { CK(dmalloc(&D.d_phase, (size_t)n_units * PH_N)); CK(hipMemsetAsync(D.d_phase, 0, sizeof(long long) * (size_t)n_units * PH_N, c->stream)); }
#else
    (void)phase_path;
#endif
    CK(hipMemcpyAsync(D.d_alpha, alpha, sizeof(double) * n_cells, hipMemcpyHostToDevice, c->stream));
    CK(hipMemcpyAsync(D.d_lambda, lambda, sizeof(double) * n_cells, hipMemcpyHostToDevice, c->stream));
    CK(hipMemcpyAsync(D.d_order, D.order.data(), sizeof(int) * n_units, hipMemcpyHostToDevice, c->stream));
    CK(hipMemsetAsync(D.d_queue, 0, sizeof(int), c->stream));
    int *const queue = shared_queue ? shared_queue : D.d_queue;
    // PAREBEN_SHARE (A/B tests): 0 = no shared phases, 1 = in the tail only, 2 = from the start; unset = automatic
    const char *share_env = getenv("PAREBEN_SHARE");
    const int share_mode = share_env ? atoi(share_env) : -1;
    if (!binom && share_mode != 0) {
        CK(dmalloc(&D.d_jobs, (size_t)blocks)); CK(dmalloc(&D.d_active, (size_t)4));
        CK(hipMemsetAsync(D.d_jobs, 0, sizeof(FsJob) * (size_t)blocks, c->stream));
        D.act_host[0] = 0;                                      // workgroups that currently own a fit
        CK(hipMemcpyAsync(D.d_active, D.act_host, sizeof D.act_host, hipMemcpyHostToDevice, c->stream));
    }
    CK(hipMemsetAsync(D.d_err, 0xFF, sizeof(double) * n_units, c->stream));       // NaN-poison
    CK(hipMemsetAsync(D.d_status, 0xFF, sizeof(int) * n_units, c->stream));

    CK(hipEventRecord(c->ev[0], c->stream));
    rc = prepare_folds(c);
    if (rc) { D.release(); return rc; }
    CK(hipEventRecord(c->ev[1], c->stream));

    CvParams P;
    P.folds = c->d_folds; P.alpha = D.d_alpha; P.lambda = D.d_lambda; P.order = D.d_order; P.queue = queue; P.queue_sys = shared_queue ? 1 : 0;
    P.fold_err = D.d_err; P.status = D.d_status; P.counters = D.d_cnt; P.phase = D.d_phase; P.ws = c->d_ws;
    P.ws_stride = c->L.bytes; P.offK = c->L.offK; P.offSig = c->L.offSig; P.offM = c->L.offM;
    P.K = c->kfull; P.cap = c->cap; P.cap_flag = c->cap_ref; P.n_folds = nF; P.n_units = n_units; P.v = c->variant;
    P.priv_rows = c->priv_rows; P.priv_base0 = nF * c->pool_rows;
    P.jobs = D.d_jobs; P.active = D.d_active;
    // with only a handful of fits per workgroup the longest fits decide the step time: share from the start
    // (measured on config-2 shares with 256 workgroups: 1250 fits 2.51 -> 2.27 s, 2500 fits 3.82 -> 3.55 s, 5000 fits 6.15 -> 6.37 s)
    P.early = (D.d_jobs && (share_mode == 2 || (share_mode < 0 && n_units / n_sharers <= 10 * blocks))) ? 1 : 0;
    { const char *hm = getenv("PAREBEN_HEAVY_M"); P.heavy_m = share_mode == 1 ? (1 << 30) : (hm ? atoi(hm) : 384); }
    { const char *df = getenv("PAREBEN_DEFER"); P.defer = (df && atoi(df) == 0) ? 0 : 1; }
    { const char *ip = getenv("PAREBEN_INV_PAIR"); P.inv_pair = ip ? atoi(ip) & 3 : 3; }
    if (binom) {
        BmCvParams Q;
        Q.folds = c->d_folds; Q.alpha = D.d_alpha; Q.lambda = D.d_lambda; Q.order = D.d_order; Q.queue = queue; Q.queue_sys = shared_queue ? 1 : 0;
        Q.fold_err = D.d_err; Q.status = D.d_status; Q.counters = D.d_cnt; Q.ws = c->d_ws; Q.L = c->BL;
        Q.K = c->kfull; Q.n_folds = nF; Q.n_units = n_units; Q.phase = D.d_phase;
        Q.epis = c->epis; Q.bmax = 2 * c->p; Q.pool_n = bm_pool;
        hipLaunchKernelGGL(bm_cv_kernel, dim3(blocks), dim3(bm_threads), LDS_BYTES_FOR(bm_pool), c->stream, Q);
    } else if (c->strict) {
        P.offX = c->L.offX; P.nmax = c->L.nmax;
        HIPCHK(hipFuncSetAttribute((const void *)gm_cv_strict_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FIT_BYTES));
        hipLaunchKernelGGL(gm_cv_strict_kernel, dim3(blocks), dim3(FIT_THREADS), LDS_FIT_BYTES, c->stream, P);
    } else {
        hipLaunchKernelGGL(gm_cv_kernel, dim3(blocks), dim3(FIT_THREADS), LDS_FIT_BYTES, c->stream, P);
    }
    CK(hipGetLastError());
    CK(hipEventRecord(c->ev[2], c->stream));
#undef CK
    c->launch_info[0] = blocks; c->launch_info[1] = binom ? bm_threads : FIT_THREADS;
    c->launch_info[2] = binom ? c->BL.cap : c->cap;
    c->launch_info[4] = binom ? c->BL.cap : c->cap_ref;
    c->launch_info[3] = (int64_t)((binom ? c->BL.bytes : c->L.bytes) >> 10);
    return PAREBEN_OK;
}

// after the stream has been synchronised: event timings of the launch into the context
static int run_timings(pareben_ctx *c)
{
    float a = 0, b = 0, t = 0;
    HIPCHK(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    HIPCHK(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
    HIPCHK(hipEventElapsedTime(&t, c->ev[0], c->ev[3]));
    c->last_ms[0] = a; c->last_ms[1] = b; c->last_ms[2] = t;
    return PAREBEN_OK;
}

extern "C" int pareben_ctx_run(pareben_ctx *c, int n_cells, const double *alpha, const double *lambda,
                               double *fold_err, int32_t *status, int64_t *counters)
{
    if (!c || n_cells < 1 || !alpha || !lambda || !fold_err) return fail(PAREBEN_EINVAL, "bad argument");
    RunDev D;
    int rc = run_enqueue(c, n_cells, alpha, lambda, counters != nullptr, D);
    if (rc) return rc;
    const int n_units = D.n_units;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { D.release(); return fail(PAREBEN_EHIP, #x, e_); } } while (0)
    CK(hipMemcpyAsync(fold_err, D.d_err, sizeof(double) * n_units, hipMemcpyDeviceToHost, c->stream));
    std::vector<int> st(n_units);
    CK(hipMemcpyAsync(st.data(), D.d_status, sizeof(int) * n_units, hipMemcpyDeviceToHost, c->stream));
    if (counters) CK(hipMemcpyAsync(counters, D.d_cnt, sizeof(int64_t) * (size_t)n_units * PAREBEN_NCOUNTERS, hipMemcpyDeviceToHost, c->stream));
    CK(hipEventRecord(c->ev[3], c->stream));
    CK(hipStreamSynchronize(c->stream));
    if (status) for (int i = 0; i < n_units; i++) status[i] = st[i];
    if (D.d_phase) {
        std::vector<long long> ph((size_t)n_units * PH_N);
        CK(hipMemcpy(ph.data(), D.d_phase, sizeof(long long) * ph.size(), hipMemcpyDeviceToHost));
        if (FILE *fp = fopen(getenv("PAREBEN_PHASE_DUMP"), "wb")) { fwrite(ph.data(), sizeof(long long), ph.size(), fp); fclose(fp); }
    }
#undef CK
    rc = run_timings(c);
    D.release();
    return rc;
}

extern "C" int pareben_ctx_gram(pareben_ctx *c, int fold, double *out)
{
    if (!c || !out || fold < 0 || fold >= c->n_folds) return fail(PAREBEN_EINVAL, "bad argument");
    if (c->prior != PAREBEN_PRIOR_GAUSSIAN || c->lazy || !c->folds[fold].G) return fail(PAREBEN_EINVAL, "this context holds no whole Gram matrices");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    const int rc = prepare_folds(c);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float a = 0;
    HIPCHK(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    c->last_ms[0] = a; c->last_ms[1] = 0; c->last_ms[2] = a;
    HIPCHK(hipMemcpy(out, c->folds[fold].G, sizeof(double) * (size_t)c->kfull * c->kfull, hipMemcpyDeviceToHost));
    return PAREBEN_OK;
}

extern "C" int pareben_ctx_last_timing(pareben_ctx *c, double ms[3])
{
    if (!c || !ms) return fail(PAREBEN_EINVAL, "bad argument");
    ms[0] = c->last_ms[0]; ms[1] = c->last_ms[1]; ms[2] = c->last_ms[2];
    return PAREBEN_OK;
}

extern "C" int pareben_ctx_launch_info(pareben_ctx *c, int64_t info[5])
{
    if (!c || !info) return fail(PAREBEN_EINVAL, "bad argument");
    for (int i = 0; i < 5; i++) info[i] = c->launch_info[i];
    return PAREBEN_OK;
}

extern "C" int pareben_cv_grid(const double *basis, int n, int p, const double *target,
                               const int32_t *fold_id, int n_folds, const double *alpha,
                               const double *lambda, int n_cells, int epis, int prior, int device,
                               double *fold_err, int32_t *status, int64_t *counters)
{
    pareben_ctx *c = nullptr;
    int rc = pareben_ctx_create(&c, device, basis, n, p, target, fold_id, n_folds, prior, epis, 0);
    if (rc) return rc;
    rc = pareben_ctx_run(c, n_cells, alpha, lambda, fold_err, status, counters);
    pareben_ctx_destroy(c);
    return rc;
}

// ------------------------------------------------------------------------------------------
// The pairwise pass of GetLambdaMax (R/BuildGrid.R:21-30): for every pair i < j the column x_i * x_j, divided by
// its norm, is correlated with the centred (NOT normalised) target; the result is the largest such
// value.  O(n K^2) -- an interpreted double loop in the reference, the dominant cost of building an Epis grid there.
// One wavefront per pair, lanes over samples (both columns are contiguous: coalesced), two passes over the pair
// column (its squared norm, then the correlation of the normalised column: the same operations per element as the
// R expressions, summed in a fixed lane-strided + butterfly order), block maximum, then one compare-and-swap
// maximum per block on the result word.  A pair column that is identically zero gives 0/0 = NaN, which -- as in
// R's `if (corBy > lambda_Max)` -- never wins.
__global__ __launch_bounds__(256) void pair_lambda_kernel(const double *__restrict__ X, int N, int K,
                                                          const double *__restrict__ centred, long long n_pairs,
                                                          unsigned long long *__restrict__ best)
{
    extern __shared__ double lc[];                                // centred target (when it fits; else read from memory)
    __shared__ double wmax[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in_lds = N <= 8192;
    if (in_lds) for (int h = threadIdx.x; h < N; h += blockDim.x) lc[h] = centred[h];
    __syncthreads();
    double mine = -INFINITY;
    for (long long q = (long long)blockIdx.x * 4 + wave; q < n_pairs; q += (long long)gridDim.x * 4) {
        int i = (int)((2.0 * K - 1.0 - sqrt((2.0 * K - 1.0) * (2.0 * K - 1.0) - 8.0 * (double)q)) * 0.5);
        while ((long long)(i + 1) * K - (long long)(i + 1) * (i + 2) / 2 <= q) i++;
        while ((long long)i * K - (long long)i * (i + 1) / 2 > q) i--;
        const int j = (int)(q - ((long long)i * K - (long long)i * (i + 1) / 2)) + i + 1;
        const double *xi = X + (size_t)i * N, *xj = X + (size_t)j * N;
        double s2 = 0;
        for (int h = lane; h < N; h += 64) { const double z = xi[h] * xj[h]; s2 += z * z; }
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const double nrm = sqrt(s2);
        double d = 0;
        for (int h = lane; h < N; h += 64) { const double z = xi[h] * xj[h]; d += (z / nrm) * (in_lds ? lc[h] : centred[h]); }
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (d > mine) mine = d;                                   // NaN never compares greater
    }
    if (lane == 0) wmax[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = wmax[0];
        for (int w = 1; w < 4; w++) if (wmax[w] > m) m = wmax[w];
        unsigned long long cur = *best;                           // compare-and-swap maximum on the double's bits
        while (m > __longlong_as_double((long long)cur)) {
            const unsigned long long seen = atomicCAS(best, cur, (unsigned long long)__double_as_longlong(m));
            if (seen == cur) break;
            cur = seen;
        }
    }
}

extern "C" int pareben_lambda_max_pairs(const double *basis, int n, int p, const double *target, int device, double *out)
{
    if (!basis || !target || !out || n < 1 || p < 1) return fail(PAREBEN_EINVAL, "bad argument");
    *out = -INFINITY;
    if (p < 2) return PAREBEN_OK;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(PAREBEN_EINVAL, "no such device");
    HIPCHK(hipSetDevice(device));
    long double ms = 0;                                           // Target - mean(Target); R's mean(): long double sum / n, then one refinement pass
    for (int i = 0; i < n; i++) ms += target[i];
    ms /= n;
    { long double r = 0; for (int i = 0; i < n; i++) r += target[i] - ms; ms += r / n; }
    const double mean = (double)ms;
    std::vector<double> c(n);
    for (int i = 0; i < n; i++) c[i] = target[i] - mean;
    double *d_x = nullptr, *d_c = nullptr; unsigned long long *d_best = nullptr;
    auto cleanup = [&]() { hipFree(d_x); hipFree(d_c); hipFree(d_best); };
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return fail(e_ == hipErrorOutOfMemory ? PAREBEN_ENOMEM : PAREBEN_EHIP, #x, e_); } } while (0)
    CK(dmalloc(&d_x, (size_t)n * p)); CK(dmalloc(&d_c, (size_t)n)); CK(dmalloc(&d_best, (size_t)1));
    CK(hipMemcpy(d_x, basis, sizeof(double) * (size_t)n * p, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_c, c.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    const double ninf = -INFINITY;
    CK(hipMemcpy(d_best, &ninf, sizeof(double), hipMemcpyHostToDevice));
    const long long n_pairs = (long long)p * (p - 1) / 2;
    const int blocks = (int)std::min<long long>((n_pairs + 3) / 4, 256LL * 64);
    const size_t lds = n <= 8192 ? sizeof(double) * (size_t)n : 0;
    hipLaunchKernelGGL(pair_lambda_kernel, dim3(blocks), dim3(256), lds, 0, d_x, n, p, d_c, n_pairs, d_best);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, d_best, sizeof(double), hipMemcpyDeviceToHost));
#undef CK
    cleanup();
    return PAREBEN_OK;
}

// ------------------------------------------------------------------------------------------
// Multi-GPU grid evaluation behind the C ABI (SURVEY.md 8(b)/(e), R/CrossValidate.R:66-70): ONE host process (the caller is
// a single R session), one host thread + context per device.  The (cell, fold) units are NOT dealt out in advance: every
// GPU's persistent fit kernel pulls from ONE cost-sorted queue whose head is an int in pinned, coherent host memory
// (system-scope atomic add), so a GPU that drew the heavy fits simply takes fewer units.  Which GPU computed a unit
// never shows in its value (a fit's arithmetic depends on nothing outside the fit), so the table is bit-identical for any
// GPU count and any timing.  The path's only exchange is one grouped ncclAllGather (RCCL over xGMI) of the per-GPU result
// tables, after which every GPU holds every slice and the host merges them from the first (a unit's slot is valid on
// exactly the GPU whose status word for it is not the poison value).
// RCCL is bound at run time (dlopen, once per process) so the library loads and the single-GPU entries work without it;
// communicators are created on first use for a given GPU count and kept (pareben_multi_release frees them).
struct Rccl {
    void *h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static std::mutex g_multi_mu;                 // one multi-GPU call at a time per process (they would share every device anyway)
static Rccl g_rccl;
static std::vector<ncclComm_t> g_comms;       // communicators of the last GPU count used, device g = rank g
static int64_t g_multi_stats[4] = {0, 0, 0, 0};   // last call: ranks in the communicator, units pulled by the busiest / idlest GPU, GPUs

static int rccl_load(Rccl &R)
{
    if (R.h) return PAREBEN_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return fail(PAREBEN_EUNSUPPORTED, "RCCL (librccl.so) not found: the multi-GPU entry needs it for more than one GPU");
    R.CommInitAll = (decltype(R.CommInitAll))dlsym(h, "ncclCommInitAll");
    R.CommDestroy = (decltype(R.CommDestroy))dlsym(h, "ncclCommDestroy");
    R.GroupStart = (decltype(R.GroupStart))dlsym(h, "ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))dlsym(h, "ncclGroupEnd");
    R.AllGather = (decltype(R.AllGather))dlsym(h, "ncclAllGather");
    R.CommCount = (decltype(R.CommCount))dlsym(h, "ncclCommCount");
    R.GetErrorString = (decltype(R.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!R.CommInitAll || !R.CommDestroy || !R.GroupStart || !R.GroupEnd || !R.AllGather || !R.CommCount || !R.GetErrorString) {
        dlclose(h);
        return fail(PAREBEN_EUNSUPPORTED, "librccl.so lacks an expected symbol");
    }
    R.h = h;
    return PAREBEN_OK;
}

static void multi_release_locked()
{
    if (g_rccl.h) for (ncclComm_t c : g_comms) if (c) g_rccl.CommDestroy(c);
    g_comms.clear();
}
extern "C" int pareben_multi_release(void)
{
    std::lock_guard<std::mutex> lk(g_multi_mu);
    int dev = 0;
    const bool have = hipGetDevice(&dev) == hipSuccess;
    multi_release_locked();
    if (have) hipSetDevice(dev);
    return PAREBEN_OK;
}
extern "C" int pareben_multi_last_stats(int64_t out[4])
{
    if (!out) return fail(PAREBEN_EINVAL, "bad argument");
    for (int i = 0; i < 4; i++) out[i] = g_multi_stats[i];
    return PAREBEN_OK;
}

// one row per cell of this GPU's table: n_folds scores, then n_folds status words as doubles (poison -1 = not mine)
__global__ void pack_kernel(const double *__restrict__ err, const int *__restrict__ st, int n_cells, int nF, double *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    double *row = out + (size_t)c * (2 * nF);
    for (int f = 0; f < nF; f++) { row[f] = err[(size_t)c * nF + f]; row[nF + f] = (double)st[(size_t)c * nF + f]; }
}

extern "C" int pareben_cv_grid_multi(const double *basis, int n, int p, const double *target,
                                     const int32_t *fold_id, int n_folds,
                                     const double *alpha, const double *lambda, int n_cells,
                                     int epis, int prior, int n_gpu,
                                     double *fold_err, int32_t *status, int64_t *counters)
{
    if (!basis || !target || !fold_id || !alpha || !lambda || !fold_err || n_cells < 1 || n_folds < 1) return fail(PAREBEN_EINVAL, "bad argument");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (n_gpu <= 0) n_gpu = ndev;
    if (n_gpu < 1 || n_gpu > ndev) return fail(PAREBEN_EINVAL, "n_gpu exceeds the visible devices");
    // Test hook: PAREBEN_MULTI_DEVICES="0,0" runs that many ranks on the named devices (the same one may repeat) with the
    // merge done on the host instead of RCCL, which refuses duplicate devices -- the N > 1 deal and merge on a one-GPU box.
    std::vector<int> devs;
    bool host_gather = false;
    if (const char *e = getenv("PAREBEN_MULTI_DEVICES")) {
        for (const char *q = e; *q;) { devs.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
        for (int d : devs) if (d < 0 || d >= ndev) return fail(PAREBEN_EINVAL, "PAREBEN_MULTI_DEVICES names a device that is not visible");
        if (!devs.empty()) { n_gpu = (int)devs.size(); host_gather = true; }
    }
    if (devs.empty()) { devs.resize(n_gpu); std::iota(devs.begin(), devs.end(), 0); }
    std::lock_guard<std::mutex> lk(g_multi_mu);
    int dev_on_entry = 0;
    HIPCHK(hipGetDevice(&dev_on_entry));
    struct Restore { int d; ~Restore() { hipSetDevice(d); } } restore{dev_on_entry};

    const int nF = n_folds, row = 2 * nF, n_units = n_cells * nF;
    if (n_gpu == 1) host_gather = true;                         // one GPU: nothing to exchange, RCCL is not touched
    // communicators first (cached per GPU count), so that their creation never competes with the persistent fit kernels
    if (!host_gather) {
        int rc = rccl_load(g_rccl);
        if (rc) return rc;
        if ((int)g_comms.size() != n_gpu) {
            multi_release_locked();
            g_comms.assign(n_gpu, nullptr);
            const ncclResult_t nr = g_rccl.CommInitAll(g_comms.data(), n_gpu, devs.data());
            if (nr != ncclSuccess) { g_comms.clear(); std::string m = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr); return fail(PAREBEN_EHIP, m.c_str()); }
        }
        int cnt = 0;
        if (g_rccl.CommCount(g_comms[0], &cnt) != ncclSuccess || cnt != n_gpu) return fail(PAREBEN_EHIP, "RCCL communicator has the wrong size");
        g_multi_stats[0] = cnt;
    } else g_multi_stats[0] = 1;
    g_multi_stats[3] = n_gpu;

    // the one queue head all GPUs pull from: pinned, coherent host memory, visible to every device
    int *queue = nullptr;
    if (hipHostMalloc((void **)&queue, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
        return fail(PAREBEN_ENOMEM, "pinned queue head");
    *queue = 0;
    struct Rank {
        pareben_ctx *ctx = nullptr; RunDev D; double *d_send = nullptr, *d_recv = nullptr; int rc = 0; std::string err;
        std::vector<double> tab; std::vector<int64_t> cnt;
    };
    std::vector<Rank> R(n_gpu);
    auto work = [&](int g) {
        Rank &r = R[g];
        auto bail = [&](int code) { r.rc = code; r.err = pareben_last_error(); };
        int rc = pareben_ctx_create(&r.ctx, devs[g], basis, n, p, target, fold_id, n_folds, prior, epis, 0);
        if (rc) return bail(rc);
        rc = run_enqueue(r.ctx, n_cells, alpha, lambda, counters != nullptr, r.D, queue, n_gpu);
        if (rc) return bail(rc);
        hipStream_t s = r.ctx->stream;
        if (dmalloc(&r.d_send, (size_t)n_cells * row) != hipSuccess ||
            (!host_gather && dmalloc(&r.d_recv, (size_t)n_gpu * n_cells * row) != hipSuccess)) return bail(fail(PAREBEN_ENOMEM, "gather buffers"));
        hipLaunchKernelGGL(pack_kernel, dim3((n_cells + 127) / 128), dim3(128), 0, s, r.D.d_err, r.D.d_status, n_cells, nF, r.d_send);
        if (counters) {                                          // diagnostics only: straight to the host, not part of the exchange
            r.cnt.resize((size_t)n_units * PAREBEN_NCOUNTERS);
            if (hipMemcpyAsync(r.cnt.data(), r.D.d_cnt, sizeof(int64_t) * r.cnt.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return bail(fail(PAREBEN_EHIP, "counter copy"));
        }
        if (host_gather) {
            r.tab.resize((size_t)n_cells * row);
            if (hipMemcpyAsync(r.tab.data(), r.d_send, sizeof(double) * r.tab.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return bail(fail(PAREBEN_EHIP, "table copy"));
        }
        if (hipEventRecord(r.ctx->ev[3], s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return bail(fail(PAREBEN_EHIP, "fit launch", hipGetLastError()));
        run_timings(r.ctx);
    };
    std::vector<std::thread> th;
    for (int g = 0; g < n_gpu; g++) th.emplace_back(work, g);
    for (auto &t : th) t.join();
    auto cleanup = [&]() {
        for (int g = 0; g < n_gpu; g++) {
            if (R[g].ctx) hipSetDevice(devs[g]);
            hipFree(R[g].d_send); hipFree(R[g].d_recv);
            R[g].D.release();
            if (R[g].ctx) pareben_ctx_destroy(R[g].ctx);
        }
        hipHostFree(queue);
    };
    for (int g = 0; g < n_gpu; g++) if (R[g].rc) { const int code = R[g].rc; const std::string m = R[g].err; cleanup(); return fail(code, m.c_str()); }
    std::vector<double> tab;                                     // [gpu][cell][2 nF]
    if (host_gather) {
        tab.resize((size_t)n_gpu * n_cells * row);
        for (int g = 0; g < n_gpu; g++) memcpy(tab.data() + (size_t)g * n_cells * row, R[g].tab.data(), sizeof(double) * (size_t)n_cells * row);
    } else {
        // the path's one collective: every GPU contributes its table, every GPU ends with all of them
        ncclResult_t nr = g_rccl.GroupStart();
        for (int g = 0; g < n_gpu && nr == ncclSuccess; g++)
            nr = g_rccl.AllGather(R[g].d_send, R[g].d_recv, (size_t)n_cells * row, ncclDouble, g_comms[g], R[g].ctx->stream);
        { const ncclResult_t ne = g_rccl.GroupEnd(); if (nr == ncclSuccess) nr = ne; }
        if (nr != ncclSuccess) { std::string m = std::string("ncclAllGather: ") + g_rccl.GetErrorString(nr); cleanup(); return fail(PAREBEN_EHIP, m.c_str()); }
        for (int g = 0; g < n_gpu; g++) {
            hipSetDevice(devs[g]);
            if (hipStreamSynchronize(R[g].ctx->stream) != hipSuccess) { cleanup(); return fail(PAREBEN_EHIP, "all-gather", hipGetLastError()); }
        }
        tab.resize((size_t)n_gpu * n_cells * row);
        hipSetDevice(devs[0]);
        if (hipMemcpy(tab.data(), R[0].d_recv, sizeof(double) * tab.size(), hipMemcpyDeviceToHost) != hipSuccess) { cleanup(); return fail(PAREBEN_EHIP, "table copy", hipGetLastError()); }
    }
    // merge: a unit belongs to the one GPU whose status word for it is not the poison value
    int seen = 0, dup = 0;
    std::vector<int64_t> pulled(n_gpu, 0);
    for (int c = 0; c < n_cells; c++)
        for (int f = 0; f < nF; f++) {
            int owner = -1;
            for (int g = 0; g < n_gpu; g++) {
                const double *rw = tab.data() + ((size_t)g * n_cells + c) * row;
                if (rw[nF + f] != -1.0) { if (owner >= 0) dup++; owner = g; }
            }
            if (owner < 0) continue;
            const double *rw = tab.data() + ((size_t)owner * n_cells + c) * row;
            fold_err[(size_t)c * nF + f] = rw[f];
            if (status) status[(size_t)c * nF + f] = (int32_t)rw[nF + f];
            if (counters) memcpy(counters + ((size_t)c * nF + f) * PAREBEN_NCOUNTERS, R[owner].cnt.data() + ((size_t)c * nF + f) * PAREBEN_NCOUNTERS, sizeof(int64_t) * PAREBEN_NCOUNTERS);
            pulled[owner]++;
            seen++;
        }
    g_multi_stats[1] = *std::max_element(pulled.begin(), pulled.end());
    g_multi_stats[2] = *std::min_element(pulled.begin(), pulled.end());
    cleanup();
    if (seen != n_units || dup) return fail(PAREBEN_EHIP, "the merged table is incomplete or a unit was computed twice");
    return PAREBEN_OK;
}

// Diagnostics: the next pareben_fit_gaussian[_epis] calls of this thread record one TR_NSLOT-word record per inner
// iteration (types.h TR_*: decision, margins, XOR hashes of the state) into buf: buf[0] = records written, records
// from buf + TR_NSLOT; at most max_records.  NULL switches it off.  tools/trace_divergence.py compares it with the
// oracle's trace of the same fit.
static thread_local uint64_t *g_trace_buf = nullptr;
static thread_local int64_t g_trace_cap = 0;
extern "C" int pareben_set_trace(uint64_t *buf, int64_t max_records)
{
    if (buf && max_records < 1) return fail(PAREBEN_EINVAL, "bad argument");
    g_trace_buf = buf; g_trace_cap = buf ? max_records : 0;
    return PAREBEN_OK;
}

// one fit on all rows: a pseudo-fold whose training set is every row and whose held-out set is empty
// `verbose`: the reference's Rprintf trace, printed to stdout after the launch from what the kernel recorded --
// level > 0 basisMax and "outer loop starts" (MainEff.c:70, :139), > 1 the start / finish lines (:71, :205; binomial
// NEmainEff.c:327, :352), > 2 one line per outer iteration (:196; binomial :342), > 4 one line per inner iteration
// (:405) from the decision trace (Gaussian only).
static int fit_one(int prior, int epis, const double *basis, const double *target, double lambda, double alpha,
                   int n, int k, int device, double *Beta, double *scalars_out, int n_scalars, int64_t *counters, int verbose = 0)
{
    std::vector<std::vector<int>> tr(1), te(1);
    tr[0].resize(n);
    std::iota(tr[0].begin(), tr[0].end(), 0);
    pareben_ctx *c = nullptr;
    int rc = ctx_create_impl(&c, device, basis, n, k, target, tr, te, prior, epis, 0);
    if (rc) return rc;
    auto bail = [&](int code) { pareben_ctx_destroy(c); return code; };
    if (hipSetDevice(c->device) != hipSuccess) return bail(fail(PAREBEN_EHIP, "hipSetDevice"));
    rc = prepare_folds(c);
    if (rc) return bail(rc);
    rc = ensure_workspace(c, 1);
    if (rc) return bail(rc);
    const int ncol = (epis && prior == PAREBEN_PRIOR_GAUSSIAN) ? 5 : 4;
    const size_t KF = (epis && prior == PAREBEN_PRIOR_BINOMIAL) ? (size_t)2 * k : (size_t)c->kfull;     // rows of the Beta table
    double *d_beta = nullptr, *d_sc = nullptr; int *d_st = nullptr; long long *d_cnt = nullptr;
    FsJob *d_jobs = nullptr; int *d_flags = nullptr; unsigned long long *d_trace = nullptr; double *d_olog = nullptr;
    auto cleanup = [&]() { hipFree(d_beta); hipFree(d_sc); hipFree(d_st); hipFree(d_cnt); hipFree(d_jobs); hipFree(d_flags); hipFree(d_trace); hipFree(d_olog); };
    std::vector<uint64_t> vtrace;                              // verbose > 4 without a caller's trace buffer: an internal one
    uint64_t *trace_host = g_trace_buf; int64_t trace_cap = g_trace_cap;
    if (verbose > 4 && !trace_host && prior == PAREBEN_PRIOR_GAUSSIAN) { trace_cap = 20000; vtrace.assign((size_t)TR_NSLOT * (trace_cap + 1), 0); trace_host = vtrace.data(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { cleanup(); return bail(fail(PAREBEN_EHIP, #x, e_)); } } while (0)
    CK(dmalloc(&d_beta, KF * ncol)); CK(dmalloc(&d_sc, (size_t)4)); CK(dmalloc(&d_st, (size_t)1));
    CK(dmalloc(&d_cnt, (size_t)PAREBEN_NCOUNTERS));
    CK(hipStreamSynchronize(c->stream));
    FoldDev F;
    CK(hipMemcpy(&F, c->d_folds, sizeof(FoldDev), hipMemcpyDeviceToHost));
    if (prior == PAREBEN_PRIOR_GAUSSIAN) {
        FitParams P;
        P.F = F; P.lambda = lambda; P.alpha = alpha; P.Beta = d_beta; P.scalars = d_sc; P.status = d_st; P.counters = d_cnt;
        P.ws = c->d_ws; P.offK = c->L.offK; P.offSig = c->L.offSig; P.offM = c->L.offM; P.K = c->kfull; P.cap = c->cap; P.cap_flag = c->cap_ref;
        P.p = k; P.v = c->variant;
        P.trace = nullptr; P.trace_cap = 0; P.outer_log = nullptr;
        { const char *df = getenv("PAREBEN_DEFER"); P.defer = (df && atoi(df) == 0) ? 0 : 1; }
    { const char *ip = getenv("PAREBEN_INV_PAIR"); P.inv_pair = ip ? atoi(ip) & 3 : 3; }
        if (trace_host) {
            CK(dmalloc(&d_trace, (size_t)TR_NSLOT * (trace_cap + 1)));
            CK(hipMemset(d_trace, 0, sizeof(unsigned long long) * TR_NSLOT * (trace_cap + 1)));
            P.trace = d_trace; P.trace_cap = trace_cap;
        }
        if (verbose > 2) { CK(dmalloc(&d_olog, (size_t)300)); CK(hipMemset(d_olog, 0, sizeof(double) * 300)); P.outer_log = d_olog; }
        CK(hipFuncSetAttribute((const void *)gm_fit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FIT_BYTES));
        // helpers: as many workgroups as are resident at once (they wait for the owner, so none may be left undispatched
        // in front of it); PAREBEN_SHARE=0 runs the fit alone
        int blocks = 1;
        const char *share_env = getenv("PAREBEN_SHARE");
        P.jobs = nullptr; P.active = nullptr; P.queue = nullptr; P.folds = c->d_folds; P.ws_stride = c->L.bytes;
        if (!(share_env && atoi(share_env) == 0)) {
            int occ = 1;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gm_fit_kernel, FIT_THREADS, LDS_FIT_BYTES));
            blocks = std::max(1, c->n_cu * std::max(occ, 1));
            CK(dmalloc(&d_jobs, (size_t)blocks)); CK(dmalloc(&d_flags, (size_t)8));
            CK(hipMemsetAsync(d_jobs, 0, sizeof(FsJob) * (size_t)blocks, c->stream));
            const int flags[8] = {1, 0, 0, 0, 1, 0, 0, 0};          // [0] active: the owner holds its fit; [4] queue head = n_units: drained
            CK(hipMemcpyAsync(d_flags, flags, sizeof flags, hipMemcpyHostToDevice, c->stream));
            CK(hipStreamSynchronize(c->stream));                     // `flags` is a stack array
            P.jobs = d_jobs; P.active = d_flags; P.queue = d_flags + 4;
        }
        if (c->strict) {
            P.offX = c->L.offX; P.nmax = c->L.nmax;
            CK(hipFuncSetAttribute((const void *)gm_fit_strict_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FIT_BYTES));
            hipLaunchKernelGGL(gm_fit_strict_kernel, dim3(1), dim3(FIT_THREADS), LDS_FIT_BYTES, c->stream, P);
        } else
            hipLaunchKernelGGL(gm_fit_kernel, dim3(blocks), dim3(FIT_THREADS), LDS_FIT_BYTES, c->stream, P);
    } else {
        BmFitParams P;
        P.F = F; P.lambda = lambda; P.alpha = alpha; P.Beta = d_beta; P.scalars = d_sc; P.status = d_st; P.counters = d_cnt;
        P.ws = c->d_ws; P.L = c->BL; P.K = c->kfull; P.epis = epis; P.p = k; P.bmax = 2 * k;
        P.outer_log = nullptr;
        if (verbose > 2) { CK(dmalloc(&d_olog, (size_t)300)); CK(hipMemset(d_olog, 0, sizeof(double) * 300)); P.outer_log = d_olog; }
        CK(hipFuncSetAttribute((const void *)bm_fit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FIT_BYTES));
        hipLaunchKernelGGL(bm_fit_kernel, dim3(1), dim3(FIT_THREADS), LDS_FIT_BYTES, c->stream, P);
    }
    CK(hipGetLastError());
    CK(hipStreamSynchronize(c->stream));
    double sc[4]; int st = 0;
    CK(hipMemcpy(Beta, d_beta, sizeof(double) * KF * ncol, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sc, d_sc, sizeof sc, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&st, d_st, sizeof st, hipMemcpyDeviceToHost));
    if (counters) CK(hipMemcpy(counters, d_cnt, sizeof(int64_t) * PAREBEN_NCOUNTERS, hipMemcpyDeviceToHost));
    if (d_trace) CK(hipMemcpy(trace_host, d_trace, sizeof(unsigned long long) * TR_NSLOT * (trace_cap + 1), hipMemcpyDeviceToHost));
    if (verbose > 0) {
        int64_t cn[PAREBEN_NCOUNTERS];
        double olog[300] = {0};
        CK(hipMemcpy(cn, d_cnt, sizeof cn, hipMemcpyDeviceToHost));
        if (d_olog) CK(hipMemcpy(olog, d_olog, sizeof olog, hipMemcpyDeviceToHost));
        const bool gauss = prior == PAREBEN_PRIOR_GAUSSIAN;
        if (gauss) printf("basisMax: %d", c->cap_ref);
        if (verbose > 1) printf(gauss ? "start EB-elasticNet with alpha: %f, lambda: %f\n" : "Empirical Bayesian Elastic Net outer loop starts\n", alpha, lambda);
        if (gauss) printf("outer loop starts");
        size_t r = 0;
        const size_t n_rec = (trace_host && verbose > 4) ? (size_t)trace_host[0] : 0;
        for (int it = 1; it <= (int)cn[0]; it++) {
            for (; r < n_rec && (int)trace_host[TR_NSLOT * (r + 1) + TR_ITER] == it; r++) {
                const uint64_t *t = trace_host + TR_NSLOT * (r + 1);
                printf("\t inner loop %d; number of basis: %d \t actionStatus: %d \tnu: %d\n", (int)t[TR_IITER], (int)t[TR_MAFTER], (int)(int64_t)t[TR_SEL], (int)(int64_t)t[TR_NU] + 1);
            }
            if (verbose > 2) {
                if (gauss) printf("Iteration number: %d, err: %f;\t mu: %f\tsigma0:%f.\n", it, olog[3 * (it - 1)], olog[3 * (it - 1) + 1], olog[3 * (it - 1) + 2]);
                else printf("Iteration number: %d, err: %f\n", it, olog[3 * (it - 1)]);
            }
        }
        if (verbose > 1) printf("EBEN Finished, number of basis: %d\n", (int)cn[9]);
        fflush(stdout);
    }
#undef CK
    for (int i = 0; i < n_scalars; i++) scalars_out[i] = sc[i];
    cleanup();
    pareben_ctx_destroy(c);
    return (st & ST_ABORT) ? fail(PAREBEN_EHIP, "fit aborted (see status bits in counters[11])") : PAREBEN_OK;
}

extern "C" int pareben_fit_gaussian(const double *basis, const double *target, double lambda, double alpha,
                                    double *Beta, double *wald, double *intercept, int n, int k,
                                    int verbose, double *residual, int device, int64_t *counters)
{
    if (!basis || !target || !Beta || !wald || !intercept || !residual || n < 2 || k < 1) return fail(PAREBEN_EINVAL, "bad argument");
    double sc[3];
    const int rc = fit_one(PAREBEN_PRIOR_GAUSSIAN, 0, basis, target, lambda, alpha, n, k, device, Beta, sc, 3, counters, verbose);
    if (rc == PAREBEN_OK) { *wald = sc[0]; *intercept = sc[1]; *residual = sc[2]; }
    return rc;
}

extern "C" int pareben_fit_gaussian_epis(const double *basis, const double *target, double lambda, double alpha,
                                         double *Beta, double *wald, double *intercept, int n, int k,
                                         int verbose, double *residual, int device, int64_t *counters)
{
    if (!basis || !target || !Beta || !wald || !intercept || !residual || n < 2 || k < 2) return fail(PAREBEN_EINVAL, "bad argument");
    if ((long long)k * (k + 1) / 2 > 2000000000LL) return fail(PAREBEN_EINVAL, "too many pairwise columns");
    double sc[3];
    const int rc = fit_one(PAREBEN_PRIOR_GAUSSIAN, 1, basis, target, lambda, alpha, n, k, device, Beta, sc, 3, counters, verbose);
    if (rc == PAREBEN_OK) { *wald = sc[0]; *intercept = sc[1]; *residual = sc[2]; }
    return rc;
}

extern "C" int pareben_fit_binomial(const double *basis, const double *target, double lambda, double alpha,
                                    double *logLikelihood, double *Beta, double *wald, double *intercept,
                                    int n, int k, int verbose, int bMax, int device, int64_t *counters)
{
    (void)bMax;
    if (!basis || !target || !Beta || !wald || !intercept || !logLikelihood || n < 2 || k < 1) return fail(PAREBEN_EINVAL, "bad argument");
    double sc[4];
    const int rc = fit_one(PAREBEN_PRIOR_BINOMIAL, 0, basis, target, lambda, alpha, n, k, device, Beta, sc, 4, counters, verbose);
    if (rc == PAREBEN_OK) { *logLikelihood = sc[0]; *wald = sc[1]; intercept[0] = sc[2]; intercept[1] = sc[3]; }
    return rc;
}

extern "C" int pareben_fit_binomial_epis(const double *basis, const double *target, double lambda, double alpha,
                                         double *logLikelihood, double *Beta, double *wald, double *intercept,
                                         int n, int k, int verbose, int bMax, int device, int64_t *counters)
{
    if (!basis || !target || !Beta || !wald || !intercept || !logLikelihood || n < 2 || k < 2) return fail(PAREBEN_EINVAL, "bad argument");
    if (bMax != 2 * k) return fail(PAREBEN_EINVAL, "bMax must be 2*k, what EBelasticNet.Binomial passes (Beta is bMax x 4)");
    if ((long long)k * (k + 1) / 2 > 2000000000LL) return fail(PAREBEN_EINVAL, "too many pairwise columns");
    double sc[4];
    const int rc = fit_one(PAREBEN_PRIOR_BINOMIAL, 1, basis, target, lambda, alpha, n, k, device, Beta, sc, 4, counters, verbose);
    if (rc == PAREBEN_OK) { *logLikelihood = sc[0]; *wald = sc[1]; intercept[0] = sc[2]; intercept[1] = sc[3]; }
    return rc;
}

// ------------------------------------------------------------------------------------------
// The reference's own .C entry points: same symbol names, argument order and pointer-only calling convention as
// EBEN_orig/src (R's .C passes every argument as a pointer and ignores the return value), so that
// .C("elasticNetLinearNeMainEff", ..., PACKAGE = "pareben_hip") in EBelasticNet.Gaussian / .Binomial binds them with
// nothing but PACKAGE changed.  Outputs are written in place.  .C has no error channel: on failure the message goes
// to stderr and the scalar outputs are set to NaN (the reference carries on into undefined behaviour instead).
// Device: PAREBEN_DEVICE (default 0).
static int dotc_device(void) { const char *e = getenv("PAREBEN_DEVICE"); return e ? atoi(e) : 0; }
static void dotc_failed(const char *entry, double *a, double *b, double *c2)
{
    fprintf(stderr, "%s: %s\n", entry, pareben_last_error());
    const double nan = __builtin_nan("");
    if (a) *a = nan; if (b) *b = nan; if (c2) *c2 = nan;
}
// EBEN_orig/src/elasticNetLinearNeMainEff.c:55-57, called from EBEN_orig/R/EBelasticNet.Gaussian.R:38-51
extern "C" void elasticNetLinearNeMainEff(double *BASIS, double *y, double *a_lambda, double *b_Alpha, double *Beta,
                                          double *wald, double *intercept, int *n, int *kdim, int *verb, double *residual)
{
    if (pareben_fit_gaussian(BASIS, y, *a_lambda, *b_Alpha, Beta, wald, intercept, *n, *kdim, verb ? *verb : 0, residual, dotc_device(), nullptr))
        dotc_failed("elasticNetLinearNeMainEff", wald, intercept, residual);
}
// EBEN_orig/src/elasticNetLinearNeFull2.c:57-58, called from EBEN_orig/R/EBelasticNet.Gaussian.R:16-29
extern "C" void elasticNetLinearNeEpisEff(double *BASIS, double *y, double *a_lambda, double *b_Alpha, double *Beta,
                                          double *wald, double *intercept, int *n, int *kdim, int *VB, double *residual)
{
    if (pareben_fit_gaussian_epis(BASIS, y, *a_lambda, *b_Alpha, Beta, wald, intercept, *n, *kdim, VB ? *VB : 0, residual, dotc_device(), nullptr))
        dotc_failed("elasticNetLinearNeEpisEff", wald, intercept, residual);
}
// EBEN_orig/src/ElasticNetBinaryNEmainEff.c:236-238, called from EBEN_orig/R/EBelasticNet.Binomial.R:32-46
extern "C" void ElasticNetBinaryNEmainEff(double *BASIS, double *Targets, double *a_Lambda, double *b_Alpha, double *logLIKELIHOOD,
                                          double *Beta, double *wald, double *intercept, int *n, int *kdim, int *VB, int *bMax)
{
    if (pareben_fit_binomial(BASIS, Targets, *a_Lambda, *b_Alpha, logLIKELIHOOD, Beta, wald, intercept, *n, *kdim, VB ? *VB : 0,
                             bMax ? *bMax : *kdim, dotc_device(), nullptr))
        dotc_failed("ElasticNetBinaryNEmainEff", logLIKELIHOOD, wald, intercept);
}
// EBEN_orig/src/ElasticNetBinaryNeFull.c:52-55, called from EBEN_orig/R/EBelasticNet.Binomial.R:10-24
extern "C" void ElasticNetBinaryNEfull(double *BASIS, double *Targets, double *a_Lambda, double *b_Alpha, double *logLIKELIHOOD,
                                       double *Beta, double *wald, double *intercept, int *n, int *kdim, int *VB, int *bMax)
{
    if (pareben_fit_binomial_epis(BASIS, Targets, *a_Lambda, *b_Alpha, logLIKELIHOOD, Beta, wald, intercept, *n, *kdim, VB ? *VB : 0,
                                  bMax ? *bMax : 2 * *kdim, dotc_device(), nullptr))
        dotc_failed("ElasticNetBinaryNEfull", logLIKELIHOOD, wald, intercept);
}

#ifdef PAREBEN_DIAG
// ------------------------------------------------------------------------------------------
// Diagnostic build only (-DPAREBEN_DIAG, tools/ubench/fullstat_rate.py): time the full-stat feature
// pass alone on `blocks` workgroups, each with its own Sigma (cap x cap) and a shared M x K Gram.
struct DiagParams { const double *G; char *ws; size_t stride, offK, offSig, offM; int K, cap, M, reps; };
__global__ __launch_bounds__(FIT_THREADS, FIT_WAVES_PER_EU) void diag_fullstat_kernel(DiagParams P)
{
    const Blk B = make_blk();
    GmWork W = ws_carve(P.ws + (size_t)blockIdx.x * P.stride, P.K, P.cap, P.offK, P.offSig, P.offM);
    FoldDev F{};
    F.G = P.G;
    for (int i = threadIdx.x; i < P.M; i += blockDim.x) { W.rowid[i] = i; W.used[i] = i; W.mu[i] = 0.001 * i; }
    __syncthreads();
    for (int r = 0; r < P.reps; r++) gm_fullstat_features(B, F, W, P.K, P.M, 1.0, 0, (P.K + FS_FT - 1) / FS_FT);
}
extern "C" int pareben_diag_fullstat(int M, int K, int blocks, int reps, double *ms_out)
{
    const int cap = ((M + 15) / 16) * 16 + 16;
    WsLayout L = ws_layout(K, cap);
    char *ws = nullptr; double *G = nullptr;
    if (hipMalloc((void **)&ws, L.bytes * (size_t)blocks) != hipSuccess) return PAREBEN_ENOMEM;
    if (hipMalloc((void **)&G, sizeof(double) * (size_t)M * K) != hipSuccess) return PAREBEN_ENOMEM;
    std::vector<double> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); i++) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    hipMemcpy(G, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    hipMemset(ws, 0, L.bytes * (size_t)blocks);
    DiagParams P{G, ws, L.bytes, L.offK, L.offSig, L.offM, K, cap, M, reps};
    hipFuncSetAttribute((const void *)diag_fullstat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FIT_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    P.reps = 1;
    hipLaunchKernelGGL(diag_fullstat_kernel, dim3(blocks), dim3(FIT_THREADS), LDS_FIT_BYTES, 0, P);   // warm-up
    P.reps = reps;
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(diag_fullstat_kernel, dim3(blocks), dim3(FIT_THREADS), LDS_FIT_BYTES, 0, P);
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) return PAREBEN_EHIP;
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / reps;
    hipFree(ws); hipFree(G); hipEventDestroy(e0); hipEventDestroy(e1);
    return PAREBEN_OK;
}
// Diagnostic build only (tools/ubench/inverse_rate.py): the blocked inverse alone on `blocks` workgroups, each on its own
// SPD matrix (diagonally dominant, cap = `cap` so that the matrices lie as far apart as in a real launch); per repetition
// the upper-stored triangle is copied from W.H into W.Sig and inverted.  phase_out[PH_N]: block 0's phase ticks
// (-DPAREBEN_PHASE_TIMERS), else zeros.
struct DiagInvParams { char *ws; size_t stride, offK, offSig, offM; int K, cap, M, reps, pair; long long *ph; double *chk; };
__global__ __launch_bounds__(FIT_THREADS, FIT_WAVES_PER_EU) void diag_inverse_kernel(DiagInvParams P)
{
    const Blk B = make_blk();
    __shared__ long long s_ph[PH_N];
    GmWork W = ws_carve(P.ws + (size_t)blockIdx.x * P.stride, P.K, P.cap, P.offK, P.offSig, P.offM);
    const int M = P.M, ld = W.ld;
    if (threadIdx.x < PH_N) s_ph[threadIdx.x] = 0;
    for (int e = threadIdx.x; e < M * M; e += blockDim.x) {
        const int j = e / M, i = e - j * M;
        if (i >= j) {
            const unsigned h = (unsigned)(i * 7919 + j * 104729 + blockIdx.x * 31) * 2654435761u;
            W.H[(size_t)j * ld + i] = (i == j) ? 2.0 + 0.01 * (h % 97) : ((double)((h >> 8) % 2001) - 1000.0) * (1e-3 / M);
        }
    }
    __syncthreads();
    int bad = 0;
    for (int r = 0; r < P.reps; r++) {
        for (int e = threadIdx.x; e < M * M; e += blockDim.x) {
            const int j = e / M, i = e - j * M;
            if (i >= j) W.Sig[(size_t)j * ld + i] = W.H[(size_t)j * ld + i];
        }
        __syncthreads();
        bad |= gm_spd_inverse(B, W, M, s_ph, P.pair);
    }
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0 && P.ph) for (int k = 0; k < PH_N; k++) P.ph[k] = s_ph[k];
        double c = bad ? -1.0 : 0.0;
        if (!bad) for (int i = 0; i < M; i++) c += W.Sig[(size_t)i * ld + i] + W.Sig[(size_t)(M - 1) * ld + i];
        P.chk[blockIdx.x] = c;
    }
}
extern "C" int pareben_diag_inverse(int M, int cap, int blocks, int reps, int pair, double *ms_out, long long *phase_out, double *chk_out)
{
    const int K = 64;
    WsLayout L = ws_layout(K, cap);
    char *ws = nullptr; long long *ph = nullptr; double *chk = nullptr;
    if (hipMalloc((void **)&ws, L.bytes * (size_t)blocks) != hipSuccess) return PAREBEN_ENOMEM;
    hipMalloc((void **)&ph, sizeof(long long) * PH_N); hipMalloc((void **)&chk, sizeof(double) * blocks);
    hipMemset(ws, 0, L.bytes * (size_t)blocks); hipMemset(ph, 0, sizeof(long long) * PH_N);
    DiagInvParams P{ws, L.bytes, L.offK, L.offSig, L.offM, K, cap, M, 1, pair, ph, chk};
    hipFuncSetAttribute((const void *)diag_inverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FIT_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(diag_inverse_kernel, dim3(blocks), dim3(FIT_THREADS), LDS_FIT_BYTES, 0, P);   // warm-up
    P.reps = reps;
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(diag_inverse_kernel, dim3(blocks), dim3(FIT_THREADS), LDS_FIT_BYTES, 0, P);
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) return PAREBEN_EHIP;
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / reps;
    hipMemcpy(phase_out, ph, sizeof(long long) * PH_N, hipMemcpyDeviceToHost);
    hipMemcpy(chk_out, chk, sizeof(double) * blocks, hipMemcpyDeviceToHost);
    hipFree(ws); hipFree(ph); hipFree(chk); hipEventDestroy(e0); hipEventDestroy(e1);
    return PAREBEN_OK;
}
#endif
