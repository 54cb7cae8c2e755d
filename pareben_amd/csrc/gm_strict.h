// gm_strict.h -- PAREBEN_STRICT_ORDER=1: one Gaussian main-effect EBEN fit in the REFERENCE'S OWN formulation and
// operation order, spread over a workgroup only across independent outputs.
//
// Why it exists.  The production fit (gm_fit.h + gm_dev.h) works in Gram space with fused multiply-adds, fixed reduction
// trees and a sweep-operator inverse; its results agree with the reference to rounding (1e-13 ... 1e-9).  On the long
// add/delete trajectories of the stored real-R tables (alpha = 1, duplicated genotype columns) that is not enough: a twin
// outside the model and its twin inside it with an astronomically large precision have add / re-estimate dML values that
// differ in the last bit or not at all, the winner's action type sets the block cut-off, and every summation order decides
// some of those coin flips differently (tools/first_divergence.py: decision margins of 0 ... 1e-15 against 1e-12 of noise
// between two orders).  This mode removes the order: every quantity is computed with the operations, the association and
// the sequential accumulation order of EBEN_orig/src/elasticNetLinearNeMainEff.c as R 3.5 ran it with its reference BLAS --
// multiply and add rounded separately (no contraction), netlib ddot / dgemv loop order, dpotf2 + dtrti2 + dlauu2 for the
// inverse, the per-fit BASIS_PHI cache rebuilt by design sweeps, the reference's visiting order for arg-max ties -- and is
// parallel only where outputs are independent (one lane per feature, per matrix element, per sample).  It is a diagnostic:
// 10-50x slower than the production path, Gaussian main effects only, selected by the environment, never by default.
//
// Each routine cites the reference lines it follows; the layout of the state is the production path's (GmWork) plus the
// arrays of GsExtra.
#pragma once
#include "blk.h"
#include "types.h"

#ifndef __HIP_DEVICE_COMPILE__
#define GS_FP
#else
#define GS_FP _Pragma("clang fp contract(off)")
#endif

#include "gs_log.h"   // the logarithm, correctly rounded (double-double): one last bit of a dML decides actions on tied twins

// a wave-uniform pointer in scalar registers on the device (loads through it become scalar loads); the pointer itself on the host
#ifdef PAREBEN_HOST_EMUL
#define GS_UNI(p) (p)
#else
#define GS_UNI(p) uni_ptr(p)
#endif

struct GsExtra {
    double *t;        // N   targets minus intercept                    :160-163
    double *e;        // N   residual / scratch
    double *phi;      // N   normalised column of the feature being added
    double *BP;       // cap x K  BASIS_PHI[l][i] = x_i . Phi_l / |x_i|, row l = active slot l   :1144-1201
    double *SigNew;   // ld x ld  (the production path's Gram-block cache buffer)
    double *D;        // ld x ld  D[j][i] = ddot(Phi_i, Phi_j) of the active columns (what FinalUpdate's PHI'PHI holds before the
                      //          factor beta, :1841-1876); kept across actions: an add appends the row it computes anyway
    double *w1, *w2;  // cap + 1 scratch
    long long *tim = nullptr;   // -DGS_TIMING (diagnostic build): ticks per phase, [0] full-stat features, [1] inverse, [2] actions, [3] design sweeps, [4] noise
};
#if defined(GS_TIMING) && !defined(PAREBEN_HOST_EMUL)
#define GS_T0() const long long gs_t0_ = (B.tid == 0) ? (long long)wall_clock64() : 0
#define GS_T1(k) do { if (B.tid == 0 && X.tim) X.tim[k] += (long long)wall_clock64() - gs_t0_; } while (0)
#else
#define GS_T0() do {} while (0)
#define GS_T1(k) do {} while (0)
#endif

DEV double gs_bcast(const Blk &B, double v)            // thread 0's value to every thread
{
#ifdef PAREBEN_HOST_EMUL
    (void)B; return v;
#else
    __syncthreads();
    if (B.tid == 0) B.red[0] = v;
    __syncthreads();
    const double r = B.red[0];
    __syncthreads();
    return r;
#endif
}
DEV int gs_bcast_i(const Blk &B, int v)
{
#ifdef PAREBEN_HOST_EMUL
    (void)B; return v;
#else
    __syncthreads();
    if (B.tid == 0) B.ired[0] = v;
    __syncthreads();
    const int r = B.ired[0];
    __syncthreads();
    return r;
#endif
}

// netlib ddot, one running sum (the reference's F77_CALL(ddot) and its hand-written loops)
DEV double gs_dot(const double *a, const double *b, int n)
{
    GS_FP
    double s = 0;
    for (int i = 0; i < n; i++) s = s + a[i] * b[i];
    return s;
}
// Phi_l[h]: the active column's normalised value, formed as the reference forms PHI (dcopy + dscal with 1/Scales, :517-520)
DEV double gs_phi(const FoldDev &F, int u, int N, int h) { GS_FP return F.X[(size_t)u * N + h] * F.rscale[u]; }
DEV double gs_dot_phi(const FoldDev &F, int ui, int uj, int N)          // ddot(Phi_i, Phi_j)
{
    GS_FP
    const double *xi = F.X + (size_t)ui * N, *xj = F.X + (size_t)uj * N;
    const double ri = F.rscale[ui], rj = F.rscale[uj];
    double s = 0;
    for (int h = 0; h < N; h++) s = s + (xi[h] * ri) * (xj[h] * rj);
    return s;
}
DEV double gs_dot_phi_vec(const FoldDev &F, int u, const double *v, int N)   // ddot(Phi_u, v)
{
    GS_FP
    const double *x = F.X + (size_t)u * N;
    const double r = F.rscale[u];
    double s = 0;
    for (int h = 0; h < N; h++) s = s + (x[h] * r) * v[h];
    return s;
}
// unbiased variance, :1826-1838 (thread 0, broadcast)
DEV double gs_var(const Blk &B, const double *v, int n)
{
    GS_FP
    double r = 0;
    if (B.tid == 0) {
        double m = 0, s = 0;
        for (int i = 0; i < n; i++) m = m + v[i];
        m = m / n;
        for (int i = 0; i < n; i++) { const double d = v[i] - m; s = s + d * d; }
        r = s / (n - 1);
    }
    return gs_bcast(B, r);
}

// BASIS_PHI row of the column `phi` (:1608-1617 / :1155-1163): row[i] = (sum_h phi[h] * x_i[h]) / |x_i|, phi first
DEV void gs_bp_row(const Blk &B, const FoldDev &F, int K, int N, const double *phi, double *row, bool phi_first)
{
    GS_FP
    if (F.Xt) {                                                  // same sums, the design read sample-major: lanes = features, whole lines
        const double *xt = GS_UNI(F.Xt), *ph = GS_UNI(phi);
        PAR(i, K) {
            double z = 0;
            if (phi_first) {
#pragma unroll 8
                for (int h = 0; h < N; h++) z = z + ph[h] * xt[(size_t)h * K + i];
            } else {
#pragma unroll 8
                for (int h = 0; h < N; h++) z = z + xt[(size_t)h * K + i] * ph[h];
            }
            row[i] = z / F.scale[i];
        }
        return;
    }
    PAR(i, K) {
        const double *x = F.X + (size_t)i * N;
        double z = 0;
        if (phi_first) for (int h = 0; h < N; h++) z = z + phi[h] * x[h];
        else for (int h = 0; h < N; h++) z = z + x[h] * phi[h];
        row[i] = z / F.scale[i];
    }
}

// S_out / Q_out of the active features, :1328-1338 and :666-671
DEV void gs_refresh_out(const Blk &B, const GmWork &W, int K)
{
    GS_FP
    PAR(i, K) {
        const double s = W.Sin[i], q = W.Qin[i];
        const int l = W.upos[i];
        if (l >= 0) { const double a = W.A[l]; W.Sout[i] = a * s / (a - s); W.Qout[i] = a * q / (a - s); }
        else { W.Sout[i] = s; W.Qout[i] = q; }
    }
    blk_sync(B);
}

// CacheBP :1144-1201: every active row of BASIS_PHI and BASIS_Targets
DEV void gs_cache(const Blk &B, const FoldDev &F, const GmWork &W, const GsExtra &X, int K, int M)
{
    GS_FP
    const int N = F.N;
    if (F.Xt) {                                                  // same sums, the design read sample-major
        const double *xt = GS_UNI(F.Xt), *tt = GS_UNI(X.t);
        const int LB = 4;
        if (B.pool_n >= LB * N) {                                // four Phi columns at a time staged in LDS: each x_i[h] load feeds four sums
            double *lp = B.pool;
            for (int l0 = 0; l0 < M; l0 += LB) {
                const int nb = M - l0 < LB ? M - l0 : LB;
                blk_sync(B);
                for (int e = B.tid; e < nb * N; e += B.nthr) {
                    const int q = e / N, h = e - q * N, u = W.used[l0 + q];
                    lp[e] = F.X[(size_t)u * N + h] * F.rscale[u];
                }
                blk_sync(B);
                PAR(i, K) {
                    double z[4] = {0, 0, 0, 0};
#pragma unroll 4
                    for (int h = 0; h < N; h++) {
                        const double x = xt[(size_t)h * K + i];
#pragma unroll
                        for (int q = 0; q < 4; q++) if (q < nb) z[q] = z[q] + lp[q * N + h] * x;
                    }
                    const double sc = F.scale[i];
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q < nb) X.BP[(size_t)(l0 + q) * K + i] = z[q] / sc;
                }
            }
            blk_sync(B);
            PAR(i, K) {
                double zt = 0;
#pragma unroll 8
                for (int h = 0; h < N; h++) zt = zt + xt[(size_t)h * K + i] * tt[h];
                W.bt[i] = zt / F.scale[i];
            }
            blk_sync(B);
            return;
        }
        PAR(i, K) {
            const double sc = F.scale[i];
            for (int l = 0; l < M; l++) {
                const int u = W.used[l];
                const double *xu = F.X + (size_t)u * N;
                const double r = F.rscale[u];
                double z = 0;
#pragma unroll 8
                for (int h = 0; h < N; h++) z = z + (xu[h] * r) * xt[(size_t)h * K + i];
                X.BP[(size_t)l * K + i] = z / sc;
            }
            double zt = 0;
#pragma unroll 8
            for (int h = 0; h < N; h++) zt = zt + xt[(size_t)h * K + i] * tt[h];
            W.bt[i] = zt / sc;
        }
        blk_sync(B);
        return;
    }
    PAR(i, K) {
        const double *x = F.X + (size_t)i * N;
        const double sc = F.scale[i];
        for (int l = 0; l < M; l++) {
            const int u = W.used[l];
            const double *xu = F.X + (size_t)u * N;
            const double r = F.rscale[u];
            double z = 0;
            for (int h = 0; h < N; h++) z = z + (xu[h] * r) * x[h];
            X.BP[(size_t)l * K + i] = z / sc;
        }
        double zt = 0;
        for (int h = 0; h < N; h++) zt = zt + x[h] * X.t[h];
        W.bt[i] = zt / sc;
    }
    blk_sync(B);
}

// mu = beta * Sigma * (Phi' t) in the column-sweep order of a reference dgemv('N') (:1266-1283, :1896-1912)
DEV void gs_mu(const Blk &B, const FoldDev &F, const GmWork &W, const GsExtra &X, int M, double beta)
{
    GS_FP
    const int N = F.N, ld = W.ld;
    GS_T0();
    PAR(l, M) X.w1[l] = gs_dot_phi_vec(F, W.used[l], X.t, N);
    blk_sync(B);
    GS_T1(6);
    PAR(i, M) {
        double a = 0;
        for (int j = 0; j < M; j++) a = a + X.w1[j] * W.Sig[(size_t)j * ld + i];
        W.mu[i] = a * beta;
    }
    blk_sync(B);
}

// FullStat :1209-1341 (Q3: gamma[0] is left alone)
DEV void gs_fullstat(const Blk &B, const FoldDev &F, const GmWork &W, const GsExtra &X, int K, GmScalars &S, bool very_first)
{
    GS_FP
    const int M = S.M, ld = W.ld, N = F.N;
    const double beta = S.beta;
    if (very_first) {
        if (B.tid == 0) {
            W.H[0] = gs_dot_phi(F, W.used[0], W.used[0], N) * beta + W.A[0];
            W.Sig[0] = 1 / W.H[0];
        }
        blk_sync(B);
    }
    gs_mu(B, F, W, X, M, beta);
    PAR(i, M) if (i >= 1) W.gam[i] = 1 - W.Sig[(size_t)i * ld + i] * W.A[i];
    blk_sync(B);
    // a_j = sum_p b_p Sigma[j][p] (p ascending), quad = sum_j a_j b_j (j ascending): eight a_j at a time share every b_p load
    // (the sums themselves are the reference's; only independent ones are interleaved).  On the device the eight Sigma rows of
    // a block are staged in LDS (every lane reads the same element: a broadcast) and the blocks run in the outer loop, the
    // running quad of a feature parked in S_in between blocks.
    const double *Sg = GS_UNI(W.Sig);
    const double *BPu = GS_UNI(X.BP);
    GS_T0();
    const bool staged = B.pool_n >= 8 * M && M >= 8;
    if (staged) {
        double *ls = B.pool;
        PAR(i, K) W.Sin[i] = 0;
        for (int j0 = 0; j0 < M; j0 += 8) {
            const int nb = M - j0 < 8 ? M - j0 : 8;
            blk_sync(B);
            for (int e = B.tid; e < nb * M; e += B.nthr) { const int r = e / M, p = e - r * M; ls[e] = Sg[(size_t)(j0 + r) * ld + p]; }
            blk_sync(B);
            PAR(i, K) {
                const double *bi = BPu + i;
                double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (nb == 8) {
#pragma unroll 4
                    for (int p = 0; p < M; p++) {
                        const double b = bi[(size_t)p * K];
#pragma unroll
                        for (int r = 0; r < 8; r++) a[r] = a[r] + b * ls[r * M + p];
                    }
                } else {
                    for (int p = 0; p < M; p++) {
                        const double b = bi[(size_t)p * K];
#pragma unroll
                        for (int r = 0; r < 8; r++) if (r < nb) a[r] = a[r] + b * ls[r * M + p];
                    }
                }
                double quad = W.Sin[i];
#pragma unroll
                for (int r = 0; r < 8; r++) if (r < nb) quad = quad + a[r] * bi[(size_t)(j0 + r) * K];
                W.Sin[i] = quad;
            }
        }
        blk_sync(B);
    }
    PAR(i, K) {
        double quad = 0;
        const double *bi = BPu + i;
        if (staged) quad = W.Sin[i];
        else {
            int j = 0;
            for (; j + 8 <= M; j += 8) {
                double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
                const double *s0 = Sg + (size_t)j * ld;
                for (int p = 0; p < M; p++) {
                    const double b = bi[(size_t)p * K];
                    a0 = a0 + b * s0[p];
                    a1 = a1 + b * s0[(size_t)ld + p];
                    a2 = a2 + b * s0[(size_t)2 * ld + p];
                    a3 = a3 + b * s0[(size_t)3 * ld + p];
                    a4 = a4 + b * s0[(size_t)4 * ld + p];
                    a5 = a5 + b * s0[(size_t)5 * ld + p];
                    a6 = a6 + b * s0[(size_t)6 * ld + p];
                    a7 = a7 + b * s0[(size_t)7 * ld + p];
                }
                quad = quad + a0 * bi[(size_t)j * K];
                quad = quad + a1 * bi[(size_t)(j + 1) * K];
                quad = quad + a2 * bi[(size_t)(j + 2) * K];
                quad = quad + a3 * bi[(size_t)(j + 3) * K];
                quad = quad + a4 * bi[(size_t)(j + 4) * K];
                quad = quad + a5 * bi[(size_t)(j + 5) * K];
                quad = quad + a6 * bi[(size_t)(j + 6) * K];
                quad = quad + a7 * bi[(size_t)(j + 7) * K];
            }
            for (; j < M; j++) {
                double a = 0;
                const double *sj = Sg + (size_t)j * ld;
                for (int p = 0; p < M; p++) a = a + bi[(size_t)p * K] * sj[p];
                quad = quad + a * bi[(size_t)j * K];
            }
        }
        double bm = 0;
        for (int p = 0; p < M; p++) bm = bm + X.BP[(size_t)p * K + i] * W.mu[p];
        W.Sin[i] = beta - beta * quad * beta;
        W.Qin[i] = beta * (W.bt[i] - bm);
    }
    blk_sync(B);
    GS_T1(0);
    gs_refresh_out(B, W, K);
    CNT(c.n_fullstat++; c.sum_m_full += M; c.sum_m2_full += (int64_t)M * M);
}

// dML and action of every feature, :1372-1582, with the reference's visiting order for ties (gm_delta_ml).  The first scan's
// quadratic coefficient is associated (so + 4 l1) + l2 for active features and (so + l2) + 4 l1 for inactive ones (:1395, :1465).
DEV int gs_delta_ml(const Blk &B, const GmWork &W, int K, int N, int M, double lambda, double alpha, double residual, double varY,
                    int iter, int i_iter, int *any_del_out, double *best)
{
    GS_FP
    const double l1 = lambda * alpha, l2 = lambda * (1 - alpha);
    int prio_add = 0, prio_del = 0;
    if (M < 10) { prio_add = 1; prio_del = 0; }
    if (M > 100 || M >= N || residual <= varY * 0.1) { prio_add = 0; prio_del = 1; }
    int my_add = 0, my_del = 0;
    double v1 = 0; int idx1 = 0x7fffffff;
    PAR(i, K) {
        const int l = W.upos[i];
        if (l == UP_LOST) { W.act[i] = ACT_NONE; continue; }
        const double so = W.Sout[i], qo = W.Qout[i];
        double d_ml = 0;
        int act = ACT_NONE;
        const double a = so - qo * qo + 2 * l1 + l2;
        const double bq = l >= 0 ? (so + l2) * (so + 4 * l1 + l2) : (so + l2) * (so + l2 + 4 * l1);
        const double g = 2 * l1 * (so + l2) * (so + l2);
        const double disc = bq * bq - 4 * a * g;
        if (a < 0 && disc > 0) {
            const double r = (-bq - sqrt(disc)) / (2 * a);
            const double L = (gs_log(r / (r + so + l2)) + qo * qo / (r + so + l2)) * 0.5 - l1 / r;
            if (L > 0) {
                W.aroot[i] = r + l2;
                if (l >= 0) {
                    act = ACT_REEST;
                    const double o = W.A[l] - l2;
                    d_ml = 0.5 * (gs_log(r * (o + so + l2) / (o * (r + so + l2))) + qo * qo * (1 / (r + so + l2) - 1 / (o + so + l2))) - l1 * (1 / r - 1 / o);
                } else { act = ACT_ADD; d_ml = L; my_add = 1; }
            }
        } else if (l >= 0 && M > 1) {
            my_del = 1;
            act = ACT_DEL;
            const double o = W.A[l] - l2;
            const double L = (gs_log(o / (o + so + l2)) + qo * qo / (o + so + l2)) * 0.5 - l1 / o;
            d_ml = -L;
        }
        W.act[i] = (signed char)act;
        W.dml[i] = d_ml;
        { const int key = l >= 0 ? l : M + i; if (d_ml > v1 || (d_ml == v1 && d_ml > 0 && key < idx1)) { v1 = d_ml; idx1 = key; } }
    }
    const int any_add = blk_or(B, my_add);
    const int any_del = blk_or(B, my_del);
    *any_del_out = any_del;
    bool rescanned = false;
    if ((any_add && prio_add) || (any_del && prio_del)) {
        PAR(i, K) {
            const int act = W.act[i];
            if (act == ACT_REEST) W.dml[i] = 0;
            else if (act == ACT_DEL) { if (any_add && prio_add && !prio_del) W.dml[i] = 0; }
            else if (act == ACT_ADD) { if (any_del && prio_del && !prio_add) W.dml[i] = 0; }
        }
        rescanned = true;
    }
    if ((!any_add && iter == 1 && i_iter < 10) || (!any_add && residual >= varY * 0.95)) {
        PAR(i, K) if (W.act[i] == ACT_DEL) W.dml[i] = 0;
        rescanned = true;
    }
    blk_sync(B);
    double v = 0; int idx = 0x7fffffff;
    if (rescanned) { PAR(i, K) { const double d = W.dml[i]; if (d > v) { v = d; idx = i; } } }
    else { v = v1; idx = idx1; }
    double bv; int bi;
    blk_argmax(B, v, idx, &bv, &bi);
    if (!rescanned && bv > 0) bi = bi < M ? W.used[bi] : bi - M;
    if (!(bv > 0)) { bv = 0; bi = 0; }
    *best = bv;
    return bi;
}

// In-place inverse of the SPD M x M matrix in Sig from its UPPER triangle exactly as dpotrf + dpotri ('U') of the reference
// BLAS/LAPACK compute it for these sizes' unblocked kernels: a = U'U (dpotf2), U <- U^-1 (dtrti2), a <- U U' (dlauu2), lower
// triangle mirrored (:1346-1369).  Element (r, c) of the column-major array is Sig[c * ld + r].  Parallel over independent
// elements only; every element's own sum runs in the sequential order of the scalar algorithm.  Returns 1 on a non-positive pivot.
DEV int gs_chol_inverse(const Blk &B, const GmWork &W, const GsExtra &X, int n)
{
    GS_FP
    const int ld = W.ld;
    double *a = W.Sig;
#define AT(r, c) a[(size_t)(c) * ld + (r)]
    // dpotf2.  Every element (r, c), r <= c, of U'U = A ends as  (A(r,c) - sum_{k<r} U(k,r) U(k,c)) / U(r,r)  with the
    // subtractions applied one at a time for k ascending (the scalar algorithm's loop); the diagonal likewise before its
    // square root.  The subtractions of different elements are independent, so they are applied eight pivot rows at a time:
    // the panel rows j0 .. j0+7 are finished first (row by row: the earlier panel rows subtracted in order, then the scaling),
    // kept in LDS, and every trailing element then takes its eight subtractions, k ascending, in one visit -- the matrix makes
    // n/8 round trips through memory instead of n.  Without LDS room (host build, huge n): one pivot row at a time.
    const int PB = 8;
    const bool panel = B.pool_n >= PB * n && n > PB;
    double *P = B.pool;                                           // [PB][n] panel rows (columns >= the row's index are valid)
    for (int j0 = 0; j0 < n; j0 += panel ? PB : 1) {
        const int nb = panel ? (n - j0 < PB ? n - j0 : PB) : 1;
        for (int q = 0; q < nb; q++) {                            // finish pivot row jj = j0 + q
            const int jj = j0 + q;
            blk_sync(B);
            if (panel) {
                for (int c = jj + B.tid; c < n; c += B.nthr) {
                    double v = AT(jj, c);
                    for (int k = 0; k < q; k++) v = v - P[k * n + jj] * P[k * n + c];
                    P[q * n + c] = v;
                }
                blk_sync(B);
                const double d0 = P[q * n + jj];
                if (!(d0 > 0)) return 1;
                const double d = sqrt(d0);
                blk_sync(B);
                for (int c = jj + B.tid; c < n; c += B.nthr) {
                    const double v = c == jj ? d : P[q * n + c] / d;
                    P[q * n + c] = v;
                    AT(jj, c) = v;
                }
            } else {
                double d = AT(jj, jj);
                if (!(d > 0)) return 1;
                d = sqrt(d);
                blk_sync(B);
                PAR(c, n) if (c > jj) AT(jj, c) = AT(jj, c) / d;
                if (B.tid == 0) AT(jj, jj) = d;
            }
        }
        blk_sync(B);
        const int t0 = j0 + nb;                                   // trailing block (t0 .. n-1)^2, upper triangle incl. diagonal
        if (panel) {
            for (int c = t0 + B.wave; c < n; c += B.nwave)
                for (int r = t0 + B.lane; r <= c; r += BLK_LANES) {
                    double v = AT(r, c);
#pragma unroll
                    for (int k = 0; k < PB; k++) if (k < nb) v = v - P[k * n + r] * P[k * n + c];
                    AT(r, c) = v;
                }
        } else {
            const int j = j0, m = n - 1 - j;
            for (int e = B.tid; e < m * m; e += B.nthr) {
                const int c = j + 1 + e / m, r = j + 1 + e % m;
                if (r <= c) AT(r, c) = AT(r, c) - AT(j, r) * AT(j, c);
            }
        }
        blk_sync(B);
    }
    // dtrti2: column j of U^-1 from the already inverted leading block T (upper, non-unit): x = T x by dtrmv, then x *= -1/U(j,j)
    for (int j = 0; j < n; j++) {
        blk_sync(B);
        if (B.tid == 0) AT(j, j) = 1.0 / AT(j, j);
        PAR(r, j) X.w1[r] = AT(r, j);                             // the original column (x)
        blk_sync(B);
        const double ajj = -AT(j, j);
        PAR(r, j) {
            // dtrmv 'U','N','N': for c ascending: if x[c] != 0 { x[r] += x[c] T(r, c) for r < c;  x[c] *= T(c, c) }
            // => x_new[r] = x[r] T(r, r) + sum_{c > r, x[c] != 0} x[c] T(r, c), in that order
            double s = X.w1[r] != 0 ? X.w1[r] * AT(r, r) : X.w1[r];
#pragma unroll 8
            for (int c = r + 1; c < j; c++) { const double xc = X.w1[c]; const double tv = AT(r, c); if (xc != 0) s = s + xc * tv; }
            X.w2[r] = s * ajj;
        }
        blk_sync(B);
        PAR(r, j) AT(r, j) = X.w2[r];
    }
    blk_sync(B);
    // dlauu2: A(i, i) = sum_{k >= i} U(i, k)^2 ;  A(r, i) = sum_{k > i} U(r, k) U(i, k) + U(i, i) U(r, i)  for r < i.  Row i of U
    // is read in columns k >= i only and every output column i is written from columns > i plus itself: all (r, i) pairs are
    // independent, results go to SigNew and are copied back.
    double *o = X.SigNew;
    for (int e = B.tid; e < n * n; e += B.nthr) {
        const int i = e / n, r = e % n;
        if (r > i) continue;
        const double aii = AT(i, i);
        double v;
        if (i < n - 1) {
            if (r == i) {
                double d = 0;
#pragma unroll 8
                for (int k = i; k < n; k++) d = d + AT(i, k) * AT(i, k);
                v = d;
            } else {
                double s = 0;
#pragma unroll 8
                for (int k = i + 1; k < n; k++) s = s + AT(r, k) * AT(i, k);
                v = s + aii * AT(r, i);
            }
        } else v = AT(r, i) * aii;
        o[(size_t)i * ld + r] = v;
    }
    blk_sync(B);
    for (int e = B.tid; e < n * n; e += B.nthr) {
        const int i = e / n, r = e % n;
        if (r <= i) { const double v = o[(size_t)i * ld + r]; AT(r, i) = v; AT(i, r) = v; }
    }
    blk_sync(B);
#undef AT
    return 0;
}

// FinalUpdate :1841-1921
DEV int gs_final_update(const Blk &B, const FoldDev &F, const GmWork &W, const GsExtra &X, GmScalars &S)
{
    GS_FP
    const int M = S.M, ld = W.ld, N = F.N;
    const double beta = S.beta;
    (void)N;
    for (int e = B.tid; e < M * M; e += B.nthr) {                 // PHI'PHI * beta + diag(A) from the kept dot products
        const int j = e / M, i = e % M;
        double h = X.D[(size_t)j * ld + i] * beta;
        if (i == j) h = h + W.A[i];
        W.H[(size_t)j * ld + i] = h;
        W.Sig[(size_t)j * ld + i] = h;
    }
    blk_sync(B);
    GS_T0();
    const int bad = gs_chol_inverse(B, W, X, M);
    GS_T1(1);
    if (bad) S.status |= ST_CHOLESKY;                             // Q11: the reference carries on regardless; so does the oracle
    gs_mu(B, F, W, X, M, beta);
    return 0;
}

// one call of the inner routine (:248-809).  *cs, *csy: column sums of C^-1 (:741-781, :172-187) in O(N M + M^2).
DEV int gs_inner(const Blk &B, const FoldDev &F, const GmWork &W, const GsExtra &X, int K, double lambda, double alpha,
                 GmScalars &S, int iter, double residual, double varY, double *cs, double *csy)
{
    GS_FP
    const int N = F.N, ld = W.ld;
    const bool first = iter <= 1;
    if (first) {                                               // :1003-1090, Q1
        S.M = 1;
        const double vt = gs_var(B, X.t, N);
        S.beta = 1 / (vt * 0.01 + 1e-10);
        PAR(i, K) W.upos[i] = UP_FREE;
        blk_sync(B);
        double a0 = 0;
        if (B.tid == 0) {
            W.used[0] = 0; W.upos[0] = 0;
            const double p = gs_dot_phi(F, 0, 0, N) * S.beta;
            const double q = gs_dot_phi_vec(F, 0, X.t, N) * S.beta;
            a0 = p * p / (q * q - p);
            if (a0 < 0) a0 = S.v.alpha_max;
            if (a0 > S.v.alpha_max) a0 = S.v.alpha_max;
            W.A[0] = a0;
        }
        blk_sync(B);
    } else {
        PAR(i, K) if (W.upos[i] == UP_LOST) W.upos[i] = UP_FREE;
        blk_sync(B);
    }
    PAR(i, W.cap + 1) W.gam[i] = 0;
    blk_sync(B);
    const int initial = W.used[0];
    int ini_removed = first ? 0 : 1;
    { GS_T0(); gs_cache(B, F, W, X, K, S.M); GS_T1(5); }
    if (first) { if (B.tid == 0) X.D[0] = gs_dot_phi(F, W.used[0], W.used[0], N); blk_sync(B); }
    int i_iter = 0;
    gs_fullstat(B, F, W, X, K, S, iter == 1);

    int sel = ACT_NONE, jj = -1, n_todo = 0, last_it = 0;
    const int it_max = iter == 1 ? 10 : 100;
    while (!last_it) {
        i_iter++;
        CNT(c.n_inner++);
        double best; int any_del;
        GS_T0();
        int nu = gs_delta_ml(B, W, K, N, S.M, lambda, alpha, residual, varY, iter, i_iter, &any_del, &best);
        GS_T1(7);
        int worthwhile;
        if (sel == ACT_TERM && !ini_removed && S.M > 1) nu = -1;
        if (nu == -1 && ini_removed) { worthwhile = 0; sel = ACT_TERM; }
        else if (nu == -1 && !ini_removed && S.M > 1) {
            worthwhile = 1;
            nu = initial;
            if (B.tid == 0) { W.act[nu] = ACT_DEL; W.todo[0] = initial; }
            blk_sync(B);
            n_todo = 1; ini_removed = 1; sel = ACT_DEL;
        } else {
            worthwhile = 1;
            const int act_nu = W.act[nu];
            double cutoff = best * (act_nu == ACT_ADD ? S.v.n_add : 1.0);
            if (cutoff < S.v.ml_delta) cutoff = S.v.ml_delta;
            n_todo = gm_collect(B, W, K, cutoff);
            if (act_nu == ACT_DEL && n_todo > 1) n_todo = 1;
            if (n_todo == 0) worthwhile = 0;
        }
        if (!worthwhile) sel = ACT_TERM;
        unsigned long long *const tr = gm_trace_rec(S);
        if (tr) gm_trace_decision(B, W, K, S, tr, iter, i_iter, nu, best, worthwhile, n_todo);
        if (worthwhile) {
            GS_T0();
            for (int u = 0; u < n_todo; u++) {
                nu = W.todo[u];
                sel = W.act[nu];
                const double newA = W.aroot[nu];
                if (sel == ACT_REEST || sel == ACT_DEL) {
                    const int l = W.upos[nu];
                    if (l >= 0) jj = l;
                    else { S.status |= ST_STALE; if (jj < 0 || jj >= S.M) { S.status |= ST_ABORT; return 1; } }
                }
                if (sel == ACT_REEST && fabs(gs_log(newA) - gs_log(W.A[jj])) <= S.v.reest_tol && any_del == 0) sel = ACT_TERM;
                blk_sync(B);
                const int M = S.M;
                bool upd = false;
                if (sel == ACT_REEST) {                          // :553-596
                    CNT(c.n_reest++; c.sum_m_action += M);
                    const double oldA = W.A[jj];
                    const double dinv = 1.0 / (newA - oldA);
                    const double kappa = 1.0 / (W.Sig[(size_t)jj * ld + jj] + dinv);
                    const double mujj = W.mu[jj];
                    const double f = -mujj * kappa;
                    blk_sync(B);
                    if (B.tid == 0) W.A[jj] = newA;
                    PAR(i, M) X.w1[i] = W.Sig[(size_t)jj * ld + i];          // Sigma row jj before the update
                    blk_sync(B);
                    PAR(i, M) W.mu[i] = W.mu[i] + f * X.w1[i];
                    for (int e = B.tid; e < M * M; e += B.nthr) {
                        const int j = e / M, i = e % M;
                        X.SigNew[(size_t)j * ld + i] = W.Sig[(size_t)j * ld + i] - kappa * X.w1[i] * X.w1[j];
                    }
                    PAR(i, K) {
                        double a = 0;
                        for (int j = 0; j < M; j++) a = a + X.BP[(size_t)j * K + i] * X.w1[j];
                        const double ba = S.beta * a;
                        W.Sin[i] = W.Sin[i] + ba * ba * kappa;
                        W.Qin[i] = W.Qin[i] + S.beta * mujj * kappa * a;
                    }
                    upd = true;
                } else if (sel == ACT_ADD) {                      // :1585-1723 + :613-627
                    if (M + 1 > W.cap) { S.status |= ST_OVERFLOW | ST_ABORT; return 1; }
                    if (M + 1 > W.cap_flag) S.status |= ST_OVERFLOW;
                    CNT(c.n_add++; c.sum_m_action += M);
                    const double rnu = F.rscale[nu];
                    PAR(h, N) X.phi[h] = F.X[(size_t)nu * N + h] * rnu;
                    blk_sync(B);
                    double *row = X.BP + (size_t)M * K;
                    { GS_T0(); gs_bp_row(B, F, K, N, X.phi, row, false); blk_sync(B); GS_T1(3); }   // z = x_i[h] * Phi[h], :1610-1615
                    PAR(i, M) {                                               // tmp = beta PHI' phi; the dots also extend PHI'PHI
                        const double dpp = gs_dot_phi_vec(F, W.used[i], X.phi, N);
                        X.D[(size_t)M * ld + i] = dpp; X.D[(size_t)i * ld + M] = dpp;
                        X.w1[i] = dpp * S.beta;
                    }
                    if (B.tid == 0) X.D[(size_t)M * ld + M] = gs_dot(X.phi, X.phi, N);
                    blk_sync(B);
                    PAR(i, M) {                                               // tp = Sigma tmp, ddot over a column of Sigma
                        double a = 0;
                        const double *si = W.Sig + (size_t)i * ld;
                        for (int j = 0; j < M; j++) a = a + si[j] * X.w1[j];
                        X.w2[i] = a;
                    }
                    blk_sync(B);
                    const double sii = 1.0 / (newA + W.Sin[nu]);
                    const double mui = sii * W.Qin[nu];
                    blk_sync(B);
                    PAR(i, M) W.mu[i] = W.mu[i] + -mui * X.w2[i];
                    for (int e = B.tid; e < M * M; e += B.nthr) {
                        const int j = e / M, i = e % M;
                        const double si_i = X.w2[i] * -sii;
                        const double tau = -si_i * X.w2[j];
                        X.SigNew[(size_t)j * ld + i] = W.Sig[(size_t)j * ld + i] + tau;
                    }
                    PAR(i, M) {
                        const double si_i = X.w2[i] * -sii;
                        X.SigNew[(size_t)M * ld + i] = si_i;
                        X.SigNew[(size_t)i * ld + M] = si_i;
                    }
                    PAR(i, K) {
                        double a = 0;
                        for (int j = 0; j < M; j++) a = a + X.BP[(size_t)j * K + i] * X.w2[j];
                        const double mc = S.beta * row[i] - S.beta * a;
                        W.Sin[i] = W.Sin[i] - mc * mc * sii;
                        W.Qin[i] = W.Qin[i] - mui * mc;
                    }
                    blk_sync(B);
                    if (B.tid == 0) {
                        X.SigNew[(size_t)M * ld + M] = sii;
                        W.A[M] = newA; W.mu[M] = mui; W.used[M] = nu; W.upos[nu] = M;
                    }
                    S.M = M + 1;
                    upd = true;
                } else if (sel == ACT_DEL) {                      // :1725-1822 + :640-651
                    CNT(c.n_del++; c.sum_m_action += M);
                    const int last = M - 1;
                    const double sjj = W.Sig[(size_t)jj * ld + jj];
                    const int mujj = (int)W.mu[jj];              // Q2
                    const int gone = W.used[jj];
                    blk_sync(B);
                    PAR(i, M) X.w1[i] = W.Sig[(size_t)jj * ld + i];
                    blk_sync(B);
                    PAR(i, M) W.mu[i] = W.mu[i] - mujj * X.w1[i] / sjj;
                    // T = Sigma - (row jj / sjj) row jj', then slot `last` moves into jj
                    for (int e = B.tid; e < M * M; e += B.nthr) {
                        const int j = e / M, i = e % M;
                        X.SigNew[(size_t)j * ld + i] = W.Sig[(size_t)j * ld + i] - X.w1[i] / sjj * X.w1[j];
                    }
                    PAR(i, K) {
                        double a = 0;
                        for (int j = 0; j < M; j++) a = a + X.BP[(size_t)j * K + i] * X.w1[j];
                        const double ba = S.beta * a;
                        W.Sin[i] = W.Sin[i] + ba * ba / sjj;
                        W.Qin[i] = W.Qin[i] + S.beta * a * mujj / sjj;
                    }
                    blk_sync(B);
                    if (jj != last) {
                        PAR(i, M) { X.w1[i] = X.SigNew[(size_t)last * ld + i]; X.w2[i] = X.SigNew[(size_t)i * ld + last]; }
                        blk_sync(B);
                        PAR(i, last) if (i != jj) { X.SigNew[(size_t)jj * ld + i] = X.w1[i]; X.SigNew[(size_t)i * ld + jj] = X.w2[i]; }
                        PAR(i, K) X.BP[(size_t)jj * K + i] = X.BP[(size_t)last * K + i];
                        PAR(i, M) { X.e[i] = X.D[(size_t)last * ld + i]; }   // PHI'PHI: row / column `last` into jj (symmetric bit for bit)
                        blk_sync(B);
                        PAR(i, last) if (i != jj) { X.D[(size_t)jj * ld + i] = X.e[i]; X.D[(size_t)i * ld + jj] = X.e[i]; }
                        if (B.tid == 0) X.D[(size_t)jj * ld + jj] = X.e[last];
                        blk_sync(B);
                        if (B.tid == 0) {
                            X.SigNew[(size_t)jj * ld + jj] = X.w1[last];
                            W.A[jj] = W.A[last]; W.mu[jj] = W.mu[last];
                            W.used[jj] = W.used[last]; W.upos[W.used[last]] = jj;
                        }
                    }
                    if (B.tid == 0) W.upos[gone] = (gone == nu) ? UP_FREE : UP_LOST;
                    S.M = last;
                    upd = true;
                }
                if (upd) {
                    blk_sync(B);
                    const int Mn = S.M;
                    for (int e = B.tid; e < Mn * Mn; e += B.nthr) { const int j = e / Mn, i = e % Mn; W.Sig[(size_t)j * ld + i] = X.SigNew[(size_t)j * ld + i]; }
                    blk_sync(B);
                    gs_refresh_out(B, W, K);
                    PAR(i, Mn) W.gam[i] = 1 - W.A[i] * W.Sig[(size_t)i * ld + i];
                    blk_sync(B);
                    CNT(if (S.M > c.m_max) c.m_max = S.M);
                }
            }
            GS_T1(2);
        }
        if (sel == ACT_TERM || i_iter <= 10 || i_iter % 5 == 0 || n_todo >= 2) {   // :685-729
            const int M = S.M;
            GS_T0();
            PAR(h, N) {
                double ev = 0;
                for (int j = 0; j < M; j++) ev = ev + W.mu[j] * gs_phi(F, W.used[j], N, h);
                X.e[h] = X.t[h] + -1.0 * ev;
            }
            blk_sync(B);
            double nb = 0;
            if (B.tid == 0) {
                const double ee = gs_dot(X.e, X.e, N);
                double gsum = 0;
                for (int i = 0; i < M; i++) gsum = gsum + W.gam[i];
                nb = (N - gsum) / ee;
            }
            nb = gs_bcast(B, nb);
            const double vt = gs_var(B, X.t, N);
            if (nb > 1e6 / vt) nb = 1e6 / vt;
            const double beta_old = S.beta;
            S.beta = nb;
            const double dlb = gs_log(nb) - gs_log(beta_old);
            GS_T1(4);
            if (fabs(dlb) > 1e-6) {
                gs_final_update(B, F, W, X, S);
                if (sel != ACT_TERM) gs_fullstat(B, F, W, X, K, S, false);
            }
        }
        if (tr) gm_trace_state(B, W, K, S, tr, sel);
        if (sel == ACT_TERM && ini_removed) last_it = 1;
        if ((i_iter == it_max && S.M == 1) || i_iter > it_max) last_it = 1;
        if (i_iter == it_max) sel = ACT_TERM;
    }
    {   // column sums of C^-1
        const int M = S.M;
        PAR(l, M) {
            double a = 0;
            const int u = W.used[l];
            for (int h = 0; h < N; h++) a += gs_phi(F, u, N, h);
            X.w1[l] = a;
        }
        blk_sync(B);
        PAR(i, M) {
            double a = 0;
            for (int j = 0; j < M; j++) a += X.w1[j] * W.Sig[(size_t)j * ld + i];
            X.w2[i] = a;
        }
        blk_sync(B);
        PAR(h, N) {
            double ev = 0;
            for (int j = 0; j < M; j++) ev += X.w2[j] * gs_phi(F, W.used[j], N, h);
            X.e[h] = ev;
        }
        blk_sync(B);
        double a = 0, b = 0;
        if (B.tid == 0) {
            const double beta = S.beta, b2 = beta * beta;
            for (int h = 0; h < N; h++) { const double c = beta - b2 * X.e[h]; a = a + c; b = b + c * F.y[h]; }
        }
        *cs = gs_bcast(B, a);
        *csy = gs_bcast(B, b);
    }
    return 0;
}

// the whole fit, :55-242
DEV void gs_fit(const Blk &B, const FoldDev &F, const GmWork &W, const GsExtra &X, int K, double lambda, double alpha, GmScalars &S)
{
    GS_FP
    const int N = F.N;
    S.status = 0; S.M = 1; S.gc_ok = 0;
    CNT(c = FitCounters{});
    double b = 0, varT = 0;
    if (B.tid == 0) { for (int i = 0; i < N; i++) b = b + 1.0 * F.y[i]; b = b / N; }
    b = gs_bcast(B, b);
    varT = gs_var(B, F.y, N);
    double residvar = 1e10, err = 1000, vk = 1e-30, vk0;
    int iter = 0;
    while (iter < 100 && err > 1e-8 && residvar >= varT * 0.01) {
        iter++;
        vk0 = vk;
        PAR(h, N) X.t[h] = -b + 1.0 * F.y[h];
        blk_sync(B);
        S.b = b;
        double cs, csy;
        if (gs_inner(B, F, W, X, K, lambda, alpha, S, iter, residvar, varT, &cs, &csy)) break;
        b = csy / (cs + S.v.b_eps);
        double v = 0;
        if (B.tid == 0) for (int i = 0; i < S.M; i++) v += W.A[i];
        vk = gs_bcast(B, v);
        err = fabs(vk - vk0) / S.M;
        residvar = 1 / (S.beta + 1e-10);
        if (S.outer_log && B.tid == 0) { double *o = S.outer_log + 3 * (iter - 1); o[0] = err; o[1] = b; o[2] = residvar; }
    }
    S.b = b;
    CNT(c.n_outer = iter; c.m_final = S.M; if (S.M > c.m_max) c.m_max = S.M; c.status = S.status);
    blk_sync(B);
}

// fold score, R/GetModelError.R:7-32 as R evaluates it: the kept rows of the weight table in feature order, one running sum per sample
DEV double gs_fold_sse(const Blk &B, const FoldDev &F, const GmWork &W, const GsExtra &X, const GmScalars &S, int K)
{
    GS_FP
    const int nte = F.nte;
    PAR(h, nte) {
        double pred = 0;
        for (int i = 0; i < K; i++) {
            const int l = W.upos[i];
            if (l < 0) continue;
            const double w = W.mu[l] / F.scale[i];
            if (w != 0) pred = pred + F.Xte[(size_t)i * nte + h] * w;
        }
        X.e[h] = pred;
    }
    blk_sync(B);
    double sse = 0;
    if (B.tid == 0) for (int h = 0; h < nte; h++) { const double r = F.yte[h] - (S.b + X.e[h]); sse = sse + r * r; }
    return gs_bcast(B, sse);
}
