// gm_host.h -- TEST INFRASTRUCTURE ONLY.  Plain-loop implementations of the phases of a Gaussian fit, with the
// signatures of pareben_amd/csrc/gm_dev.h, so that tests/emul/gm_emul.cpp can step the control flow of gm_fit.h on
// the CPU (one "thread" per workgroup) against the oracle.  Never part of the shipped library; nothing falls back to it.
#pragma once
#include <stdio.h>

#ifdef PAREBEN_TRACE
#define GM_TRACE(...) fprintf(stderr, __VA_ARGS__)
#else
#define GM_TRACE(...)
#endif
#define PH_BEGIN() do {} while (0)
#define PH_END(k) do {} while (0)
#define PHX_BEGIN(v) do {} while (0)
#define PHX_END(v, k) do {} while (0)
#define GM_FS_CLOCK_BEGIN() do {} while (0)
#define GM_FS_CLOCK_END() do {} while (0)
typedef double *lptr_d;
typedef const double *gptr_cd;
typedef const int *gptr_ci;
DEV gptr_cd as_global(const double *p) { return p; }
DEV gptr_ci as_global(const int *p) { return p; }
#define FS_MAX_M 2048

// MainEff.c:1291-1319 in the reference's own loop order
DEVNI void gm_fullstat_pass(const Blk &B, const FoldDev &F, const GmWork &W, int K, int M, double beta, GmScalars &S)
{
    (void)S;
    const int ld = W.ld;
    for (int i = 0; i < K; i++) {
        double quad = 0, bm = 0;
        for (int j = 0; j < M; j++) {
            double a = 0;
            for (int p = 0; p < M; p++) a += F.G[(size_t)W.rowid[p] * K + i] * W.Sig[(size_t)j * ld + p];
            double bj = F.G[(size_t)W.rowid[j] * K + i];
            quad += a * bj;
            bm += bj * W.mu[j];
        }
        W.Sin[i] = beta - beta * quad * beta;
        W.Qin[i] = beta * (W.bt[i] - bm);
    }
    blk_sync(B);
}
DEV void gm_fs_pad(const Blk &, const GmWork &, int) {}
DEV void gm_fs_count(const Blk &, GmScalars &, int, int) {}

// a[i] = sum_j G[used[j], i] * vec[j] for all features, fused with the S_in/Q_in update that
// consumes it.  `rid`: Gram row id of the new feature (mode 1), else -1.  del_jj / del_row: see gm_sq_stage.
DEVNI void gm_sq_update(const Blk &B, const FoldDev &F, const GmWork &W, int K, int M, const double *vec,
                        int mode, double beta, double c1, double c2, int rid, GmScalars &S, int del_jj = -1, int del_row = -1)
{
    const double *newrow = rid >= 0 ? F.G + (size_t)rid * K : nullptr;
    CNT(c.sum_m_swept += M);
    PAR(i, K) {
        double a = 0;
        for (int j = 0; j < M; j++) {
            int r = W.rowid[j];
            if (del_jj >= 0) { if (j == del_jj) r = del_row; else if (j == M - 1) r = W.rowid[del_jj]; }
            a += F.G[(size_t)r * K + i] * vec[j];
        }
        gm_sq_apply(W, mode, beta, c1, c2, newrow, i, a);
    }
    blk_sync(B);
}

template <class FA, class FB>
DEV void gm_rank1(const Blk &B, const GmWork &W, int M, double *scr, FA fa, FB fb)
{
    const int ld = W.ld;
    (void)scr;
    for (int j = 0; j < M; j++) {
        const double f = fa(j);
        for (int i = 0; i < M; i++) W.Sig[(size_t)j * ld + i] += f * fb(i);
    }
}

DEV double wave_sum(double v) { return v; }
#define ROW_LOAD(p) (*(p))
#define ROW_CAS(p, e, d) (*(p) == (e) ? (*(p) = (d), true) : ((e) = *(p), false))
#define ROW_STORE(p, v) (*(p) = (v))
#define ROW_FETCH_ADD(p, v) ((*(p) += (v)) - (v))
DEVNI int gm_row(const Blk &B, const FoldDev &F, const GmWork &W, int K, int u)
{
    if (!F.lazy) return u;
    enum { R_OWNER = -3, R_PRIVATE = -4 };
    blk_sync(B);
    if (B.tid == 0) {
        int *st = F.slot_of + u;
        int s = ROW_LOAD(st);
        if (s == -1) {
            int expect = -1;
            if (ROW_CAS(st, expect, -2)) s = R_OWNER; else s = expect;
        }
        if (s == -2 || s == -1) s = R_PRIVATE;                  // timed out / the owner found the pool full
        int my = -1;
        if (s == R_OWNER) {
            if (ROW_LOAD(F.pool_next) < F.pool_rows) my = ROW_FETCH_ADD(F.pool_next, 1);
            if (my >= 0 && my < F.pool_rows) my += F.pool_base;
            else { my = -1; ROW_STORE(st, -1); s = R_PRIVATE; }    // pool exhausted
        }
        if (s == R_PRIVATE && W.pfree[0] > 0) my = W.pfree[W.pfree[0]--];   // one of this fit's own rows
        B.ired[0] = s;
        B.ired[1] = my;
    }
    blk_sync(B);
    const int s = B.ired[0], my = B.ired[1];
    blk_sync(B);
    if (s >= 0) return s;
    if (my < 0) return -1;
    const int N = F.N;
    double *row = const_cast<double *>(F.G) + (size_t)my * K;
    const double *xu = F.X + (size_t)u * N;
    // PHI as the reference forms it: a main-effect column times the reciprocal of its norm (:517-520), a pair column divided by it (Full2.c:544)
    const double su = u < F.n_main ? 1.0 : F.scale[u], ru = u < F.n_main ? F.rscale[u] : 1.0;
    const bool in_lds = N <= B.pool_n;
    if (in_lds) {
        PAR(h, N) B.pool[h] = xu[h] * ru / su;                  // one of the two factors is exactly 1
        blk_sync(B);
    }
    for (int i = B.wave; i < K; i += B.nwave) {
        const double *xi = F.X + (size_t)i * N;
        double a = 0;
        for (int h = B.lane; h < N; h += BLK_LANES) a += xi[h] * (xu[h] * ru / su);
        if (B.lane == 0) row[i] = a / F.scale[i];
    }
    blk_sync(B);
    if (s == R_OWNER && B.tid == 0) {
        ROW_STORE(F.slot_of + u, my);
    }
    blk_sync(B);
    return my;
}

#define MV_LDS(M, nwave, CW) ((((M) + 15) & ~15) + (nwave) * 64 * ((CW) + 1))
template <int CW>
DEV void gm_sigma_matvec(const Blk &B, const GmWork &W, int M, const double *v, double *out, double *scr, int scr_n, lptr_d out_lds)
{
    const int ld = W.ld;
    (void)scr; (void)scr_n; (void)out_lds;
    PAR(i, M) {
        double a = 0;
        for (int j = 0; j < M; j++) a += W.Sig[(size_t)i * ld + j] * v[j];
        out[i] = a;
    }
}

DEV void gm_gc_add(const Blk &, const FoldDev &, const GmWork &, int, const GmScalars &, int, int, int) {}
// runs of adds are a device optimisation: every add takes the single-action path here
DEV int gm_add_run(const Blk &, const FoldDev &, const GmWork &, int, GmScalars &, int, int, bool) { return 0; }
DEV void gm_flush_add_run(const Blk &, const FoldDev &, const GmWork &, int, GmScalars &, int, int, double) {}
DEV int gm_spd_inverse(const Blk &B, const GmWork &W, int M, long long *phx = nullptr, int pair = 1) { (void)phx; (void)pair; return gm_spd_inverse_scalar(B, W, M); }

// H = beta G[used, used] + diag(A) into W.H and W.Sig (MainEff.c:1841-1876)
DEV void gm_hessian_build(const Blk &B, const FoldDev &F, const GmWork &W, int K, GmScalars &S)
{
    const int M = S.M, ld = W.ld;
    const double beta = S.beta;
    for (int j = 0; j < M; j++) {
        const int uj = W.used[j];
        for (int i = 0; i < M; i++) {
            const int ui = W.used[i];
            // Phi_i.Phi_j from the Gram matrix; one triangle so that H is exactly symmetric
            double h = (i <= j ? F.G[(size_t)W.rowid[i] * K + uj] : F.G[(size_t)W.rowid[j] * K + ui]) * beta;
            if (i == j) h += W.A[i];
            W.H[(size_t)j * ld + i] = h;
            W.Sig[(size_t)j * ld + i] = h;
        }
    }
    (void)K;
}

DEV void gm_inverse_count(const Blk &, GmScalars &, int) {}
DEV void gm_mu_update(const Blk &B, const GmWork &W, int M, double beta)
{
    const int ld = W.ld;
    PAR(i, M) {
        double a = 0;
        for (int j = 0; j < M; j++) a += W.v1[j] * W.Sig[(size_t)j * ld + i];
        W.mu[i] = a * beta;
    }
}

DEV void gm_stage_model(const Blk &, const FoldDev &, const GmWork &, int, const double *) {}
DEV double gm_model_at(const Blk &, const FoldDev &F, const GmWork &W, int M, const double *vec, int N, int h)
{
    double v = 0;
    for (int j = 0; j < M; j++) { const int uj = W.used[j]; v += vec[j] * (F.X[(size_t)uj * N + h] * F.rscale[uj]); }
    return v;
}
DEV void gm_model_at2(const Blk &B, const FoldDev &F, const GmWork &W, int M, const double *vec, int N, int h0, int h1, double &v0, double &v1)
{
    v0 = gm_model_at(B, F, W, M, vec, N, h0);
    v1 = gm_model_at(B, F, W, M, vec, N, h1);
}

DEV unsigned long long blk_xor64(const Blk &B, unsigned long long v)
{
    (void)B;
    return v;
}

