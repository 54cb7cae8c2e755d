// bm_fit.h -- one binomial (logistic) EBEN fit executed by ONE workgroup: main effects
// (ElasticNetBinaryNEmainEff.c, "Bm") or main effects + pairwise epistasis (ElasticNetBinaryNeFull.c, "Bf").
//
// Bf runs the same code on the expanded design (expand_kernel: the K main columns followed by the pair
// columns x_i*x_j in the reference's order) with the rule set in which NeFull.c differs from NEmainEff.c,
// each cited where `S.v.epis` / `W.phi_div` switches it: block cut-off 0.99 (:254), phi = column / scale by
// division (:437-450, :755), Newton step with y clamped to [1e-5, 1-1e-5] and weights < 1e-5 -> 1e-3,
// > 1e5 -> 1e3 (:1042-1048) that stops when ALL M gradient entries are small (:1085-1099), Sigma downdate of a
// delete associated as S - (s_i/s_jj) s_j (:1606), delete-priority only above 100 bases (:1805-1809), outer
// stopping sum over the M-1 precisions (:133-134), capacity bMax = 2K bases from the R wrapper
// (EBelasticNet.Binomial.R:7-9; NeFull.c:549-553).  The reference regenerates a pair column inside every
// sweep and associates its products as ((x_i*phi)*w)*x_j etc.; on the materialised column the same products
// round differently in the last bit for non-integer designs (exactly equal for genotype codes -1/0/1).
//
// What it computes is what EBEN_orig/src/ElasticNetBinaryNEmainEff.c computes for one
// (training fold, alpha, lambda): the outer loop (:329-344) around the inner
// add / re-estimate / delete ascent (:397-827) with a Laplace (IRLS) posterior mode
// (fEBCatPostModeBmNeEN :1808-2010) after every block of actions.
//
// The per-sample IRLS weights w = y(1-y) change at every mode update, so unlike the Gaussian
// kernel nothing can be cached as a Gram matrix: every action needs the weighted cross products
// BP[i][p] = x_i' diag(w) Phi_p / |x_i| of all K features with the M model columns.  They are
// computed once per (mode update / action) by the whole workgroup into a K x M scratch matrix in
// HBM -- a small GEMM on the FP64 matrix cores (bm_weighted_rows) -- and then reused by the
// per-feature quadratic forms (bm_quad_features, matrix cores too); the Newton step's Hessian
// Phi' diag(w) Phi is the third such product (bm_postmode).  Phi is never materialised: column 0
// is the intercept, column l+1 is the design column of used[l] times 1/|x|.
//
// Reference quirks kept: Q1 first basis = column 0, force-deleted once;
// Q12 the outer stopping sum covers M = N_used+1 precisions (one stale slot); Q15 add-priority
// never fires; the re-estimate S/Q update reads the already updated Sigma row; the Newton loop
// keeps the y of a rejected line search.
#pragma once
#include "blk.h"
#include "types.h"
#include "gm_fit.h"      // action codes, GmScalars, counters, gm_spd_inverse

struct BmWork {
    double *Sin, *Qin, *Sout, *Qout, *dml, *aroot, *bb;   // K each
    int *upos, *todo;                                      // K each
    signed char *act;                                      // K
    double *Sig, *H;                                       // ld x ld, column-major
    double *A, *mu, *g, *dmu, *mnew, *tmp, *tp, *v3, *v4;  // ld each
    int *used;                                             // ld
    double *w, *pm, *yv, *e, *bphi;                        // Nmax each
    double *BP;                                            // K x ld (row i = feature i)
    int cap, ld;                                           // cap = max model size M, ld = cap
    int phi_div;                                           // Bf: model column = x / |x| (division); Bm: x * (1/|x|)
    int bmax;                                              // Bf: most bases a model may hold (the R wrapper's bMax = 2K); Bm: unused (cap decides)
};

// design column u, normalised, at sample h: NEmainEff.c:644-646 multiplies by the reciprocal norm, NeFull.c:437-450 divides
DEV double bm_col(const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int N, int u, int h)
{
    const double x = F.X[(size_t)u * N + h];
    return W.phi_div ? x / F.scale[u] : x * F.rscale[u];
}
// model column p at sample h
#define BM_PHI(p, h) ((p) == 0 ? 1.0 : bm_col(F, W, N, W.used[(p) - 1], (h)))

// a GmWork view so the Gaussian kernel's SPD inverse can be reused on (Sig, ld)
DEV GmWork bm_as_gm(const BmWork &NOALIAS W)
{
    GmWork G{};
    G.Sig = W.Sig; G.H = W.H; G.v3 = W.v3; G.v4 = W.v4; G.cap = W.cap; G.ld = W.ld;
    return G;
}

// pm[h] = sum_p Phi_p[h] * mu[p], every term through BM_PHI
DEVNI void bm_phi_mu_plain(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int M, const double *mu, double *out)
{
    const int N = F.N;
    PAR(h, N) {
        double a = 0;
        for (int p = 0; p < M; p++) a += BM_PHI(p, h) * mu[p];
        out[h] = a;
    }
    blk_sync(B);
}

// y = sigmoid(pm); returns -sum t log y + (1-t) log(1-y)  (:2013-2032)
DEVNI double bm_data_error(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W)
{
    const int N = F.N;
    double part = 0;
    PAR(h, N) {
        const double y = 1 / (1 + exp(-W.pm[h]));
        W.yv[h] = y;
        const double t = F.y[h];
        if (y != 0) part -= t * log(y);
        if (y != 1) part -= (1 - t) * log(1 - y);
    }
    const double s = blk_sum(B, part);
    return s;
}

// ---- the phases: the three dense products and the small reductions whose implementation is the hardware mapping.  The
// shipped library gets bm_dev.h (FP64 matrix cores, LDS staging); the CPU test harness under tests/emul names its own
// header of plain loops with the same signatures.
#ifndef BM_PHASES_H
#define BM_PHASES_H "bm_dev.h"
#endif
#include BM_PHASES_H

// posterior mode, :1808-2010.  Leaves w, Sig (= H^-1), H from its last Hessian evaluation.
DEVNI int bm_postmode(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, GmScalars &NOALIAS S)
{
    const int N = F.N, M = S.M, ld = W.ld;
    const double step_min = 1.0 / 256.0;
    bm_phi_mu(B, F, W, M, W.mu, W.pm);
    double derr = bm_data_error(B, F, W);
    double rp = 0;
    PAR(i, M) if (i >= 1) rp += W.A[i - 1] * W.mu[i] * W.mu[i] / 2;
    double total = blk_sum(B, rp) + derr;
    for (int it = 0; it < 25; it++) {
        const double elog = total;
        double gp = 0, hp = 0;
        PHX2_BEGIN(pt0_);
        PAR(h, N) {
            double y = W.yv[h];
            if (S.v.epis) {                                    // NeFull.c:1042-1043
                if (y < 1e-5) y = 1e-5;
                if (y > (1 - 1e-5)) y = 1 - 1e-5;
            }
            const double e = F.y[h] - y;
            W.e[h] = e;
            double b = y * (1 - y);
            if (S.v.epis) { if (b < 1e-5) b = 1e-3; if (b > 1e5) b = 1e3; }       // NeFull.c:1047-1048
            else { if (b < 1e-10) b = 1e-5; if (b > 1e10) b = 1e5; }             // NEmainEff.c:1878-1883
            W.w[h] = b;
            gp += e; hp += b;
        }
        const double g0 = blk_sum(B, gp), h0 = blk_sum(B, hp);
        PHX2_END(pt0_, 16);
        PHX2_BEGIN(pt1_);
        bm_grad_hessian(B, F, W, M, N);                          // gradient entries 1.., Hessian Phi' diag(w) Phi + diag(A)
        if (B.tid == 0) { W.g[0] = g0; W.H[0] = h0; }
        blk_sync(B);
        PHX2_END(pt1_, 17);
        PHX2_BEGIN(pt2_);
        for (int j = B.wave; j < M; j += B.nwave)
            for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] = W.H[(size_t)j * ld + i];
        blk_sync(B);
        {
            const GmWork G = bm_as_gm(W);
            if (gm_spd_inverse(B, G, M)) return 1;
        }
        blk_sync(B);
        PHX2_END(pt2_, 18);
        int cp = 0;
        PAR(j, M) if ((j >= 1 || S.v.epis) && fabs(W.g[j]) < 1e-6) cp++;   // NeFull.c:1085-1099 counts the intercept's entry too
        const int cnt = blk_isum(B, cp);
        if (cnt == (S.v.epis ? M : M - 1)) break;
        PHX2_BEGIN(pt3_);
        PAR(k, M) {
            double a = 0;
            for (int L = 0; L < M; L++) a += W.g[L] * W.Sig[(size_t)L * ld + k];
            W.dmu[k] = a;
        }
        blk_sync(B);
        double step = 1;
        while (step > step_min) {
            PAR(j, M) W.mnew[j] = W.mu[j] + step * W.dmu[j];
            blk_sync(B);
            bm_phi_mu(B, F, W, M, W.mnew, W.pm);
            derr = bm_data_error(B, F, W);
            double rq = 0;
            PAR(j, M) if (j >= 1) rq += W.A[j - 1] * W.mnew[j] * W.mnew[j] / 2;
            total = derr + blk_sum(B, rq);
            if (total >= elog) step = step / 2;
            else {
                PAR(j, M) W.mu[j] = W.mnew[j];
                blk_sync(B);
                step = 0;
            }
        }
        PHX2_END(pt3_, 19);
#ifdef PAREBEN_PHASE_TIMERS
        if (B.tid == 0) S.ph[20]++;                              // Newton iterations
#endif
    }
    blk_sync(B);
    return 0;
}

// full statistics, :1633-1803
DEVNI int bm_fullstat(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, GmScalars &NOALIAS S)
{
    const int N = F.N, ld = W.ld;
    S.bp_ok = 0;                                               // the weights change: cached weighted rows are stale
    { PH_BEGIN(); const int bad = bm_postmode(B, F, W, S); PH_END(PH_INVERSE); if (bad) return 1; }
    const int M = S.M;
    PH_BEGIN();
    bm_phi_mu(B, F, W, M, W.mu, W.pm);
    PAR(h, N) { const double y = 1 / (1 + exp(-W.pm[h])); W.e[h] = F.y[h] - y; }
    blk_sync(B);
    const int have_stats = bm_weighted_rows(B, F, W, K, M, true, S.ph);
    S.bp_ok = M;
    PH_END(PH_FS_FEAT);
    // S_in = x_i' diag(w) x_i / |x_i|^2 - BP_i' Sigma BP_i ;  Q_in = x_i' e / |x_i|
    PHX2_BEGIN(ph_t1_);
    bm_feature_stats(B, F, W, K, N, have_stats);               // bb_i = x_i' diag(w) x_i, ze_i = x_i' e where the weighted-row pass did not deliver them
    blk_sync(B);
    bm_quad_features(B, F, W, K, M);
    PAR(i, K) {
        const double s = W.Sin[i], q = W.Qin[i];
        const int l = W.upos[i];
        if (l >= 0) { const double a = W.A[l]; W.Sout[i] = a * s / (a - s); W.Qout[i] = a * q / (a - s); }
        else { W.Sout[i] = s; W.Qout[i] = q; }
    }
    blk_sync(B);
    PHX2_END(ph_t1_, PH_FS_REST);
    CNT(c.n_fullstat++; c.sum_m_full += M; c.sum_m2_full += (int64_t)M * M);
    return 0;
}

// dML / action choice, :2063-2238 (Q15: only delete-priority can fire)
DEVNI int bm_delta_ml(const Blk &NOALIAS B, const BmWork &NOALIAS W, int K, int N, int NU, double lambda, double alpha,
                      int epis, int *any_del_out, double *best)
{
    const double l1 = lambda * alpha, l2 = lambda * (1 - alpha);
    int prio_add = 0, prio_del = 0;
    if (NU < 10) { prio_add = 1; prio_del = 0; }
    if (NU > 100 || (!epis && NU >= N)) { prio_add = 0; prio_del = 1; }   // NeFull.c:1805-1809 has no N clause
    int my_del = 0;
    PAR(i, K) {
        const int l = W.upos[i];
        if (l == UP_LOST) { W.act[i] = ACT_NONE; continue; }
        const double so = W.Sout[i], qo = W.Qout[i];
        double d_ml = 0;
        int act = ACT_NONE;
        const double a = so - qo * qo + 2 * l1 + l2;
        const double bq = (so + l2) * (so + 4 * l1 + l2);
        const double g = 2 * l1 * (so + l2) * (so + l2);
        const double disc = bq * bq - 4 * a * g;
        if (a < 0 && disc > 0) {
            const double r = (-bq - sqrt(disc)) / (2 * a);
            const double L = (log(r / (r + so + l2)) + qo * qo / (r + so + l2)) * 0.5 - l1 / r;
            if (L > 0) {
                W.aroot[i] = r + l2;
                if (l >= 0) {
                    act = ACT_REEST;
                    const double o = W.A[l] - l2;
                    d_ml = 0.5 * (log(r * (o + so + l2) / (o * (r + so + l2))) +
                                  qo * qo * (1 / (r + so + l2) - 1 / (o + so + l2))) -
                           l1 * (1 / r - 1 / o);
                } else { act = ACT_ADD; d_ml = L; }
            }
        } else if (l >= 0 && NU > 1) {
            my_del = 1;
            act = ACT_DEL;
            const double o = W.A[l] - l2;
            const double L = (log(o / (o + so + l2)) + qo * qo / (o + so + l2)) * 0.5 - l1 / o;
            d_ml = -L;
        }
        W.act[i] = (signed char)act;
        W.dml[i] = d_ml;
    }
    const int any_del = blk_or(B, my_del);
    *any_del_out = any_del;
    bool rescanned = false;
    if (any_del && prio_del) {
        PAR(i, K) {
            const int act = W.act[i];
            if (act == ACT_REEST) W.dml[i] = 0;
            else if (act == ACT_ADD) { if (!prio_add) W.dml[i] = 0; }
        }
        rescanned = true;
    }
    blk_sync(B);
    // ties: the reference's first scan walks the used list in slot order, then the unused one, and keeps the first strict
    // maximum (an active feature wins over an inactive one at bit-equal dML: see gm_delta_ml); a rescan runs in index order
    double v = 0; int idx = 0x7fffffff;
    PAR(i, K) {
        const int l = W.upos[i];
        if (!rescanned && l == UP_LOST) continue;
        const double d = W.dml[i];
        const int key = rescanned ? i : (l >= 0 ? l : NU + i);
        if (d > v || (d == v && d > 0 && key < idx)) { v = d; idx = key; }
    }
    double bv; int bi;
    blk_argmax(B, v, idx, &bv, &bi);
    if (!rescanned && bv > 0) bi = bi < NU ? W.used[bi] : bi - NU;
    if (!(bv > 0)) { bv = 0; bi = 0; }
    *best = bv;
    return bi;
}

DEVNI int bm_collect(const Blk &NOALIAS B, const BmWork &NOALIAS W, int K, double cutoff)
{
    int base = 0;
    for (int i0 = 0; i0 < K; i0 += B.nthr) {
        const int i = i0 + B.tid;
        const int f = (i < K && W.dml[i] >= cutoff) ? 1 : 0;
        int tot;
        const int off = blk_scan_excl(B, f, &tot);
        if (f) W.todo[base + off] = i;
        base += tot;
    }
    blk_sync(B);
    return base;
}

// add feature nu, :830-1003 + :701-711
DEVNI void bm_add(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, GmScalars &NOALIAS S, int nu, double newA)
{
    const int N = F.N, M = S.M, ld = W.ld, NU = M - 1;
    PAR(h, N) W.bphi[h] = W.w[h] * bm_col(F, W, N, nu, h);
    blk_sync(B);
    // bb[i] = x_i' (w .* phi) / |x_i| ; tmp[p] = Phi_p' (w .* phi)
    bm_add_products(B, F, W, K, M, N);
    blk_sync(B);
    PAR(i, M) {
        double a = 0;
        for (int j = 0; j < M; j++) a += W.Sig[(size_t)j * ld + i] * W.tmp[j];
        W.tp[i] = a;
    }
    const double sii = 1.0 / (newA + W.Sin[nu]);
    const double mui = sii * W.Qin[nu];
    blk_sync(B);
    // rows against the OLD model columns: still current from the last full-stat pass / action unless the
    // weights changed in between (they only change in the posterior-mode step)
    if (S.bp_ok != M) bm_weighted_rows(B, F, W, K, M);
    bm_rows_dot(B, W, K, M, W.tp, [&](int i, double t) {
        const double mc = W.bb[i] - t;
        W.Sin[i] = W.Sin[i] - mc * mc * sii;
        W.Qin[i] = W.Qin[i] - mui * mc;
        W.BP[(size_t)i * ld + M] = W.bb[i];                    // the new model column's weighted row entry
    });
    S.bp_ok = M + 1;
    PAR(i, M) W.mu[i] += -mui * W.tp[i];
    for (int j = B.wave; j < M; j += B.nwave) {
        const double f = sii * W.tp[j];
        for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] += f * W.tp[i];
    }
    PAR(i, M) {
        const double si = -sii * W.tp[i];
        W.Sig[(size_t)M * ld + i] = si;
        W.Sig[(size_t)i * ld + M] = si;
    }
    if (B.tid == 0) {
        W.Sig[(size_t)M * ld + M] = sii;
        W.A[NU] = newA;
        W.mu[M] = mui;
        W.used[NU] = nu;
        W.upos[nu] = NU;
    }
    S.M = M + 1;
    blk_sync(B);
}

// delete used slot jj (feature nu), :1010-1121 + :728-744
DEVNI void bm_delete(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, GmScalars &NOALIAS S, int jj, int nu)
{
    const int M = S.M, ld = W.ld, last = M - 1, j1 = jj + 1;
    PAR(i, M) W.tp[i] = W.Sig[(size_t)j1 * ld + i];
    blk_sync(B);
    const double sjj = W.tp[j1];
    const double mujj = W.mu[j1];
    const int gone = W.used[jj];
    if (S.bp_ok != M) bm_weighted_rows(B, F, W, K, M);
    bm_rows_dot(B, W, K, M, W.tp, [&](int i, double t) {
        W.Sin[i] = W.Sin[i] + t * t / sjj;
        W.Qin[i] = W.Qin[i] + t * mujj / sjj;
        W.BP[(size_t)i * ld + j1] = W.BP[(size_t)i * ld + last];   // the last model column moves into the freed slot
    });
    S.bp_ok = last;
    PAR(i, M) W.mu[i] = W.mu[i] - mujj * W.tp[i] / sjj;
    for (int j = B.wave; j < M; j += B.nwave) {
        const double vj = W.tp[j];
        if (S.v.epis) for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] -= W.tp[i] / sjj * vj;   // NeFull.c:1606
        else          for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] -= W.tp[i] * vj / sjj;   // NEmainEff.c:1069
    }
    blk_sync(B);
    if (j1 != last) {
        PAR(i, M) { W.v3[i] = W.Sig[(size_t)last * ld + i]; W.v4[i] = W.Sig[(size_t)i * ld + last]; }
        blk_sync(B);
        PAR(i, last) {
            if (i != j1) { W.Sig[(size_t)j1 * ld + i] = W.v3[i]; W.Sig[(size_t)i * ld + j1] = W.v4[i]; }
        }
        if (B.tid == 0) {
            W.Sig[(size_t)j1 * ld + j1] = W.v3[last];
            W.A[jj] = W.A[last - 1];
            W.mu[j1] = W.mu[last];
            W.used[jj] = W.used[last - 1];
            W.upos[W.used[last - 1]] = jj;
        }
    }
    if (B.tid == 0) W.upos[gone] = (gone == nu) ? UP_FREE : UP_LOST;
    S.M = last;
    blk_sync(B);
}

// re-estimate used slot jj, :1127-1203 (S/Q update reads the NEW Sigma row j1)
DEVNI void bm_reestimate(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, GmScalars &NOALIAS S, int jj, double newA)
{
    const int M = S.M, ld = W.ld, j1 = jj + 1;
    PAR(i, M) W.tp[i] = W.Sig[(size_t)j1 * ld + i];
    blk_sync(B);
    const double oldA = W.A[jj];
    const double dinv = 1.0 / (newA - oldA);
    const double kappa = 1.0 / (W.tp[j1] + dinv);
    const double mujj = W.mu[j1];
    blk_sync(B);
    if (B.tid == 0) W.A[jj] = newA;
    PAR(i, M) W.mu[i] += (-mujj * kappa) * W.tp[i];
    for (int j = B.wave; j < M; j += B.nwave) {
        const double f = kappa * W.tp[j];
        for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] -= f * W.tp[i];
    }
    blk_sync(B);
    PAR(i, M) W.tmp[i] = W.Sig[(size_t)j1 * ld + i];           // the updated row
    blk_sync(B);
    if (S.bp_ok != M) bm_weighted_rows(B, F, W, K, M);
    bm_rows_dot(B, W, K, M, W.tmp, [&](int i, double t) {
        W.Sin[i] = W.Sin[i] + t * t * kappa;
        W.Qin[i] = W.Qin[i] + mujj * kappa * t;
    });
    blk_sync(B);
}

// one call of the inner routine, :397-827.  *loglik = training log-likelihood at exit.
DEV int bm_inner(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, double lambda, double alpha,
                 GmScalars &NOALIAS S, int iter, double *loglik)
{
    const int N = F.N;
    const bool first = iter <= 1;
    if (first) {                                               // :1225-1385
        S.M = 2;
        PAR(i, K) W.upos[i] = UP_FREE;
        blk_sync(B);
        double pa = 0, pb = 0, pc = 0, pd = 0;
        PAR(h, N) {
            const double tp = -1 + 2 * F.y[h];
            const double lo = log(((tp * 0.9 + 1) / 2) / (1 - (tp * 0.9 + 1) / 2));
            const double ph = bm_col(F, W, N, 0, h);
            pa += ph; pb += ph * ph; pc += lo; pd += ph * lo;
        }
        const double sa = blk_sum(B, pa), sb = blk_sum(B, pb), sc = blk_sum(B, pc), sd = blk_sum(B, pd);
        if (B.tid == 0) {
            W.used[0] = 0; W.upos[0] = 0;
            const double det = (double)N * sb - sa * sa;
            double m0, m1;
            if (fabs(det) > 1e-10 * N * (sb > 0 ? sb : 1)) { m0 = (sb * sc - sa * sd) / det; m1 = ((double)N * sd - sa * sc) / det; }
            else { const double c0 = sa / N, den = (double)N * (1 + c0 * c0); m0 = sc / den; m1 = c0 * sc / den; }
            W.mu[0] = m0; W.mu[1] = m1;
            double a0 = (m1 == 0) ? 1 : 1 / (m1 * m1);
            if (a0 < 1e-3) a0 = 1e-3;
            if (a0 > 1e3) a0 = 1e3;
            W.A[0] = a0;
        }
    } else {
        PAR(i, K) if (W.upos[i] == UP_LOST) W.upos[i] = UP_FREE;
    }
    blk_sync(B);
    const int initial = W.used[0];
    int ini_removed = first ? 0 : 1;
    if (bm_fullstat(B, F, W, K, S)) { S.status |= ST_CHOLESKY | ST_ABORT; return 1; }
    int sel = ACT_NONE, jj = -1, n_todo = 0, last_it = 0, i_iter = 0;
    const int it_max = iter == 1 ? 10 : 100;
    double ll = 1e-30, ll0;
    while (!last_it) {
        i_iter++;
        CNT(c.n_inner++);
        ll0 = ll;
        double best; int any_del;
        int nu;
        { PH_BEGIN(); nu = bm_delta_ml(B, W, K, N, S.M - 1, lambda, alpha, S.v.epis, &any_del, &best); PH_END(PH_DML); }
        int worthwhile;
        if (sel == ACT_TERM && !ini_removed && S.M > 2) nu = -1;
        if (nu == -1 && ini_removed) { worthwhile = 0; sel = ACT_TERM; }
        else if (nu == -1 && !ini_removed && S.M > 2) {
            worthwhile = 1;
            nu = initial;
            if (B.tid == 0) { W.act[nu] = ACT_DEL; W.todo[0] = initial; }
            blk_sync(B);
            n_todo = 1; ini_removed = 1; sel = ACT_DEL;
        } else {
            worthwhile = 1;
            const int act_nu = W.act[nu];
            double cutoff = best * (act_nu == ACT_ADD ? S.v.n_add : 1.0);     // NEmainEff.c:422 0.90 | NeFull.c:254 0.99
            if (cutoff < 0.001) cutoff = 0.001;
            n_todo = bm_collect(B, W, K, cutoff);
            if (act_nu == ACT_DEL && n_todo > 1) n_todo = 1;
            if (n_todo == 0) worthwhile = 0;
        }
        if (!worthwhile) sel = ACT_TERM;
        if (worthwhile) {
            for (int u = 0; u < n_todo; u++) {
                nu = W.todo[u];
                sel = W.act[nu];
                const double newA = W.aroot[nu];
                if (sel == ACT_REEST || sel == ACT_DEL) {
                    const int l = W.upos[nu];
                    if (l >= 0) jj = l;
                    else { S.status |= ST_STALE; if (jj < 0 || jj >= S.M - 1) { S.status |= ST_ABORT; return 1; } }
                }
                if (sel == ACT_REEST && fabs(log(newA) - log(W.A[jj])) <= 1e-3 && any_del == 0) sel = ACT_TERM;
                blk_sync(B);
                PH_BEGIN();
                if (sel == ACT_REEST) {
                    CNT(c.n_reest++; c.sum_m_action += S.M);
                    bm_reestimate(B, F, W, K, S, jj, newA);
                } else if (sel == ACT_ADD) {
                    if (S.M + 1 > W.cap) { S.status |= ST_OVERFLOW | ST_ABORT; return 1; }
                    // NeFull.c:549-553: more bases than the R wrapper's bMax (the reference tests this from the second outer
                    // iteration on and overruns its arrays in the first; here the fit is stopped either way)
                    if (S.v.epis && S.M > W.bmax) { S.status |= ST_OVERFLOW | ST_ABORT; return 1; }
                    CNT(c.n_add++; c.sum_m_action += S.M);
                    bm_add(B, F, W, K, S, nu, newA);
                } else if (sel == ACT_DEL) {
                    CNT(c.n_del++; c.sum_m_action += S.M);
                    bm_delete(B, F, W, K, S, jj, nu);
                    if (nu == initial) ini_removed = 1;
                }
                PH_END(PH_ACTION);
                CNT(if (S.M > c.m_max) c.m_max = S.M);
                if (u == n_todo - 1) {                         // :749-762
                    if (bm_fullstat(B, F, W, K, S)) { S.status |= ST_CHOLESKY | ST_ABORT; return 1; }
                }
            }
        }
        if (sel == ACT_TERM && ini_removed) last_it = 1;
        if ((i_iter == it_max && S.M == 2) || i_iter > it_max) last_it = 1;
        if (i_iter == it_max) sel = ACT_TERM;
        bm_phi_mu(B, F, W, S.M, W.mu, W.pm);
        double lp = 0;
        PAR(h, N) {
            const double ex = exp(W.pm[h]);
            lp += F.y[h] * log(ex / (1 + ex)) + (1 - F.y[h]) * log(1 / (1 + ex));
        }
        ll = blk_sum(B, lp);
        const double dL = fabs((ll - ll0) / ll0);
        if (dL < 1e-3) sel = ACT_TERM;
    }
    *loglik = ll;
    return 0;
}

// The whole fit, :236-389.  On return W.mu[0] is the intercept, W.mu[l+1] the weight of used[l]
// (normalised-column units), S.M the model size incl. the intercept.
DEV void bm_fit(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, double lambda, double alpha,
                GmScalars &NOALIAS S, double *loglik)
{
    S.status = 0; S.M = 2; S.beta = 0; S.b = 0;
    S.v.n_add = S.v.epis ? 0.99 : 0.90;
    CNT(c = FitCounters{});
    PAR(i, W.ld) W.A[i] = 0;                                   // Calloc'd per fit, never cleared after (Q12)
    blk_sync(B);
    double vk = 1e-30, vk0, err = 1000, ll = 0;
    int iter = 0;
    while (iter < 100 && err > 1e-8) {
        iter++;
        vk0 = vk;
        if (bm_inner(B, F, W, K, lambda, alpha, S, iter, &ll)) break;
        double ap = 0;
        if (S.v.epis) { PAR(i, S.M - 1) ap += W.A[i]; }        // NeFull.c:133-134: the M-1 precisions
        else { PAR(i, S.M) ap += fabs(W.A[i]); }               // NEmainEff.c:340: dasum over M = N_used + 1 entries (Q12)
        vk = blk_sum(B, ap);
        err = fabs(vk - vk0) / S.M;
        if (S.outer_log && B.tid == 0) { double *o = S.outer_log + 3 * (iter - 1); o[0] = err; o[1] = 0; o[2] = 0; }   // NEmainEff.c:342
    }
    *loglik = ll;
    CNT(c.n_outer = iter; c.m_final = S.M - 1; c.status = S.status);
    blk_sync(B);
}

// fold score, R/GetModelError.R:34-57: mean Bernoulli log-likelihood of the held-out rows
DEV double bm_fold_loglik(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, const GmScalars &NOALIAS S)
{
    const int nte = F.nte, M = S.M;
    // eta -> exp(eta), with R's clamp applied only when max/min cross the thresholds
    double mx = -1e300, mn = 1e300;
    int any = 0;
    for (int j = 1; j < M; j++) if (W.mu[j] / F.scale[W.used[j - 1]] != 0) any = 1;
    if (!any) return 0.0;                                      // null model: logL <- 0 (:42-43)
    PAR(h, nte) {
        double eta = 0;
        for (int j = 1; j < M; j++) { const int u = W.used[j - 1]; eta += F.Xte[(size_t)u * nte + h] * (W.mu[j] / F.scale[u]); }
        const double t = exp(W.mu[0] + eta);
        W.pm[h] = t;
        if (t > mx) mx = t;
        if (t < mn) mn = t;
    }
    double dummy; int di;
    blk_argmax(B, mx, 0, &mx, &di);
    blk_argmax(B, -mn, 0, &dummy, &di); mn = -dummy;
    blk_sync(B);
    double part = 0;
    PAR(h, nte) {
        double t = W.pm[h];
        if (mx > 1e10 && t > 1e10) t = 1e5;
        if (mn < 1e-10 && t < 1e-10) t = 1e-5;
        part += F.yte[h] * log(t / (1 + t)) + (1 - F.yte[h]) * log(1 / (1 + t));
    }
    return blk_sum(B, part) / nte;
}
