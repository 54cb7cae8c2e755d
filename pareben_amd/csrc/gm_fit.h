// gm_fit.h -- one Gaussian main-effect EBEN fit executed by ONE workgroup.
//
// What it computes is what EBEN_orig/src/elasticNetLinearNeMainEff.c computes for one
// (training fold, alpha, lambda): the outer intercept loop (:155-197) around the inner
// add / re-estimate / delete marginal-likelihood ascent (:248-809).  How it computes it is
// different and MI355X-first:
//
//   * Gram space.  The reference caches BASIS_PHI[l][i] = x_i . Phi_l / scale_i per fit and
//     rebuilds it by sweeping the N x K design at every add (:1608-1630) and every outer
//     iteration (:1144-1201).  Phi_l is just the normalised design column of the l-th active
//     feature, so that row depends only on (fold, feature): it is row used[l] of the fold's
//     normalised Gram matrix F.G, computed once per fold by gram_kernel and shared by every
//     (alpha, lambda) cell.  Likewise Phi'Phi (:1874) = G[used,used], Phi't (:1266) =
//     bt[used], Phi'1 = cs[used].  The fit never sweeps the design; its HBM traffic is Gram
//     rows (8*K*M bytes per action / full-stat) and a handful of K-vectors.
//   * One workgroup per fit; every K-, M- and M^2-sized loop is spread over the workgroup;
//     reductions are fixed trees (blk.h) so results do not depend on timing or GPU count.
//   * Sigma is updated in place (rank-1 up/down-dates, bordered add, swap-with-last delete)
//     with a fixed leading dimension instead of the reference's SIGMANEW copy.
//
// Reference quirks kept because they change results: Q1 first basis is
// column 0 and is force-deleted once in outer iteration 1; Q2 the deleted weight is truncated to
// an int; Q3 gamma[0] is not refreshed by the full-stat pass; Q4 stale roots inside a block
// update; the stale-slot delete when the forced removal finds column 0 already gone.
#pragma once
#include "blk.h"
#include "types.h"


enum { ACT_NONE = -10, ACT_REEST = 0, ACT_ADD = 1, ACT_DEL = -1, ACT_TERM = 10 };
enum { UP_FREE = -1, UP_LOST = -2 };

enum { PH_FS_FEAT = 0, PH_FS_REST = 1, PH_DML = 2, PH_ACTION = 3, PH_NOISE = 4, PH_INVERSE = 5, PH_KSWEEP = 6, PH_TOTAL = 7,
       PH_MATVEC = 8, PH_RANK1 = 9, PH_REFRESH = 10, PH_HBUILD = 11, PH_MU = 12, PH_TRACK = 13, PH_INV_PIVOT = 14, PH_INV_TN = 15,
       PH_FS_MINE = 16, PH_FS_CHUNKS = 17, PH_SQ_MINE = 18, PH_SQ_CHUNKS = 19, PH_FS_WAIT = 20, PH_SQ_WAIT = 21, PH_N = 24 };

struct GmScalars {
    GmVariant v;       // main-effect / epistasis rule set
    long long *ph;     // phase ticks (LDS), diagnostic build only
    double beta;       // noise precision
    double b;          // intercept
    int M;             // active-set size
    int status;
    FitCounters *c;    // one copy per workgroup (LDS), updated by thread 0 only
    const FsShare *share = nullptr;   // shared full-stat passes (device only; null = off)
    int fold = 0;
    int bp_ok = 0;     // binomial: model columns for which the weighted rows BP are current (0 = stale)
    int gc_ok = 0;     // Gaussian, device: the Gram block cache W.Gc matches the active set's slots (0 = rebuild at the next Hessian)
    unsigned long long *trace = nullptr;   // decision trace (types.h TR_*; null = off), trace_cap records
    long long trace_cap = 0;
    // The K-space half (Gram-row sweep + S_in / Q_in update) of the LAST unit of a block of actions, held back until the
    // iteration knows whether a full-stat pass follows (which recomputes S_in / Q_in from scratch and makes the sweep
    // dead work): gm_inner.  kind: 0 none, 1 re-estimate, 2 single add, 3 delete, 4 run of adds.
    struct Pending { int kind = 0, M = 0, T = 0, jj = -1, row = -1; double beta = 0, c1 = 0, c2 = 0; } pend;
    int defer = 1;
    int inv_pair = 3;                      // blocked inverse (gm_dev.h): bit 0 two pivot blocks per trip through memory, bit 1 the register form; PAREBEN_INV_PAIR=<bits>
    double *outer_log = nullptr;           // per-fit entries with verbose > 2: (err, intercept | -, residual variance | -) per outer iteration
};
#define CNT(stmt) do { if (B.tid == 0) { FitCounters &c = *S.c; stmt; } } while (0)

// S_out/Q_out from S_in/Q_in, MainEff.c:1320-1338 and :664-671
DEVNI void gm_refresh_out(const Blk &NOALIAS B, const GmWork &NOALIAS W, int K)
{
    PAR(i, K) {
        double s = W.Sin[i], q = W.Qin[i];
        int l = W.upos[i];
        if (l >= 0) {
            double a = W.A[l];
            W.Sout[i] = a * s / (a - s);
            W.Qout[i] = a * q / (a - s);
        } else {
            W.Sout[i] = s;
            W.Qout[i] = q;
        }
    }
    blk_sync(B);
}

// S_in / Q_in update of feature i from a = sum_j G[used[j], i] * vec[j].
// mode 0: re-estimate (:577-587), 1: add (:1699-1711), 2: delete (:1800-1808).
DEV void gm_sq_apply(const GmWork &NOALIAS W, int mode, double beta, double c1, double c2, const double *newrow, int i, double a)
{
    if (mode == 0) {                             // c1 = kappa, c2 = mu_jj
        const double ba = beta * a;
        W.Sin[i] = W.Sin[i] + ba * ba * c1;
        W.Qin[i] = W.Qin[i] + beta * c2 * c1 * a;
    } else if (mode == 1) {                      // c1 = s_ii, c2 = mu_i
        const double mc = beta * newrow[i] - beta * a;
        W.Sin[i] = W.Sin[i] - mc * mc * c1;
        W.Qin[i] = W.Qin[i] - c2 * mc;
    } else {                                     // c1 = Sigma_jj, c2 = (int) mu_jj
        const double ba = beta * a;
        W.Sin[i] = W.Sin[i] + ba * ba / c1;
        W.Qin[i] = W.Qin[i] + ba * c2 / c1;
    }
}

// Sigma <- H^-1 for the SPD M x M matrix held in Sig (in place, Gauss-Jordan without pivoting:
// the pivots are the Cholesky pivots squared, so a non-positive pivot means "not SPD").
// Stands in for dpotrf+dpotri (:1346-1369).  Returns 0 on success.
DEVNI int gm_spd_inverse_scalar(const Blk &NOALIAS B, const GmWork &NOALIAS W, int M)
{
    const int ld = W.ld;
    for (int k = 0; k < M; k++) {
        PAR(i, M) { W.v3[i] = W.Sig[(size_t)k * ld + i]; W.v4[i] = W.Sig[(size_t)i * ld + k]; }
        blk_sync(B);
        const double d = W.v3[k];
        if (!(d > 0)) return 1;
        const double rd = 1.0 / d;
        for (int j = B.wave; j < M; j += B.nwave) {
            const double rkj = W.v4[j] * rd;
            for (int i = B.lane; i < M; i += BLK_LANES) {
                double a;
                if (i == k) a = (j == k) ? rd : rkj;
                else if (j == k) a = -W.v3[i] * rd;
                else a = W.Sig[(size_t)j * ld + i] - W.v3[i] * rkj;
                W.Sig[(size_t)j * ld + i] = a;
            }
        }
        blk_sync(B);
    }
    return 0;
}

// ---- the phases: everything whose implementation is the hardware mapping (matrix-core passes, LDS staging, the job
// board, address-space-qualified loads).  The shipped library gets gm_dev.h; the CPU test harness under tests/emul, which
// steps this file's control flow without a GPU, names its own header of plain loops with the same signatures instead.
#ifndef GM_PHASES_H
#define GM_PHASES_H "gm_dev.h"
#endif
#include GM_PHASES_H

// Full statistics, MainEff.c:1209-1341 (Q3: gamma[0] is left alone).
DEV void gm_fullstat(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S,
                     bool very_first, bool mu_current = false)
{
    const int M = S.M, ld = W.ld;
    const double beta = S.beta;
    if (very_first) {
        if (B.tid == 0) {
            W.H[0] = F.G[(size_t)W.rowid[0] * K] * beta + W.A[0];
            W.Sig[0] = 1 / W.H[0];
        }
    }
    // mu = beta Sigma Phi't (:1266-1289).  mu_current: the caller comes straight from gm_final_update, which has just formed
    // exactly this product from the same Sigma, Phi't and beta (same expression, same order: the same bits) -- the M^2 loop
    // is not run a second time (nine of ten passes of a grid are reached that way)
    if (!mu_current) {
        PAR(l, M) W.v1[l] = W.bt[W.used[l]];                 // Phi' t
        blk_sync(B);
        PAR(i, M) {
            double a = 0;
            for (int j = 0; j < M; j++) a += W.v1[j] * W.Sig[(size_t)j * ld + i];
            W.mu[i] = a * beta;
        }
    }
    PAR(i, M) if (i >= 1) W.gam[i] = 1 - W.Sig[(size_t)i * ld + i] * W.A[i];
    gm_fs_pad(B, W, M);                                       // matrix-core pass: exact zeros in the ragged 16-block (gm_dev.h)
    blk_sync(B);
    {
        PH_BEGIN();
        GM_FS_CLOCK_BEGIN();
        gm_fullstat_pass(B, F, W, K, M, beta, S);
        GM_FS_CLOCK_END();
        PH_END(PH_FS_FEAT);
    }
    gm_refresh_out(B, W, K);
    CNT(c.n_fullstat++; c.sum_m_full += M; c.sum_m2_full += (int64_t)M * M);
    gm_fs_count(B, S, K, M);
}

// Per-feature marginal-likelihood change and action, MainEff.c:1372-1582.  Returns the arg-max
// feature and its value; ties are resolved in the reference's visiting order (below).
DEVNI int gm_delta_ml(const Blk &NOALIAS B, const GmWork &NOALIAS W, int K, int N, int M, double lambda, double alpha,
                    double residual, double varY, int iter, int i_iter, int epis, int *any_del_out, double *best)
{
    const double l1 = lambda * alpha, l2 = lambda * (1 - alpha);
    int prio_add = 0, prio_del = 0;
    if (M < 10) { prio_add = 1; prio_del = 0; }
    if (epis ? (M > 100 || residual <= varY * 0.1)                    // Full2.c:1257
             : (M > 100 || M >= N || residual <= varY * 0.1)) { prio_add = 0; prio_del = 1; }
    int my_add = 0, my_del = 0;
    // arg-max of this pass (valid when no rescan follows).  The reference's first scan (:1381-1535) walks the active list in
    // slot order, then the inactive one, and keeps the first strict maximum: among bit-equal dML values an active feature wins
    // over an inactive one and the lower slot over the higher.  Such ties are not rare on genotype designs -- a duplicated
    // column outside the model and its twin inside it with an astronomically large precision have bit-equal add and
    // re-estimate dML -- and the action TYPE of the winner sets the block cut-off, so the visiting order is kept: candidates
    // are ranked by key = slot (active) | M + index (inactive; among those only the type matters, and it is the same).
    double v1 = 0; int idx1 = 0x7fffffff;
    // the three loads of the next feature are issued before the (long) arithmetic of this one
    const gptr_ci g_upos = as_global(W.upos);
    const gptr_cd g_so = as_global(W.Sout), g_qo = as_global(W.Qout);
    int l_n = 0; double so_n = 0, qo_n = 0;
    if (B.tid < K) { l_n = g_upos[B.tid]; so_n = g_so[B.tid]; qo_n = g_qo[B.tid]; }
    PAR(i, K) {
        const int l = l_n;
        const double so = so_n, qo = qo_n;
        { const int in = i + B.nthr; if (in < K) { l_n = g_upos[in]; so_n = g_so[in]; qo_n = g_qo[in]; } }
        if (l == UP_LOST) { W.act[i] = ACT_NONE; continue; }   // in neither list: stale dML stays
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
                } else {
                    act = ACT_ADD;
                    d_ml = L;
                    if (!epis) my_add = 1;                     // Q15: only the main-effect kernel sets it
                }
            }
        } else if (l >= 0 && M > 1) {
            my_del = 1;
            act = ACT_DEL;
            const double o = W.A[l] - l2;
            const double L = (log(o / (o + so + l2)) + qo * qo / (o + so + l2)) * 0.5 - l1 / o;
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
    if (!epis && ((!any_add && iter == 1 && i_iter < 10) || (!any_add && residual >= varY * 0.95))) {   // :1557-1577
        PAR(i, K) if (W.act[i] == ACT_DEL) W.dml[i] = 0;
        rescanned = true;
    }
    blk_sync(B);
    double v = 0; int idx = 0x7fffffff;
    if (rescanned) {
        PAR(i, K) {
            const double d = W.dml[i];
            if (d > v) { v = d; idx = i; }
        }
    } else { v = v1; idx = idx1; }                            // first scan walks the two lists only: what the pass above saw
    double bv; int bi;
    blk_argmax(B, v, idx, &bv, &bi);                          // ties: lowest key (first scan) / lowest index (a rescan runs over all features in index order)
    if (!rescanned && bv > 0) bi = bi < M ? W.used[bi] : bi - M;
    if (!(bv > 0)) { bv = 0; bi = 0; }
    *best = bv;
    return bi;
}

// ordered list of features with dML >= cutoff (ascending index), MainEff.c:463-473
DEVNI int gm_collect(const Blk &NOALIAS B, const GmWork &NOALIAS W, int K, double cutoff)
{
    // every thread takes a contiguous slice of the features: one count, ONE block scan, one ordered write
    const int per = (K + B.nthr - 1) / B.nthr, i0 = B.tid * per, i1 = i0 + per < K ? i0 + per : K;
    const gptr_cd dml = as_global(W.dml);
    int n = 0;
    for (int i = i0; i < i1; i++) n += dml[i] >= cutoff ? 1 : 0;
    int tot;
    int off = blk_scan_excl(B, n, &tot);
    if (n) for (int i = i0; i < i1; i++) if (dml[i] >= cutoff) W.todo[off++] = i;
    blk_sync(B);
    return tot;
}

// re-estimate slot jj, MainEff.c:553-596
// `defer` (all four actions): leave the K-space half -- the Gram-row sweep with the S_in / Q_in update -- in S.pend
// instead of running it (gm_flush_pending runs it later, or nobody does: gm_inner).  The vector stays in W.v2 (W.vb for
// a run of adds), which nothing between here and the flush writes.
DEVNI void gm_reestimate(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S, int jj, double newA, bool defer)
{
    const int M = S.M, ld = W.ld;
    PAR(i, M) W.v2[i] = W.Sig[(size_t)jj * ld + i];
    blk_sync(B);
    const double oldA = W.A[jj];
    const double dinv = 1.0 / (newA - oldA);
    const double kappa = 1.0 / (W.v2[jj] + dinv);
    const double mujj = W.mu[jj];
    blk_sync(B);
    if (B.tid == 0) W.A[jj] = newA;
    PAR(i, M) W.mu[i] += (-mujj * kappa) * W.v2[i];
    { PH_BEGIN();
    gm_rank1(B, W, M, B.pool, [&](int j) { return -(kappa * W.v2[j]); }, [&](int i) { return W.v2[i]; });
    PH_END(PH_RANK1); }
    if (defer) { S.pend.kind = 1; S.pend.M = M; S.pend.beta = S.beta; S.pend.c1 = kappa; S.pend.c2 = mujj; }
    else gm_sq_update(B, F, W, K, M, W.v2, 0, S.beta, kappa, mujj, -1, S);
}

// add feature nu, MainEff.c:1585-1723 + :613-627
DEVNI void gm_add(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S, int nu, int rid, double newA, bool defer)
{
    const int M = S.M, ld = W.ld;
    const double beta = S.beta;
    const double *row = F.G + (size_t)rid * K;                // x_i . phi_nu / scale_i (Gram row of nu)
    PAR(l, M) W.v1[l] = beta * row[W.used[l]];                // beta Phi' phi
    blk_sync(B);
    { PH_BEGIN();
    gm_sigma_matvec<16>(B, W, M, W.v1, W.v2, B.pool, B.pool_n, (lptr_d)0);    // Sigma * (beta Phi' phi)
    PH_END(PH_MATVEC); }
    const double sii = 1.0 / (newA + W.Sin[nu]);
    const double mui = sii * W.Qin[nu];
    blk_sync(B);
    PAR(i, M) W.mu[i] += -mui * W.v2[i];
    { PH_BEGIN();
    gm_rank1(B, W, M, B.pool, [&](int j) { return sii * W.v2[j]; }, [&](int i) { return W.v2[i]; });
    PH_END(PH_RANK1); }
    PAR(i, M) {
        const double si = -sii * W.v2[i];
        W.Sig[(size_t)M * ld + i] = si;
        W.Sig[(size_t)i * ld + M] = si;
    }
    if (B.tid == 0) {
        W.Sig[(size_t)M * ld + M] = sii;
        W.A[M] = newA;
        W.mu[M] = mui;
        W.used[M] = nu;
        W.rowid[M] = rid;
        W.upos[nu] = M;
    }
    gm_gc_add(B, F, W, K, S, M, nu, rid);                      // the new slot's column of the Gram block cache (gm_dev.h)
    if (defer) { S.pend.kind = 2; S.pend.M = M; S.pend.beta = beta; S.pend.c1 = sii; S.pend.c2 = mui; S.pend.row = rid; }
    else gm_sq_update(B, F, W, K, M, W.v2, 1, beta, sii, mui, rid, S);
    GM_TRACE("    add nu=%d newA=%.15g sii=%.15g mui=%.15g tp0=%.15g tmp0=%.15g Sin=%.15g Qin=%.15g mu0=%.15g\n", nu, newA, sii, mui, W.v2[0], W.v1[0], W.Sin[nu], W.Qin[nu], W.mu[0]);
    S.M = M + 1;
}

// the K-space half a block's last unit left in S.pend (gm_inner)
DEV void gm_flush_pending(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S)
{
    const GmScalars::Pending p = S.pend;
    S.pend.kind = 0;
    if (p.kind == 1) gm_sq_update(B, F, W, K, p.M, W.v2, 0, p.beta, p.c1, p.c2, -1, S);
    else if (p.kind == 2) gm_sq_update(B, F, W, K, p.M, W.v2, 1, p.beta, p.c1, p.c2, p.row, S);
    else if (p.kind == 3) gm_sq_update(B, F, W, K, p.M, W.v2, 2, p.beta, p.c1, p.c2, -1, S, p.jj, p.row);
    else if (p.kind == 4) gm_flush_add_run(B, F, W, K, S, p.M, p.T, p.beta);
}

// delete slot jj, MainEff.c:1725-1822 + :640-651.  `nu` is the feature the action named; it
// differs from used[jj] only on the reference's stale-slot path.
DEVNI void gm_delete(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S, int jj, int nu, bool defer)
{
    const int M = S.M, ld = W.ld, last = M - 1;
    PAR(i, M) W.v2[i] = W.Sig[(size_t)jj * ld + i];
    blk_sync(B);
    const double sjj = W.v2[jj];
    const int mujj = (int)W.mu[jj];                           // Q2: int truncation
    const int gone = W.used[jj];
    const int gone_row = W.rowid[jj];
    // the sweep runs over the rows as they are BEFORE the slot shuffle below; held back, it is told which slot was freed
    if (defer) { S.pend.kind = 3; S.pend.M = M; S.pend.beta = S.beta; S.pend.c1 = sjj; S.pend.c2 = (double)mujj; S.pend.jj = jj; S.pend.row = gone_row; }
    else gm_sq_update(B, F, W, K, M, W.v2, 2, S.beta, sjj, (double)mujj, -1, S);
    PAR(i, M) W.mu[i] = W.mu[i] - mujj * W.v2[i] / sjj;
    gm_rank1(B, W, M, B.pool, [&](int j) { return -W.v2[j]; }, [&](int i) { return W.v2[i] / sjj; });
    blk_sync(B);
    if (jj != last) {                                         // move the last slot into jj
        PAR(i, M) { W.v3[i] = W.Sig[(size_t)last * ld + i]; W.v4[i] = W.Sig[(size_t)i * ld + last]; }
        blk_sync(B);
        PAR(i, last) {
            if (i != jj) {
                W.Sig[(size_t)jj * ld + i] = W.v3[i];         // column jj <- column last
                W.Sig[(size_t)i * ld + jj] = W.v4[i];         // row jj    <- row last
            }
        }
        if (B.tid == 0) {
            W.Sig[(size_t)jj * ld + jj] = W.v3[last];
            W.A[jj] = W.A[last];
            W.mu[jj] = W.mu[last];
            W.used[jj] = W.used[last];
            W.rowid[jj] = W.rowid[last];
            W.upos[W.used[last]] = jj;
        }
    }
    if (B.tid == 0) {
        if (gone == nu) W.upos[gone] = UP_FREE;
        else { W.upos[gone] = UP_LOST; }
        if (F.lazy && gone_row >= W.priv_base && gone_row < W.priv_base + W.priv_rows) W.pfree[++W.pfree[0]] = gone_row;
    }
    S.M = last;
    S.gc_ok = 0;                                              // slots were reshuffled: the next Hessian regathers its Gram block
    blk_sync(B);
}

// H = beta Phi'Phi + diag(A); Sigma = H^-1; mu = beta Sigma Phi't.  MainEff.c:1841-1921
DEVNI int gm_final_update(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S)
{
    const int M = S.M, ld = W.ld;
    const double beta = S.beta;
    gm_hessian_build(B, F, W, K, S);
    PAR(l, M) W.v1[l] = W.bt[W.used[l]];
    blk_sync(B);
    { PH_BEGIN(); const int bad = gm_spd_inverse(B, W, M, S.ph, S.inv_pair); PH_END(PH_INVERSE); if (bad) return 1; }
    gm_inverse_count(B, S, M);
    PH_BEGIN();
    gm_mu_update(B, W, M, beta);
    blk_sync(B);
    PH_END(PH_MU);
    return 0;
}

// ---- decision trace (diagnostics only: S.trace is null in every production launch) -------------------------
DEV unsigned long long gm_dbits(double v) { unsigned long long u; __builtin_memcpy(&u, &v, 8); return u; }
DEV double blk_max(const Blk &NOALIAS B, double v)
{
    double bv; int bi;
    blk_argmax(B, v, 0, &bv, &bi);
    blk_sync(B);
    return bv;
}
// the record of this inner iteration, or null (thread-uniform)
DEV unsigned long long *gm_trace_rec(const GmScalars &NOALIAS S)
{
    return (S.trace && (long long)S.trace[0] < S.trace_cap) ? S.trace + TR_NSLOT * (S.trace[0] + 1) : nullptr;
}
// the decision: arg-max feature, its action and value, the runner-up, the block cut-off and the relative distance of
// the nearest dML to it (what a last-bit change would have to bridge to alter the to-do list)
DEVNI void gm_trace_decision(const Blk &NOALIAS B, const GmWork &NOALIAS W, int K, const GmScalars &NOALIAS S, unsigned long long *tr, int iter, int i_iter,
                             int nu, double best, int worthwhile, int n_todo)
{
    double cutoff = 0;
    if (worthwhile && nu >= 0) {
        cutoff = best * (W.act[nu] == ACT_ADD ? S.v.n_add : 1.0);
        if (cutoff < S.v.ml_delta) cutoff = S.v.ml_delta;
    }
    double second = 0, nearest = -INFINITY;                    // nearest: max of the negated distance
    PAR(i, K) {
        const double d = W.dml[i];
        if (!(d > 0)) continue;
        if (i != nu && d > second) second = d;
        if (cutoff > 0 && -(fabs(d - cutoff) / cutoff) > nearest) nearest = -(fabs(d - cutoff) / cutoff);
    }
    second = blk_max(B, second);
    nearest = blk_max(B, nearest);
    if (B.tid == 0) {
        tr[TR_ITER] = iter; tr[TR_IITER] = i_iter; tr[TR_MBEFORE] = S.M; tr[TR_NU] = (unsigned long long)(long long)nu;
        tr[TR_ACT] = (unsigned long long)(long long)(nu >= 0 ? W.act[nu] : ACT_NONE); tr[TR_NTODO] = worthwhile ? n_todo : 0;
        tr[TR_BEST] = gm_dbits(best); tr[TR_SECOND] = gm_dbits(second); tr[TR_CUTOFF] = gm_dbits(cutoff); tr[TR_NEAREST] = gm_dbits(-nearest);
    }
    blk_sync(B);
}
// the state after the iteration: XOR of the bit patterns of S_in, Q_in and (Sigma, mu) -- order-free, so equal
// hashes on two builds mean equal bits whatever the layout
DEVNI void gm_trace_state(const Blk &NOALIAS B, const GmWork &NOALIAS W, int K, const GmScalars &NOALIAS S, unsigned long long *tr, int sel)
{
    unsigned long long hs = 0, hq = 0, hg = 0;
    PAR(i, K) { hs ^= gm_dbits(W.Sin[i]); hq ^= gm_dbits(W.Qin[i]); }
    const int M = S.M;
    for (int e = B.tid; e < M * M; e += B.nthr) { const int j = e / M, i = e - j * M; hg ^= gm_dbits(W.Sig[(size_t)j * W.ld + i]); }
    PAR(j, M) hg ^= gm_dbits(W.mu[j]);
    hs = blk_xor64(B, hs); hq = blk_xor64(B, hq); hg = blk_xor64(B, hg);
    if (B.tid == 0) {
        unsigned hu = 0;                                       // order-free hash of the active set: which features, not only how many
        for (int j = 0; j < M; j++) hu += (unsigned)(W.used[j] + 1) * 2654435761u;
        tr[TR_SEL] = (unsigned long long)(long long)sel; tr[TR_MAFTER] = (unsigned long long)M | ((unsigned long long)hu << 32); tr[TR_BETA] = gm_dbits(S.beta);
        tr[TR_HSIN] = hs; tr[TR_HQIN] = hq; tr[TR_HSIG] = hg;
        S.trace[0]++;
    }
    blk_sync(B);
}

// One call of the inner routine (MainEff.c:248-809) for outer iteration `iter`.  On return
// *cs = sum_i Csum_i and *csy = Csum.y with Csum the column sums of
// C^-1 = beta I - beta^2 Phi Sigma Phi' (:741-781, :172-187), formed in O(N M + M^2).
DEV int gm_inner(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, double lambda, double alpha,
                 GmScalars &NOALIAS S, int iter, double residual, double varY, double *cs, double *csy)
{
    const int N = F.N, ld = W.ld;
    const bool first = iter <= 1;
    const bool defer_ok = S.defer;                            // PAREBEN_DEFER=0 (A/B runs): every sweep at its action
    if (first) {                                              // :1003-1090, Q1
        S.M = 1;
        if (!S.v.epis) S.beta = 1 / (varY * 0.01 + 1e-10);
        else {                                                // Full2.c:917-921
            double sd = sqrt(varY);
            if (sd < 1e-6) sd = 1e-6;
            S.beta = 1 / ((sd * 0.1) * (sd * 0.1));
        }
        PAR(i, K) W.upos[i] = UP_FREE;
        if (B.tid == 0 && F.lazy) {                              // all private rows are free again
            W.pfree[0] = W.priv_rows;
            for (int j = 0; j < W.priv_rows; j++) W.pfree[1 + j] = W.priv_base + W.priv_rows - 1 - j;
        }
        const int rid0 = gm_row(B, F, W, K, 0);
        if (rid0 < 0) { S.status |= ST_OVERFLOW | ST_ABORT; return 1; }
        blk_sync(B);
        if (B.tid == 0) {
            W.used[0] = 0;
            W.rowid[0] = rid0;
            W.upos[0] = 0;
            const double p = F.G[(size_t)rid0 * K] * S.beta;
            const double q = (F.bt0[0] - S.b * F.cs[0]) * S.beta;
            double a0 = p * p / (q * q - p);
            if (a0 < 0) a0 = S.v.alpha_max;
            if (a0 > S.v.alpha_max) a0 = S.v.alpha_max;
            W.A[0] = a0;
        }
    } else {
        PAR(i, K) if (W.upos[i] == UP_LOST) W.upos[i] = UP_FREE;   // :1092-1107 rebuilds the list
    }
    PAR(i, W.cap + 1) W.gam[i] = 0;                          // calloc per call, :344
    PAR(i, K) W.bt[i] = F.bt0[i] - S.b * F.cs[i];            // x_i.(y - b)/scale_i, :1171-1177
    blk_sync(B);
    const int initial = W.used[0];
    int ini_removed = first ? 0 : 1;
    int i_iter = 0;
    gm_fullstat(B, F, W, K, S, iter == 1);

    int sel = ACT_NONE, jj = -1, n_todo = 0, last_it = 0;
    const int it_max = iter == 1 ? 10 : 100;
    while (!last_it) {
        i_iter++;
        CNT(c.n_inner++);
        double best; int any_del;
        PH_BEGIN();
        int nu = gm_delta_ml(B, W, K, N, S.M, lambda, alpha, residual, varY, iter, i_iter, S.v.epis, &any_del, &best);
        int worthwhile;
        if (sel == ACT_TERM && !ini_removed && S.M > 1) nu = -1;
        if (nu == -1 && ini_removed) {
            worthwhile = 0; sel = ACT_TERM;
        } else if (nu == -1 && !ini_removed && S.M > 1) {     // forced removal, :437-446
            worthwhile = 1;
            nu = initial;
            if (B.tid == 0) { W.act[nu] = ACT_DEL; W.todo[0] = initial; }
            blk_sync(B);
            n_todo = 1;
            ini_removed = 1;
            sel = ACT_DEL;
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
        PH_END(PH_DML);
        bool any_upd = false;
        S.pend.kind = 0;
        unsigned long long *const tr = gm_trace_rec(S);
        if (tr) gm_trace_decision(B, W, K, S, tr, iter, i_iter, nu, best, worthwhile, n_todo);
        if (worthwhile) {
            PH_BEGIN();
            for (int u = 0; u < n_todo; u++) {
                nu = W.todo[u];
                sel = W.act[nu];
                if (sel == ACT_ADD) {                             // a run of consecutive adds: one Gram-row sweep for all (gm_dev.h)
                    const int T = gm_add_run(B, F, W, K, S, u, n_todo, defer_ok);
                    if (T < 0) return 1;
                    if (T > 0) {
                        u += T - 1;
                        nu = W.todo[u];
                        any_upd = true;
                        blk_sync(B);
                        CNT(if (S.M > c.m_max) c.m_max = S.M);
                        continue;
                    }
                }
                const double newA = W.aroot[nu];
                if (sel == ACT_REEST || sel == ACT_DEL) {
                    const int l = W.upos[nu];
                    if (l >= 0) jj = l;
                    else {                                    // stale slot (reference UB path)
                        S.status |= ST_STALE;
                        if (jj < 0 || jj >= S.M) { S.status |= ST_ABORT; return 1; }
                    }
                }
                if (sel == ACT_REEST && fabs(log(newA) - log(W.A[jj])) <= S.v.reest_tol && any_del == 0)
                    sel = ACT_TERM;
                blk_sync(B);
                const bool defer = defer_ok && u == n_todo - 1;
                bool upd = false;
                if (sel == ACT_REEST) {
                    CNT(c.n_reest++; c.sum_m_action += S.M);
                    gm_reestimate(B, F, W, K, S, jj, newA, defer);
                    upd = true;
                } else if (sel == ACT_ADD) {
                    if (S.M + 1 > W.cap) { S.status |= ST_OVERFLOW | ST_ABORT; return 1; }
                    const int rid = gm_row(B, F, W, K, nu);
                    if (rid < 0) { S.status |= ST_OVERFLOW | ST_ABORT; return 1; }
                    CNT(c.n_add++; c.sum_m_action += S.M);
                    gm_add(B, F, W, K, S, nu, rid, newA, defer);
                    if (S.M > W.cap_flag) S.status |= ST_OVERFLOW;
                    upd = true;
                } else if (sel == ACT_DEL) {
                    CNT(c.n_del++; c.sum_m_action += S.M);
                    gm_delete(B, F, W, K, S, jj, nu, defer);
                    upd = true;
                }
                if (upd) {
                    any_upd = true;
                    blk_sync(B);
                    CNT(if (S.M > c.m_max) c.m_max = S.M);
                }
            }
            if (any_upd) {                                    // gamma of the block's final model (:664-671; only the last refresh of a block is ever read)
                PAR(i, S.M) W.gam[i] = 1 - W.A[i] * W.Sig[(size_t)i * ld + i];
                blk_sync(B);
            }
            PH_END(PH_ACTION);
        }
        // The reference refreshes S_out / Q_out after every action (:664-671), but nothing reads them before the next dML
        // pass, and S_in / Q_in between two actions only through the add of a later unit of the same block.  So the K-space
        // half of a block's LAST unit waits (S.pend) until it is known whether the noise update below ends in a full-stat
        // pass: that pass recomputes S_in / Q_in for every feature from Sigma and mu, and the sweep -- M Gram rows of K
        // entries -- would be dead work (two of three inner iterations of BASELINE config 2 end that way).  Otherwise the
        // sweep runs now, with the operands of its action, and S_out / Q_out are refreshed once.  Same values either way.
        bool fs_done = false;
        if (sel == ACT_TERM || i_iter <= 10 || i_iter % 5 == 0 || n_todo >= 2) {   // :685-729
            const int M = S.M;
            PH_BEGIN();
            double ee_part = 0;
            gm_stage_model(B, F, W, M, W.mu);
            for (int h = B.tid; h < N; h += 2 * B.nthr) {     // a thread's samples in the same order, two per trip
                const int h2 = h + B.nthr;
                double pm0, pm1;
                gm_model_at2(B, F, W, M, W.mu, N, h, h2 < N ? h2 : h, pm0, pm1);
                const double e0 = (F.y[h] - S.b) - pm0;
                ee_part += e0 * e0;
                if (h2 < N) { const double e1 = (F.y[h2] - S.b) - pm1; ee_part += e1 * e1; }
            }
            const double ee = blk_sum(B, ee_part);
            double g_part = 0;
            PAR(i, M) g_part += W.gam[i];
            const double gsum = blk_sum(B, g_part);
            const double beta_old = S.beta;
            double nb = (N - gsum) / ee;
            if (nb > 1e6 / varY) nb = 1e6 / varY;
            S.beta = nb;
            const double dlb = log(nb) - log(beta_old);
            PH_END(PH_NOISE);
            if (fabs(dlb) > 1e-6) {
                if (gm_final_update(B, F, W, K, S)) { S.status |= ST_CHOLESKY | ST_ABORT; return 1; }
                if (sel != ACT_TERM) { S.pend.kind = 0; gm_fullstat(B, F, W, K, S, false, true); fs_done = true; }
            }
        }
        if (S.pend.kind) { PH_BEGIN(); gm_flush_pending(B, F, W, K, S); PH_END(PH_ACTION); }
        if (any_upd && !fs_done) { PH_BEGIN(); gm_refresh_out(B, W, K); PH_END(PH_REFRESH); }
        GM_TRACE("  it %d.%d M=%d sel=%d ntodo=%d beta=%.15g mu0=%.15g A0=%.15g gam0=%.15g\n", iter, i_iter, S.M, sel, n_todo, S.beta, W.mu[0], W.A[0], W.gam[0]);
        if (tr) gm_trace_state(B, W, K, S, tr, sel);
        if (sel == ACT_TERM && ini_removed) last_it = 1;
        if ((i_iter == it_max && S.M == 1) || i_iter > it_max) last_it = 1;
        if (i_iter == it_max) sel = ACT_TERM;
    }
    {   // column sums of C^-1
        const int M = S.M;
        PAR(l, M) W.v1[l] = F.cs[W.used[l]];                 // Phi' 1
        blk_sync(B);
        PAR(i, M) {
            double a = 0;
            for (int j = 0; j < M; j++) a += W.v1[j] * W.Sig[(size_t)j * ld + i];
            W.v2[i] = a;
        }
        blk_sync(B);
        const double beta = S.beta, b2 = beta * beta;
        double a_part = 0, b_part = 0;
        gm_stage_model(B, F, W, M, W.v2);
        for (int h = B.tid; h < N; h += 2 * B.nthr) {
            const int h2 = h + B.nthr;
            double v0, v1;
            gm_model_at2(B, F, W, M, W.v2, N, h, h2 < N ? h2 : h, v0, v1);
            const double c0 = beta - b2 * v0;
            a_part += c0;
            b_part += c0 * F.y[h];
            if (h2 < N) { const double c1 = beta - b2 * v1; a_part += c1; b_part += c1 * F.y[h2]; }
        }
        *cs = blk_sum(B, a_part);
        *csy = blk_sum(B, b_part);
    }
    return 0;
}

// The whole fit: MainEff.c:55-242.  On return S.b = intercept, S.beta = noise precision,
// W.used/W.mu/W.Sig hold the model (mu in normalised-column units).
DEV void gm_fit(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, double lambda, double alpha,
                GmScalars &NOALIAS S)
{
    S.b = F.ymean;
    S.status = 0;
    S.gc_ok = 0;
    S.M = 1;
    CNT(c = FitCounters{});
    const double varT = F.varY;
    double residvar = 1e10, err = 1000, vk = 1e-30, vk0;
    int iter = 0;
    while (iter < 100 && err > 1e-8 && residvar >= varT * 0.01) {
        iter++;
        vk0 = vk;
        double cs, csy;
        if (gm_inner(B, F, W, K, lambda, alpha, S, iter, residvar, varT, &cs, &csy)) break;
        S.b = csy / (cs + S.v.b_eps);
        double a_part = 0;
        PAR(i, S.M) a_part += W.A[i];
        vk = blk_sum(B, a_part);
        err = fabs(vk - vk0) / S.M;
        residvar = 1 / (S.beta + 1e-10);
        if (S.outer_log && B.tid == 0) { double *o = S.outer_log + 3 * (iter - 1); o[0] = err; o[1] = S.b; o[2] = residvar; }   // MainEff.c:196
    }
    CNT(c.n_outer = iter; c.m_final = S.M; if (S.M > c.m_max) c.m_max = S.M; c.status = S.status);
    blk_sync(B);
}

// fold score, R/GetModelError.R:7-32: SSE of the held-out rows under the fitted model
DEV double gm_fold_sse(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, const GmScalars &NOALIAS S)
{
    const int nte = F.nte, M = S.M;
    double part = 0;
    PAR(h, nte) {
        double pred = 0;
        for (int j = 0; j < M; j++) {
            const int uj = W.used[j];
            pred += F.Xte[(size_t)uj * nte + h] * (W.mu[j] / F.scale[uj]);
        }
        const double r = F.yte[h] - (S.b + pred);
        part += r * r;
    }
    return blk_sum(B, part);
}
