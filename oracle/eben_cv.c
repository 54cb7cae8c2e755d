/*
 * eben_cv.c -- oracle: the per-cell cross-validation harness, a CPU restatement of
 * R/TestModel.R:6-39 (row split per fold, fit, score) and R/GetModelError.R:6-59 (fold SSE /
 * mean Bernoulli log-likelihood from the fit's non-zero weights), plus the "keep rows with a
 * non-zero weight" rule of EBEN_orig/R/EBelasticNet.Gaussian.R:56-66 and ...Binomial.R:47-50.
 *
 * TEST INFRASTRUCTURE ONLY (see eben_oracle.h).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "eben_oracle.h"

/* gaussian fold score, R/GetModelError.R:7-32: SSE on the held-out rows */
static double fold_sse(const double *Xte, const double *yte, int nte, int K, const double *Beta,
                       long n_eff, int epis, double mu0)
{
    double *pred = (double *)calloc(nte, sizeof(double));
    for (long e = 0; e < n_eff; e++) {
        double w = Beta[2 * n_eff + e];
        int keep = epis ? (Beta[4 * n_eff + e] != 0) : (w != 0);
        if (!keep) continue;
        int l1 = (int)Beta[e] - 1, l2 = (int)Beta[n_eff + e] - 1;
        const double *a = Xte + (size_t)l1 * nte, *b = Xte + (size_t)l2 * nte;
        if (l1 == l2) for (int i = 0; i < nte; i++) pred[i] += a[i] * w;
        else          for (int i = 0; i < nte; i++) pred[i] += a[i] * b[i] * w;
    }
    double sse = 0;
    for (int i = 0; i < nte; i++) { double r = yte[i] - (mu0 + pred[i]); sse += r * r; }
    free(pred);
    (void)K;
    return sse;
}

/* binomial fold score, R/GetModelError.R:34-57 */
static double fold_loglik(const double *Xte, const double *yte, int nte, int K, const double *Beta,
                          long n_eff, double mu0)
{
    int any = 0;
    double *eta = (double *)calloc(nte, sizeof(double));
    for (long e = 0; e < n_eff; e++) {
        double w = Beta[2 * n_eff + e];
        if (w == 0) continue;
        any = 1;
        int l1 = (int)Beta[e] - 1, l2 = (int)Beta[n_eff + e] - 1;
        const double *a = Xte + (size_t)l1 * nte, *b = Xte + (size_t)l2 * nte;
        if (l1 == l2) for (int i = 0; i < nte; i++) eta[i] += a[i] * w;
        else          for (int i = 0; i < nte; i++) eta[i] += a[i] * b[i] * w;
    }
    double out = 0;
    if (any) {
        double mx = -INFINITY, mn = INFINITY;
        for (int i = 0; i < nte; i++) { eta[i] = exp(mu0 + eta[i]); if (eta[i] > mx) mx = eta[i]; if (eta[i] < mn) mn = eta[i]; }
        if (mx > 1e10) for (int i = 0; i < nte; i++) if (eta[i] > 1e10) eta[i] = 1e5;
        if (mn < 1e-10) for (int i = 0; i < nte; i++) if (eta[i] < 1e-10) eta[i] = 1e-5;
        double s = 0;
        for (int i = 0; i < nte; i++) s += yte[i] * log(eta[i] / (1 + eta[i])) + (1 - yte[i]) * log(1 / (1 + eta[i]));
        out = s / nte;
    }
    free(eta);
    (void)K;
    return out;
}

static void add_counters(eben_counters *a, const eben_counters *b)
{
    a->n_outer += b->n_outer; a->n_inner += b->n_inner; a->n_add += b->n_add; a->n_del += b->n_del;
    a->n_reest += b->n_reest; a->n_fullstat += b->n_fullstat; a->sum_m_action += b->sum_m_action;
    a->sum_m_full += b->sum_m_full; a->sum_m2_full += b->sum_m2_full;
    a->m_final += b->m_final; if (b->m_max > a->m_max) a->m_max = b->m_max; a->status |= b->status;
}

int eben_cv_grid(const double *basis, int n, int p, const double *y, const int32_t *fold_id,
                 int n_folds, const double *alpha, const double *lambda, int n_cells,
                 int prior, int epis, int n_threads, double *fold_err, eben_counters *cnt)
{
    /* split once per fold (the reference re-derives the same split inside every fit) */
    double **Xtr = (double **)calloc(n_folds, sizeof(double *)), **Xte = (double **)calloc(n_folds, sizeof(double *));
    double **ytr = (double **)calloc(n_folds, sizeof(double *)), **yte = (double **)calloc(n_folds, sizeof(double *));
    int *ntr = (int *)calloc(n_folds, sizeof(int)), *nte = (int *)calloc(n_folds, sizeof(int));
    for (int f = 0; f < n_folds; f++) {
        for (int i = 0; i < n; i++) { if (fold_id[i] == f + 1) nte[f]++; else ntr[f]++; }
        Xtr[f] = (double *)malloc(sizeof(double) * (size_t)ntr[f] * p);
        Xte[f] = (double *)malloc(sizeof(double) * (size_t)(nte[f] ? nte[f] : 1) * p);
        ytr[f] = (double *)malloc(sizeof(double) * ntr[f]);
        yte[f] = (double *)malloc(sizeof(double) * (nte[f] ? nte[f] : 1));
        int a = 0, b = 0;
        for (int i = 0; i < n; i++) { if (fold_id[i] == f + 1) yte[f][b++] = y[i]; else ytr[f][a++] = y[i]; }
        for (int j = 0; j < p; j++) {
            a = 0; b = 0;
            for (int i = 0; i < n; i++) {
                double v = basis[(size_t)j * n + i];
                if (fold_id[i] == f + 1) Xte[f][(size_t)j * nte[f] + b++] = v; else Xtr[f][(size_t)j * ntr[f] + a++] = v;
            }
        }
    }
    /* rows of the fit's Beta table: K (main effects), K(K+1)/2 (gaussian + epistasis: every implicit column), or the R
     * wrapper's bMax = 2K used-bases list of the binomial + epistasis fit (EBelasticNet.Binomial.R:7-9) */
    const long n_eff = (epis && prior == 1) ? 2L * p : (epis ? (long)p * (p + 1) / 2 : p);
    const int bcols = (epis && prior == 0) ? 5 : 4;
    int rc_all = 0;
    eben_counters total; memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    #pragma omp parallel for schedule(dynamic, 1)
    for (long u = 0; u < (long)n_cells * n_folds; u++) {
        int c = (int)(u / n_folds), f = (int)(u % n_folds);
        double *Beta = (double *)calloc((size_t)n_eff * bcols, sizeof(double));
        double wald = 0, icpt[2] = {0, 0}, resid = 0, ll = 0;
        eben_counters k; memset(&k, 0, sizeof(k));
        int rc;
        if (prior == 0 && !epis) rc = eben_gm_fit(Xtr[f], ytr[f], ntr[f], p, lambda[c], alpha[c], Beta, &wald, icpt, &resid, &k);
        else if (prior == 0)     rc = eben_gf_fit(Xtr[f], ytr[f], ntr[f], p, lambda[c], alpha[c], Beta, &wald, icpt, &resid, &k);
        else if (!epis)          rc = eben_bm_fit(Xtr[f], ytr[f], ntr[f], p, lambda[c], alpha[c], &ll, Beta, &wald, icpt, &k);
        else                     rc = eben_bf_fit(Xtr[f], ytr[f], ntr[f], p, lambda[c], alpha[c], &ll, Beta, (int)n_eff, &wald, icpt, &k);
        double err;
        if (prior == 0) err = fold_sse(Xte[f], yte[f], nte[f], p, Beta, n_eff, epis, icpt[0]);
        else            err = fold_loglik(Xte[f], yte[f], nte[f], p, Beta, n_eff, icpt[0]);
        fold_err[u] = err;
        free(Beta);
        #pragma omp critical
        { if (rc) rc_all = rc; add_counters(&total, &k); }
    }
    if (cnt) *cnt = total;
    for (int f = 0; f < n_folds; f++) { free(Xtr[f]); free(Xte[f]); free(ytr[f]); free(yte[f]); }
    free(Xtr); free(Xte); free(ytr); free(yte); free(ntr); free(nte);
    return rc_all;
}
