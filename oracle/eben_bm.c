/*
 * eben_bm.c -- oracle: binomial (logistic) EBEN fits, a CPU restatement of the algorithms in
 * EBEN_orig/src/ElasticNetBinaryNEmainEff.c (main effects, "Bm") and
 * EBEN_orig/src/ElasticNetBinaryNeFull.c (main effects + pairwise epistasis, "Bf").
 *
 * Bf is Bm's skeleton over K(K+1)/2 implicit columns (x_i, then x_i*x_j in the order (1,2),(1,3)..)
 * with these differences, each cited where it is applied (`epis` switches them on):
 *   block cut-off 0.99 instead of 0.90 (NeFull.c:254); phi = column / scale by division (:437-450, :755;
 *   Bm multiplies by the reciprocal, NEmainEff.c:644-646, :1349-1350); pair columns are regenerated inside
 *   every sweep with the reference's association ((x_i*phi)*w)*x_j (:925-931), (x_i*w)*x_j (:1471, :1583,
 *   :1738), (x_i*x_j)*(w*phi) (:1356); Newton step: y clamped to [1e-5, 1-1e-5] before the residual, weights
 *   < 1e-5 -> 1e-3 and > 1e5 -> 1e3 (:1042-1048), stop when ALL M gradient entries are small (:1085-1099);
 *   delete downdates Sigma as S - (s_i/s_jj)*s_j (:1606); delete-priority only for more than 100 bases
 *   (:1805-1809); the outer stopping sum runs over the M-1 precisions (:133-134); capacity bMax = 2K from
 *   the R wrapper (EBelasticNet.Binomial.R:7-9), checked as N_used+1 > bMax (:549-553); output = the used
 *   bases in model order, bMax x 4 (:154-211).
 *
 * TEST INFRASTRUCTURE ONLY (see eben_oracle.h).  PARITY UNPINNED: no reference-held output of a binomial fit exists
 * to check this file against (eben_oracle.h, "Pinning").  Own data structures (state struct, 0-based
 * feature ids), the reference's arithmetic order (sequential sums, -ffp-contract=off) and its
 * quirks: Q1 first basis is column 0; Q12 the outer stopping sum reads one slot past the active
 * precisions; Q15 add-priority never fires; the re-estimate S/Q update reads the already
 * updated Sigma row (:1166 before :1191); the Newton loop keeps the y of a failed line search.
 *
 * Model layout as in the reference: M = N_used + 1, slot 0 of Mu/Sigma/H/PHI is the intercept
 * (PHI column of ones), slot l+1 belongs to used[l]; Alpha has N_used entries.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "eben_oracle.h"
#include "eben_linalg.h"

enum { ACT_NONE = -10, ACT_REEST = 0, ACT_ADD = 1, ACT_DEL = -1, ACT_TERM = 10 };

typedef struct {
    int N, K, cap;             /* K = number of candidate columns: P, or P(P+1)/2 with epistasis */
    int epis, P;               /* P = columns of X                                              */
    int bmax;                  /* epistasis: the R wrapper's bMax = 2P                          */
    int *fi, *fj;              /* epistasis: column f is x_fi (fj < 0) or x_fi * x_fj           */
    const double *X, *y;
    double lambda, alpha;
    double *scale;
    int *used, n_used;
    int *unused, n_unused;
    double *A;                 /* cap+1, zero-initialised, never cleared (Q12)        */
    double *mu;                /* cap+1                                                */
    double *Sig, *H;           /* (cap+1)^2, leading dimension = current M            */
    double *Phi;               /* N x (cap+1), column 0 = 1                            */
    double *w;                 /* N  IRLS weights y(1-y), clamped                      */
    double *x2;                /* N x K  squared design (basisCache, :468-478)         */
    double *Sin, *Qin, *Sout, *Qout, *dml, *aroot;
    int *act, *todo;
    eben_counters c;
} bm;

#define M_OF(s) ((s)->n_used + 1)

static void sigmoid_vec(double *y, const double *pm, int n)
{
    for (int i = 0; i < n; i++) y[i] = 1 / (1 + exp(-pm[i]));
}

/* :2013-2025 (also refreshes y) */
static double data_error(double *y, const double *pm, const double *t, int n)
{
    double e = 0;
    sigmoid_vec(y, pm, n);
    for (int i = 0; i < n; i++) {
        if (y[i] != 0) e = e - t[i] * log(y[i]);
        if (y[i] != 1) e = e - (1 - t[i]) * log(1 - y[i]);
    }
    return e;
}

/* candidate column f at sample h (used only where the reference forms the product this way) */
static double col_at(const bm *s, int f, int h)
{
    const int N = s->N;
    if (!s->epis || s->fj[f] < 0) return s->X[(size_t)(s->epis ? s->fi[f] : f) * N + h];
    return s->X[(size_t)s->fi[f] * N + h] * s->X[(size_t)s->fj[f] * N + h];
}

static void phi_mu(const bm *s, const double *mu, int M, double *out)
{
    const int N = s->N;
    for (int i = 0; i < N; i++) {
        double a = 0;
        for (int j = 0; j < M; j++) a = a + s->Phi[(size_t)j * N + i] * mu[j];
        out[i] = a;
    }
}

/* ---- posterior mode by damped Newton / IRLS, :1808-2010 */
static void bm_postmode(bm *s)
{
    const int N = s->N, M = M_OF(s);
    const double step_min = 1 / pow(2.0, 8);
    double *pm = (double *)calloc(N, sizeof(double)), *y = (double *)calloc(N, sizeof(double));
    double *e = (double *)calloc(N, sizeof(double)), *g = (double *)calloc(M, sizeof(double));
    double *dmu = (double *)calloc(M, sizeof(double)), *mnew = (double *)calloc(M, sizeof(double));
    double elog[25];
    phi_mu(s, s->mu, M, pm);
    double derr = data_error(y, pm, s->y, N);
    double reg = 0;
    for (int i = 1; i < M; i++) reg = reg + s->A[i - 1] * s->mu[i] * s->mu[i] / 2;
    double total = reg + derr;
    for (int it = 0; it < 25; it++) {
        elog[it] = total;
        double g0 = 0, h0 = 0;
        if (s->epis) for (int j = 0; j < N; j++) {         /* NeFull.c:1042-1043 */
            if (y[j] < 1e-5) y[j] = 1e-5;
            if (y[j] > (1 - 1e-5)) y[j] = 1 - 1e-5;
        }
        for (int j = 0; j < N; j++) { e[j] = s->y[j] + -1.0 * y[j]; }
        for (int j = 0; j < N; j++) g0 = g0 + 1.0 * e[j];
        for (int j = 0; j < N; j++) {
            double b = y[j] * (1 - y[j]);
            if (s->epis) { if (b < 1e-5) b = 1e-3; if (b > 1e5) b = 1e3; }      /* NeFull.c:1047-1048 */
            else { if (b < 1e-10) b = 1e-5; if (b > 1e10) b = 1e5; }            /* NEmainEff.c:1878-1883 */
            s->w[j] = b;
        }
        for (int j = 0; j < N; j++) h0 = h0 + 1.0 * s->w[j];
        g[0] = g0; s->H[0] = h0;
        for (int j = 1; j < M; j++) {
            const double *ph = s->Phi + (size_t)j * N;
            g[j] = dot_seq(N, e, ph);
            s->H[j] = dot_seq(N, s->w, ph);
            g[j] = g[j] - s->A[j - 1] * s->mu[j];
            s->H[(size_t)j * M] = s->H[j];
        }
        for (int j = 1; j < M; j++)
            for (int k = 1; k < M; k++) {
                const double *pj = s->Phi + (size_t)j * N, *pk = s->Phi + (size_t)k * N;
                double a = 0;
                for (int L = 0; L < N; L++) a = a + pj[L] * s->w[L] * pk[L];
                if (j == k) a = a + s->A[k - 1];
                s->H[(size_t)k * M + j] = a;
            }
        memcpy(s->Sig, s->H, sizeof(double) * (size_t)M * M);
        if (chol_inverse_upper(s->Sig, M)) s->c.status |= 2;
        int cnt = 0;
        for (int j = s->epis ? 0 : 1; j < M; j++) if (fabs(g[j]) < 1e-6) cnt++;   /* NeFull.c:1085-1099 counts the intercept too */
        if (cnt == (s->epis ? M : M - 1)) break;
        for (int k = 0; k < M; k++) {
            double a = 0;
            for (int L = 0; L < M; L++) a = a + g[L] * s->Sig[(size_t)L * M + k];
            dmu[k] = a;
        }
        double step = 1;
        while (step > step_min) {
            for (int j = 0; j < M; j++) mnew[j] = s->mu[j] + step * dmu[j];
            phi_mu(s, mnew, M, pm);
            derr = data_error(y, pm, s->y, N);
            reg = 0;
            for (int j = 1; j < M; j++) reg = reg + s->A[j - 1] * mnew[j] * mnew[j] / 2;
            total = derr + reg;
            if (total >= elog[it]) step = step / 2;
            else { memcpy(s->mu, mnew, sizeof(double) * M); step = 0; }
        }
        if (step == 1) break;
    }
    free(pm); free(y); free(e); free(g); free(dmu); free(mnew);
}

/* ---- full statistics, :1633-1803 */
static void bm_fullstat(bm *s)
{
    const int N = s->N, K = s->K;
    bm_postmode(s);
    const int M = M_OF(s);
    double *pm = (double *)calloc(N, sizeof(double)), *y = (double *)calloc(N, sizeof(double));
    double *e = (double *)calloc(N, sizeof(double));
    double *bp = (double *)calloc(M, sizeof(double)), *tmp = (double *)calloc(M, sizeof(double));
    phi_mu(s, s->mu, M, pm);
    sigmoid_vec(y, pm, N);
    for (int i = 0; i < N; i++) e[i] = s->y[i] + -1.0 * y[i];
    for (int i = 0; i < K; i++) {
        const int ci = s->epis ? s->fi[i] : i, cj = s->epis ? s->fj[i] : -1;
        const double *x = s->X + (size_t)ci * N;
        double bb = 0, ze = 0;
        if (cj < 0) {
            for (int p = 0; p < M; p++) {
                const double *ph = s->Phi + (size_t)p * N;
                double a = 0;
                for (int j = 0; j < N; j++) a = a + x[j] * ph[j] * s->w[j];
                bp[p] = a / s->scale[i];
            }
            bb = dot_seq(N, s->w, s->x2 + (size_t)ci * N);
            ze = dot_seq(N, x, e);
        } else {                                        /* pair column, NeFull.c:921-957 */
            const double *xl = s->X + (size_t)cj * N;
            for (int p = 0; p < M; p++) {
                const double *ph = s->Phi + (size_t)p * N;
                double a = 0;
                for (int j = 0; j < N; j++) a = a + x[j] * ph[j] * s->w[j] * xl[j];
                bp[p] = a / s->scale[i];
            }
            for (int p = 0; p < N; p++) {
                bb = bb + s->w[p] * s->x2[(size_t)ci * N + p] * s->x2[(size_t)cj * N + p];
                ze = ze + x[p] * e[p] * xl[p];
            }
        }
        for (int p = 0; p < M; p++) tmp[p] = dot_seq(M, s->Sig + (size_t)p * M, bp);
        double quad = dot_seq(M, tmp, bp);
        s->Sin[i] = bb / (s->scale[i] * s->scale[i]) - quad;
        s->Qin[i] = ze / s->scale[i];
        s->Sout[i] = s->Sin[i];
        s->Qout[i] = s->Qin[i];
    }
    for (int i = 0; i < s->n_used; i++) {
        int f = s->used[i];
        s->Sout[f] = s->A[i] * s->Sin[f] / (s->A[i] - s->Sin[f]);
        s->Qout[f] = s->A[i] * s->Qin[f] / (s->A[i] - s->Sin[f]);
    }
    free(pm); free(y); free(e); free(bp); free(tmp);
    s->c.n_fullstat++; s->c.sum_m_full += M; s->c.sum_m2_full += (int64_t)M * M;
}

/* ---- dML / action choice, :2063-2238 (Q15: any_add stays 0) */
static int bm_delta_ml(bm *s, int *any_del, double *best)
{
    const int K = s->K, NU = s->n_used, N = s->N;
    const double l1 = s->lambda * s->alpha, l2 = s->lambda * (1 - s->alpha);
    const int any_add = 0;
    int prio_add = 0, prio_del = 0;
    *any_del = 0;
    if (NU < 10) { prio_add = 1; prio_del = 0; }
    if (NU > 100 || (!s->epis && NU >= N)) { prio_add = 0; prio_del = 1; }   /* NeFull.c:1805-1809 has no N clause */
    for (int i = 0; i < K; i++) s->act[i] = ACT_NONE;
    double dmax = 0; int imax = 0;
    for (int i = 0; i < NU; i++) {
        int f = s->used[i];
        double so = s->Sout[f], qo = s->Qout[f];
        s->dml[f] = 0;
        double a = so - qo * qo + 2 * l1 + l2;
        double b = (so + l2) * (so + 4 * l1 + l2);
        double g = 2 * l1 * (so + l2) * (so + l2);
        double d = b * b - 4 * a * g;
        if (a < 0 && d > 0) {
            double r = (-b - sqrt(d)) / (2 * a);
            double L = (log(r / (r + so + l2)) + pow(qo, 2) / (r + so + l2)) * 0.5 - l1 / r;
            if (L > 0) {
                s->aroot[f] = r + l2;
                s->act[f] = ACT_REEST;
                double o = s->A[i] - l2;
                s->dml[f] = 0.5 * (log(r * (o + so + l2) / (o * (r + so + l2))) +
                                   qo * qo * (1 / (r + so + l2) - 1 / (o + so + l2))) -
                            l1 * (1 / r - 1 / o);
            }
        } else if (NU > 1) {
            *any_del = 1;
            s->act[f] = ACT_DEL;
            double o = s->A[i] - l2;
            double L = (log(o / (o + so + l2)) + pow(qo, 2) / (o + so + l2)) * 0.5 - l1 / o;
            s->dml[f] = -L;
        }
        if (s->dml[f] > dmax) { imax = f; dmax = s->dml[f]; }
    }
    for (int i = 0; i < s->n_unused; i++) {
        int f = s->unused[i];
        double so = s->Sout[f], qo = s->Qout[f];
        s->dml[f] = 0;
        double a = so - qo * qo + 2 * l1 + l2;
        double b = (so + l2) * (so + l2 + 4 * l1);
        double g = 2 * l1 * (so + l2) * (so + l2);
        double d = b * b - 4 * a * g;
        if (a < 0 && d > 0) {
            double r = (-b - sqrt(d)) / (2 * a);
            double L = (log(r / (r + so + l2)) + pow(qo, 2) / (r + so + l2)) * 0.5 - l1 / r;
            if (L > 0) { s->aroot[f] = r + l2; s->act[f] = ACT_ADD; s->dml[f] = L; }
        }
        if (s->dml[f] > dmax) { imax = f; dmax = s->dml[f]; }
    }
    if ((any_add && prio_add) || (*any_del && prio_del)) {
        for (int i = 0; i < K; i++) {
            if (s->act[i] == ACT_REEST) s->dml[i] = 0;
            else if (s->act[i] == ACT_DEL) { if (any_add && prio_add && !prio_del) s->dml[i] = 0; }
            else if (s->act[i] == ACT_ADD) { if (*any_del && prio_del && !prio_add) s->dml[i] = 0; }
        }
        dmax = 0; imax = 0;
        for (int i = 0; i < K; i++) if (s->dml[i] > dmax) { imax = i; dmax = s->dml[i]; }
    }
    *best = dmax;
    return imax;
}

/* x_i .* w against every model column, / scale_i: the 1 x M row the reference recomputes for
 * every feature in every action (:967-977, :1040-1050, :1171-1181) */
static void weighted_row(const bm *s, int i, int M, double *out)
{
    const int N = s->N;
    const int ci = s->epis ? s->fi[i] : i, cj = s->epis ? s->fj[i] : -1;
    const double *x = s->X + (size_t)ci * N;
    const double *xl = cj >= 0 ? s->X + (size_t)cj * N : NULL;
    for (int j = 0; j < M; j++) {
        const double *ph = s->Phi + (size_t)j * N;
        double a = 0;
        if (!xl) for (int h = 0; h < N; h++) a = a + (x[h] * s->w[h]) * ph[h];
        else     for (int h = 0; h < N; h++) a = a + (x[h] * s->w[h] * xl[h]) * ph[h];   /* NeFull.c:1471, :1583, :1738 */
        out[j] = a / s->scale[i];
    }
}

/* ---- add, :830-1003 + :701-711 */
static void bm_add(bm *s, int nu, double newA, const double *phi)
{
    const int N = s->N, K = s->K, M = M_OF(s), M1 = M + 1, NU = s->n_used;
    double *bphi = (double *)calloc(N, sizeof(double)), *bb = (double *)calloc(K, sizeof(double));
    double *tmp = (double *)calloc(M, sizeof(double)), *tp = (double *)calloc(M, sizeof(double));
    double *si = (double *)calloc(M, sizeof(double)), *row = (double *)calloc(M, sizeof(double));
    double *SN = (double *)calloc((size_t)M1 * M1, sizeof(double));
    for (int j = 0; j < N; j++) bphi[j] = s->w[j] * phi[j];
    for (int i = 0; i < K; i++) {
        const int ci = s->epis ? s->fi[i] : i, cj = s->epis ? s->fj[i] : -1;
        const double *x = s->X + (size_t)ci * N;
        double a = 0;
        if (cj < 0) for (int h = 0; h < N; h++) a = a + x[h] * bphi[h];
        else { const double *xl = s->X + (size_t)cj * N; for (int h = 0; h < N; h++) a = a + x[h] * xl[h] * bphi[h]; }   /* NeFull.c:1356 */
        bb[i] = a / s->scale[i];
    }
    for (int i = 0; i < M; i++) tmp[i] = dot_seq(N, s->Phi + (size_t)i * N, bphi);
    for (int i = 0; i < M; i++) tp[i] = dot_seq(M, s->Sig + (size_t)i * M, tmp);
    s->A[NU] = newA;
    memcpy(s->Phi + (size_t)M * N, phi, sizeof(double) * N);
    double sii = 1.0 / (newA + s->Sin[nu]);
    double mui = sii * s->Qin[nu];
    for (int i = 0; i < M; i++) s->mu[i] += -mui * tp[i];
    s->mu[M] = mui;
    for (int i = 0; i < M; i++) si[i] = tp[i] * -sii;
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++) SN[(size_t)j * M1 + i] = s->Sig[(size_t)j * M + i] + -si[i] * tp[j];
    for (int i = 0; i < M; i++) { SN[(size_t)M * M1 + i] = si[i]; SN[(size_t)i * M1 + M] = si[i]; }
    SN[(size_t)M * M1 + M] = sii;
    memcpy(s->Sig, SN, sizeof(double) * (size_t)M1 * M1);
    for (int i = 0; i < K; i++) {
        weighted_row(s, i, M, row);
        double t = dot_seq(M, row, tp);
        double mc = bb[i] - t;
        s->Sin[i] = s->Sin[i] - mc * mc * sii;
        s->Qin[i] = s->Qin[i] - mui * mc;
    }
    s->used[NU] = nu;
    s->n_used = NU + 1;
    s->n_unused--;
    for (int i = 0; i < s->n_unused; i++) if (s->unused[i] == nu) s->unused[i] = s->unused[s->n_unused];
    free(bphi); free(bb); free(tmp); free(tp); free(si); free(row); free(SN);
}

/* ---- delete used slot jj, :1010-1121 + :728-744 */
static void bm_delete(bm *s, int jj, int nu)
{
    const int N = s->N, K = s->K, M = M_OF(s), last = M - 1, j1 = jj + 1;
    double *T = (double *)calloc((size_t)M * M, sizeof(double)), *SN = (double *)calloc((size_t)last * last + 1, sizeof(double));
    double *row = (double *)calloc(M, sizeof(double));
    double *Sg = s->Sig;
    const double sjj = Sg[(size_t)j1 * M + j1];
    const double mujj = s->mu[j1];
    for (int i = 0; i < M; i++) s->mu[i] = s->mu[i] - mujj * Sg[(size_t)j1 * M + i] / sjj;
    for (int i = 0; i < K; i++) {
        weighted_row(s, i, M, row);
        double t = 0;
        for (int j = 0; j < M; j++) t = t + row[j] * Sg[(size_t)j1 * M + j];
        s->Sin[i] = s->Sin[i] + pow(t, 2) / sjj;
        s->Qin[i] = s->Qin[i] + t * mujj / sjj;
    }
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++)
            T[(size_t)j * M + i] = s->epis ? Sg[(size_t)j * M + i] - Sg[(size_t)j1 * M + i] / sjj * Sg[(size_t)j1 * M + j]      /* NeFull.c:1606 */
                                           : Sg[(size_t)j * M + i] - Sg[(size_t)j1 * M + i] * Sg[(size_t)j1 * M + j] / sjj;     /* NEmainEff.c:1069 */
    for (int i = 0; i < last; i++)
        for (int j = 0; j < last; j++) SN[(size_t)j * last + i] = T[(size_t)j * M + i];
    if (j1 != last) {
        s->A[jj] = s->A[last - 1];
        s->mu[j1] = s->mu[last];
        memcpy(s->Phi + (size_t)j1 * N, s->Phi + (size_t)last * N, sizeof(double) * N);
        for (int i = 0; i < last; i++) SN[(size_t)j1 * last + i] = T[(size_t)last * M + i];
        T[(size_t)j1 * M + M - 1] = T[(size_t)M * M - 1];
        for (int c = 0; c < last; c++) SN[(size_t)c * last + j1] = T[(size_t)c * M + M - 1];
    }
    memcpy(Sg, SN, sizeof(double) * (size_t)last * last);
    s->used[jj] = s->used[s->n_used - 1];
    s->n_used--;
    s->n_unused++;
    s->unused[s->n_unused - 1] = nu;
    free(T); free(SN); free(row);
}

/* ---- re-estimate used slot jj, :1127-1203 (S/Q update reads the NEW Sigma row) */
static void bm_reestimate(bm *s, int jj, double newA)
{
    const int K = s->K, M = M_OF(s), j1 = jj + 1;
    double *SN = (double *)calloc((size_t)M * M, sizeof(double)), *row = (double *)calloc(M, sizeof(double));
    double *Sg = s->Sig;
    const double oldA = s->A[jj];
    s->A[jj] = newA;
    const double dinv = 1.0 / (newA - oldA);
    const double kappa = 1.0 / (Sg[(size_t)j1 * M + j1] + dinv);
    const double mujj = s->mu[j1];
    const double f = -mujj * kappa;
    for (int i = 0; i < M; i++) s->mu[i] += f * Sg[(size_t)j1 * M + i];
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++)
            SN[(size_t)j * M + i] = Sg[(size_t)j * M + i] - kappa * Sg[(size_t)j1 * M + i] * Sg[(size_t)j1 * M + j];
    memcpy(Sg, SN, sizeof(double) * (size_t)M * M);
    for (int i = 0; i < K; i++) {
        weighted_row(s, i, M, row);
        double t = 0;
        for (int j = 0; j < M; j++) t = t + row[j] * Sg[(size_t)j1 * M + j];
        s->Sin[i] = s->Sin[i] + pow(t, 2) * kappa;
        s->Qin[i] = s->Qin[i] + mujj * kappa * t;
    }
    free(SN); free(row);
}

/* ---- first model, :1215-1406: intercept + column 0 (Q1), weights from a 2-column least-squares
 * fit of logit(0.05 / 0.95) pseudo-targets (dgelsy, :1366-1376) */
static void bm_initialise(bm *s, int first)
{
    const int N = s->N, K = s->K;
    if (first) {
        s->n_used = 1;
        s->used[0] = 0;
        for (int i = 0; i < N; i++) s->Phi[i] = 1;
        double r = 1 / s->scale[0];
        if (s->epis) for (int i = 0; i < N; i++) s->Phi[N + i] = s->X[i] / s->scale[0];     /* NeFull.c:755 */
        else         for (int i = 0; i < N; i++) s->Phi[N + i] = s->X[i] * r;               /* NEmainEff.c:1349-1350 */
        double sa = 0, sb = 0, sc = 0, sd = 0;
        for (int i = 0; i < N; i++) {
            double tp = -1 + 2 * s->y[i];
            double lo = log(((tp * 0.9 + 1) / 2) / (1 - (tp * 0.9 + 1) / 2));
            double ph = s->Phi[N + i];
            sa += ph; sb += ph * ph; sc += lo; sd += ph * lo;
        }
        double det = (double)N * sb - sa * sa;
        if (fabs(det) > 1e-10 * N * (sb > 0 ? sb : 1)) {
            s->mu[0] = (sb * sc - sa * sd) / det;
            s->mu[1] = ((double)N * sd - sa * sc) / det;
        } else {                                   /* rank 1: minimum-norm solution */
            double c0 = sa / N, den = (double)N * (1 + c0 * c0);
            s->mu[0] = sc / den; s->mu[1] = c0 * sc / den;
        }
        if (s->mu[1] == 0) s->A[0] = 1; else s->A[0] = 1 / (s->mu[1] * s->mu[1]);
        if (s->A[0] < 1e-3) s->A[0] = 1e-3;
        if (s->A[0] > 1e3) s->A[0] = 1e3;
    }
    int kk = 0;
    for (int i = 0; i < K; i++) {
        int is_used = 0;
        for (int j = 0; j < s->n_used; j++) if (s->used[j] == i) is_used = 1;
        if (!is_used) s->unused[kk++] = i;
    }
    s->n_unused = K - s->n_used;
}

/* ---- one call of the inner routine, :397-827 */
static int bm_inner(bm *s, int iter, double *loglik)
{
    const int N = s->N, K = s->K;
    bm_initialise(s, iter <= 1);
    const int initial = s->used[0];
    int ini_removed = iter <= 1 ? 0 : 1;
    bm_fullstat(s);
    int sel = ACT_NONE, jj = -1, n_todo = 0, any_del = 0, last_it = 0, i_iter = 0;
    const int it_max = iter == 1 ? 10 : 100;
    double ll = 1e-30, ll0;
    double *phi = (double *)calloc(N, sizeof(double)), *pm = (double *)calloc(N, sizeof(double));
    while (!last_it) {
        i_iter++;
        s->c.n_inner++;
        ll0 = ll;
        double best;
        int nu = bm_delta_ml(s, &any_del, &best);
        int M = M_OF(s), worthwhile;
        if (sel == ACT_TERM && !ini_removed && M > 2) nu = -1;
        if (nu == -1 && ini_removed) { worthwhile = 0; sel = ACT_TERM; }
        else if (nu == -1 && !ini_removed && M > 2) {
            worthwhile = 1; nu = initial; s->act[nu] = ACT_DEL; n_todo = 1; s->todo[0] = initial;
            ini_removed = 1; sel = ACT_DEL;
        } else {
            worthwhile = 1;
            double cutoff = best * (s->act[nu] == ACT_ADD ? (s->epis ? 0.99 : 0.90) : 1.0);      /* NeFull.c:254 | NEmainEff.c:422 */
            if (cutoff < 0.001) cutoff = 0.001;
            n_todo = 0;
            for (int i = 0; i < K; i++) if (s->dml[i] >= cutoff) s->todo[n_todo++] = i;
            if (s->act[nu] == ACT_DEL && n_todo > 1) n_todo = 1;
            if (n_todo == 0) worthwhile = 0;
        }
        if (!worthwhile) sel = ACT_TERM;
        if (worthwhile) {
            for (int u = 0; u < n_todo; u++) {
                nu = s->todo[u];
                sel = s->act[nu];
                double newA = s->aroot[nu];
                if (sel == ACT_REEST || sel == ACT_DEL) {
                    int found = 0;
                    for (int i = 0; i < s->n_used; i++) if (s->used[i] == nu) { jj = i; found = 1; }
                    if (!found) { s->c.status |= 4; if (jj < 0 || jj >= s->n_used) { free(phi); free(pm); return 1; } }
                }
                double r = 1.0 / s->scale[nu];
                if (s->epis) for (int h = 0; h < N; h++) phi[h] = col_at(s, nu, h) / s->scale[nu];   /* NeFull.c:437-450 */
                else         for (int h = 0; h < N; h++) phi[h] = s->X[(size_t)nu * N + h] * r;
                if (sel == ACT_REEST && fabs(log(newA) - log(s->A[jj])) <= 1e-3 && any_del == 0) sel = ACT_TERM;
                if (sel == ACT_REEST) {
                    s->c.n_reest++; s->c.sum_m_action += M_OF(s);
                    bm_reestimate(s, jj, newA);
                } else if (sel == ACT_ADD) {
                    if (s->n_used + 2 > s->cap) { s->c.status |= 1; free(phi); free(pm); return 1; }
                    /* NeFull.c:549-553 (the reference tests this from the second outer iteration on and overruns its arrays in the first) */
                    if (s->epis && s->n_used + 1 > s->bmax) { s->c.status |= 1; free(phi); free(pm); return 1; }
                    s->c.n_add++; s->c.sum_m_action += M_OF(s);
                    bm_add(s, nu, newA, phi);
                } else if (sel == ACT_DEL) {
                    s->c.n_del++; s->c.sum_m_action += M_OF(s);
                    bm_delete(s, jj, nu);
                    if (nu == initial) ini_removed = 1;
                }
                if (M_OF(s) > s->c.m_max) s->c.m_max = M_OF(s);
                if (u == n_todo - 1) bm_fullstat(s);            /* :749-762 */
            }
        }
        M = M_OF(s);
        if (sel == ACT_TERM && ini_removed) last_it = 1;
        if ((i_iter == it_max && M == 2) || i_iter > it_max) last_it = 1;
        if (i_iter == it_max) sel = ACT_TERM;
        phi_mu(s, s->mu, M, pm);
        ll = 0;
        for (int i = 0; i < N; i++)
            ll = ll + s->y[i] * log(exp(pm[i]) / (1 + exp(pm[i]))) + (1 - s->y[i]) * log(1 / (1 + exp(pm[i])));
        double dL = fabs((ll - ll0) / ll0);
        if (dL < 1e-3) sel = ACT_TERM;
    }
    *loglik = ll;
    free(phi); free(pm);
    return 0;
}

static void bm_alloc(bm *s, int cap)
{
    const int N = s->N, K = s->K;
    s->cap = cap;
    s->used = (int *)calloc(cap, sizeof(int)); s->unused = (int *)calloc((size_t)K + 1, sizeof(int));
    s->A = (double *)calloc(cap + 1, sizeof(double)); s->mu = (double *)calloc(cap + 1, sizeof(double));
    s->Sig = (double *)calloc((size_t)(cap + 1) * (cap + 1), sizeof(double));
    s->H = (double *)calloc((size_t)(cap + 1) * (cap + 1), sizeof(double));
    s->Phi = (double *)calloc((size_t)N * (cap + 1), sizeof(double));
    s->w = (double *)calloc(N, sizeof(double));
    s->Sin = (double *)calloc(K, sizeof(double)); s->Qin = (double *)calloc(K, sizeof(double));
    s->Sout = (double *)calloc(K, sizeof(double)); s->Qout = (double *)calloc(K, sizeof(double));
    s->dml = (double *)calloc(K, sizeof(double)); s->aroot = (double *)calloc(K, sizeof(double));
    s->act = (int *)calloc(K, sizeof(int)); s->todo = (int *)calloc(K, sizeof(int));
    s->n_used = 1;
}

static void bm_free(bm *s)
{
    free(s->scale); free(s->x2); free(s->used); free(s->unused); free(s->A); free(s->mu); free(s->Sig); free(s->H);
    free(s->Phi); free(s->w); free(s->Sin); free(s->Qin); free(s->Sout); free(s->Qout); free(s->dml); free(s->aroot);
    free(s->act); free(s->todo); free(s->fi); free(s->fj);
}

/* outer loop: NEmainEff.c:329-344 / NeFull.c:121-137 */
static int bm_outer(bm *s, double *ll)
{
    double vk = 1e-30, vk0, err = 1000;
    int iter = 0, rc = 0;
    while (iter < 100 && err > 1e-8) {
        iter++;
        vk0 = vk;
        rc = bm_inner(s, iter, ll);
        if (rc) break;
        const int M = M_OF(s);
        vk = 0;
        if (s->epis) for (int i = 0; i < M - 1; i++) vk = vk + s->A[i];   /* NeFull.c:133-134: the M-1 precisions */
        else         for (int i = 0; i < M; i++) vk += fabs(s->A[i]);     /* dasum over M = N_used+1 (Q12) */
        err = fabs(vk - vk0) / M;
    }
    s->c.n_outer = iter;
    return rc;
}

int eben_bm_fit(const double *X, const double *y, int N, int K, double lambda, double alpha,
                double *loglik, double *Beta, double *wald, double *intercept, eben_counters *cnt)
{
    bm S; memset(&S, 0, sizeof(S));
    bm *s = &S;
    s->N = N; s->K = K; s->P = K; s->X = X; s->y = y; s->lambda = lambda; s->alpha = alpha;
    s->scale = (double *)calloc(K, sizeof(double));
    s->x2 = (double *)calloc((size_t)N * K, sizeof(double));
    for (int i = 0; i < K; i++) {
        Beta[i] = i + 1; Beta[K + i] = i + 1; Beta[2 * (size_t)K + i] = 0; Beta[3 * (size_t)K + i] = 0;
        double q = dot_seq(N, X + (size_t)i * N, X + (size_t)i * N);
        if (q == 0) q = 1;
        s->scale[i] = sqrt(q);
        for (int j = 0; j < N; j++) s->x2[(size_t)i * N + j] = X[(size_t)i * N + j] * X[(size_t)i * N + j];
    }
    bm_alloc(s, K + 2);                             /* R passes bMax = K (EBelasticNet.Binomial.R:28,45) */
    double ll = 0;
    int rc = bm_outer(s, &ll);
    {
        const int M = M_OF(s);
        double *tw = (double *)calloc(M, sizeof(double));
        for (int i = 0; i < M; i++) tw[i] = dot_seq(M, s->H + (size_t)i * M, s->mu);
        *wald = dot_seq(M, tw, s->mu);
        free(tw);
        for (int i = 1; i < M; i++) {
            int f = s->used[i - 1];
            Beta[2 * (size_t)K + f] = s->mu[i] / s->scale[f];
            Beta[3 * (size_t)K + f] = s->Sig[(size_t)i * M + i] / (s->scale[f] * s->scale[f]);
        }
        intercept[0] = s->mu[0];
        intercept[1] = s->Sig[0];
    }
    *loglik = ll;
    s->c.m_final = s->n_used;
    if (cnt) *cnt = s->c;
    bm_free(s);
    return rc;
}

/* Binomial, main effects + pairwise epistasis: EBEN_orig/src/ElasticNetBinaryNeFull.c:52-232.
 * Beta is bMax x 4 column-major and lists the USED bases in model order (locus1, locus2, effect, variance),
 * zero rows after them (:66, :154-211); the R wrapper passes bMax = 2K (EBelasticNet.Binomial.R:7-9, :25). */
int eben_bf_fit(const double *X, const double *y, int N, int K, double lambda, double alpha,
                double *loglik, double *Beta, int bMax, double *wald, double *intercept, eben_counters *cnt)
{
    bm S; memset(&S, 0, sizeof(S));
    bm *s = &S;
    const long MF = (long)K * (K + 1) / 2;
    s->N = N; s->K = (int)MF; s->P = K; s->epis = 1; s->bmax = bMax;
    s->X = X; s->y = y; s->lambda = lambda; s->alpha = alpha;
    s->scale = (double *)calloc(MF, sizeof(double));
    s->x2 = (double *)calloc((size_t)N * K, sizeof(double));     /* basisCache: squares of the K main columns only (:291-296) */
    s->fi = (int *)calloc(MF, sizeof(int)); s->fj = (int *)calloc(MF, sizeof(int));
    for (int i = 0; i < 4 * bMax; i++) Beta[i] = 0;               /* as allocated by R; the reference clears column 3 (:66) */
    for (int i = 0; i < K; i++) {                                 /* :79-88 */
        double q = 0;
        for (int l = 0; l < N; l++) { q = q + X[(size_t)i * N + l] * X[(size_t)i * N + l]; s->x2[(size_t)i * N + l] = X[(size_t)i * N + l] * X[(size_t)i * N + l]; }
        if (q == 0) q = 1;
        s->scale[i] = sqrt(q);
        s->fi[i] = i; s->fj[i] = -1;
    }
    long kk = K;
    for (int i = 0; i < K - 1; i++)                               /* :90-105 */
        for (int j = i + 1; j < K; j++) {
            double q = 0;
            const double *xi = X + (size_t)i * N, *xj = X + (size_t)j * N;
            for (int l = 0; l < N; l++) q = q + xi[l] * xi[l] * xj[l] * xj[l];
            if (q == 0) q = 1;
            s->scale[kk] = sqrt(q);
            s->fi[kk] = i; s->fj[kk] = j;
            kk++;
        }
    bm_alloc(s, bMax + 2);
    double ll = 0;
    int rc = bm_outer(s, &ll);
    {
        const int M = M_OF(s);
        double *tw = (double *)calloc(M, sizeof(double));
        *wald = 0;
        for (int i = 0; i < M; i++) {                             /* :146-153 */
            tw[i] = 0;
            for (int j = 0; j < M; j++) tw[i] = tw[i] + s->mu[j] * s->H[(size_t)i * M + j];
            *wald = *wald + tw[i] * s->mu[i];
        }
        free(tw);
        int meff = M - 1;
        if (M > bMax) meff = bMax;                                /* :163-167 */
        for (int i = 0; i < meff; i++) {
            const int f = s->used[i];
            Beta[i] = s->fi[f] + 1;                               /* the reference decodes (locus1, locus2) from the column id, :172-199 */
            Beta[bMax + i] = (s->fj[f] < 0 ? s->fi[f] : s->fj[f]) + 1;
            Beta[2 * (size_t)bMax + i] = s->mu[i + 1] / s->scale[f];
            Beta[3 * (size_t)bMax + i] = s->Sig[(size_t)(i + 1) * M + i + 1] / (s->scale[f] * s->scale[f]);
        }
        intercept[0] = s->mu[0];
        intercept[1] = s->Sig[0];
    }
    *loglik = ll;
    s->c.m_final = s->n_used;
    if (cnt) *cnt = s->c;
    bm_free(s);
    return rc;
}
