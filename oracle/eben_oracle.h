/*
 * eben_oracle.h -- CPU restatement of the EBEN fit kernels that parEBEN's CrossValidate()
 * drives.  TEST INFRASTRUCTURE ONLY: nothing under pareben_amd/ may include, link or call
 * this; it is the checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).
 *
 * Pinning:
 *   Gm (eben_gm.c, main effects): PINNED to the reference itself -- the authors' stored real-R output
 *     paper_materials/Real Data Analysis/10000_Features/LooserSubset_10000_ParCV_5-3-2018.RDS (R 3.5.0 + CRAN
 *     EBEN) is reproduced to 1e-15 on every fit tried; tests/test_oracle_golden.py::test_oracle_reproduces_real_r_fit
 *     keeps one such fit in the suite.  The R-level pieces (grid, folds, summary) are pinned by the same file.
 *     Second pin, whole fit outputs: the stored EBelasticNet.Gaussian results under paper_materials/Real Data
 *     Analysis/Full_Test (R 3.5 + CRAN EBEN, Nov/Dec 2018) -- feature list, effects, posterior variances, WaldScore,
 *     Intercept, residVar -- are reproduced to 2e-11 / 1e-14 (tests/test_oracle_golden.py::test_oracle_reproduces_real_r_refits).
 *     And the optimum cell of the second stored CV table (parEBENoutput_2018-08-15*.RDS, K = 13 248) to 4e-16
 *     (profiles/r02/oracle_vs_real_r_fits.json).
 *   Gf (eben_gm.c with the epistasis variant), Bm and Bf (eben_bm.c): PARITY UNPINNED.  The reference tree holds no
 *     output of an epistasis or a binomial fit, its C cannot be built in this image (it needs R's headers and
 *     BLAS/LAPACK), and the known answers in SURVEY.md section 10 came from a build behind stand-in headers, which
 *     does not count as a pin.  These restatements follow the cited reference lines and agree with the independent
 *     device-side restatement (tests), nothing more.
 */
#ifndef EBEN_ORACLE_H
#define EBEN_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-fit event counters (SURVEY.md 8(d) accounting) */
typedef struct {
    int64_t n_outer;      /* outer (intercept) iterations                           */
    int64_t n_inner;      /* inner iterations, all outer iterations together        */
    int64_t n_add, n_del, n_reest;
    int64_t n_fullstat;   /* full S/Q recomputations                                */
    int64_t sum_m_action; /* sum over add/delete/re-estimate events of M at event   */
    int64_t sum_m_full;   /* sum over full-stat calls of M                          */
    int64_t sum_m2_full;  /* sum over full-stat calls of M*M                        */
    int64_t m_final;      /* active-set size at exit                                */
    int64_t m_max;        /* largest active set seen                                */
    int64_t status;       /* 0 ok; bit0 active set hit capacity; bit1 Cholesky failed;
                             bit2 stale-index delete path (reference UB) taken      */
} eben_counters;

/* Capacity policy of the Gaussian fits (eben_gm.c): default (0, 0) = the reference's basisMax stops a fit;
 * (1, r) = flag (status bit 0) and continue up to max(basisMax, min(N, 2048)) like the HIP build, with
 * basisMax lowered to r when r > 0.  Process-wide; set before fitting. */
void eben_set_capacity_policy(int continue_past_basismax, int ref_cap_override);

/* Decision trace of the next Gaussian fit(s) (diagnostics, single-threaded use): 16 64-bit words per inner iteration
 * (layout: eben_gm.c, TR_*), buf[0] = records written, records from buf + 16; NULL switches it off. */
void eben_set_trace(uint64_t *buf, int64_t max_records);

/* Gaussian, main effects.  Follows EBEN_orig/src/elasticNetLinearNeMainEff.c:55-242.
 * X is N x K column-major, Beta is K x 4 column-major (loc1, loc2, beta, var). */
int eben_gm_fit(const double *X, const double *y, int N, int K, double lambda, double alpha,
                double *Beta, double *wald, double *intercept, double *residual,
                eben_counters *cnt);

/* Gaussian, main + pairwise epistasis.  Follows EBEN_orig/src/elasticNetLinearNeFull2.c:57-261.
 * Beta is K(K+1)/2 x 5 column-major (loc1, loc2, beta, var, used). */
int eben_gf_fit(const double *X, const double *y, int N, int K, double lambda, double alpha,
                double *Beta, double *wald, double *intercept, double *residual,
                eben_counters *cnt);

/* Binomial, main effects.  Follows EBEN_orig/src/ElasticNetBinaryNEmainEff.c:236-389.
 * Beta is K x 4 column-major; intercept[2] = (mu0, sigma00). */
int eben_bm_fit(const double *X, const double *y, int N, int K, double lambda, double alpha,
                double *loglik, double *Beta, double *wald, double *intercept,
                eben_counters *cnt);

/* Binomial, main + pairwise epistasis.  Follows EBEN_orig/src/ElasticNetBinaryNeFull.c:52-232.
 * Beta is bMax x 4 column-major: the used bases in model order (locus1, locus2, beta, var); R passes bMax = 2K. */
int eben_bf_fit(const double *X, const double *y, int N, int K, double lambda, double alpha,
                double *loglik, double *Beta, int bMax, double *wald, double *intercept,
                eben_counters *cnt);

/* Whole CV grid: for each cell c and fold f (1..n_folds) fit on rows fold_id != f and score the
 * rows fold_id == f exactly as R/TestModel.R:6-39 + R/GetModelError.R:6-59 do.
 * prior: 0 gaussian (fold SSE), 1 binomial (mean log-lik).  epis: 0/1.
 * fold_err[c*n_folds + (f-1)].  n_threads <= 0 -> all cores (OpenMP). */
int eben_cv_grid(const double *basis, int n, int p, const double *y, const int32_t *fold_id,
                 int n_folds, const double *alpha, const double *lambda, int n_cells,
                 int prior, int epis, int n_threads, double *fold_err, eben_counters *cnt);

#ifdef __cplusplus
}
#endif
#endif
