/*
 * pareben_hip.h -- C ABI of libpareben_hip.so: the MI355X (gfx950) replacement for the part of
 * parEBEN that CrossValidate() farms out through foreach: the nFolds x alpha x lambda grid of
 * EBelasticNet fits plus the per-fold scoring.
 *
 * Plain pointers and sizes only.  All matrices are column-major doubles exactly as R passes
 * them through .C()/.Call() (as.double(BASIS)); all index vectors are 32-bit ints; fold ids are
 * 1-based like R's.  Every function returns 0 on success or a negative PAREBEN_E* code; none
 * calls exit() or the R API, so they may be called from any host language (R .Call glue, ctypes,
 * cgo ...).  INTEGRATION.md shows the reference-side bindings.
 *
 * What each entry point replaces in the reference (paths relative to the parEBEN tree):
 *   pareben_cv_grid / pareben_ctx_run   the foreach over grid rows in R/CrossValidate.R:66-70 and
 *                                       :88-92, i.e. R/TestModel.R:6-39 (fold split, fit, score),
 *                                       EBEN_orig/R/EBelasticNet.Gaussian.R:39-66 (.C marshalling,
 *                                       non-zero-row filter) and R/GetModelError.R:6-59
 *   pareben_lambda_max_pairs            the Epis = "yes" double loop of GetLambdaMax, R/BuildGrid.R:21-30
 *   pareben_cv_grid_multi               the same foreach, its workers spread over the GPUs of one node
 *   pareben_fit_gaussian                .C("elasticNetLinearNeMainEff", ...) in
 *                                       EBEN_orig/R/EBelasticNet.Gaussian.R:39-51, i.e.
 *                                       EBEN_orig/src/elasticNetLinearNeMainEff.c:55
 *   pareben_fit_gaussian_epis           .C("elasticNetLinearNeEpisEff", ...) in
 *                                       EBEN_orig/R/EBelasticNet.Gaussian.R:16-28, i.e.
 *                                       EBEN_orig/src/elasticNetLinearNeFull2.c:57
 *   pareben_fit_binomial                .C("ElasticNetBinaryNEmainEff", ...) in
 *                                       EBEN_orig/R/EBelasticNet.Binomial.R:32-46, i.e.
 *                                       EBEN_orig/src/ElasticNetBinaryNEmainEff.c:236
 *   pareben_fit_binomial_epis           .C("ElasticNetBinaryNEfull", ...) in
 *                                       EBEN_orig/R/EBelasticNet.Binomial.R:6-26, i.e.
 *                                       EBEN_orig/src/ElasticNetBinaryNeFull.c:52
 */
#ifndef PAREBEN_HIP_H
#define PAREBEN_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAREBEN_OK            0
#define PAREBEN_EINVAL       -1   /* bad argument                                   */
#define PAREBEN_EHIP         -2   /* HIP runtime error (see pareben_last_error)     */
#define PAREBEN_ENOMEM       -3   /* device workspace does not fit                  */
#define PAREBEN_EUNSUPPORTED -4   /* not available in this build / on this machine   */

#define PAREBEN_PRIOR_GAUSSIAN 0
#define PAREBEN_PRIOR_BINOMIAL 1

/* per-fit status bits written to status[] */
#define PAREBEN_ST_OVERFLOW 1     /* active set grew past the reference's basisMax; with ST_ABORT: past the workspace too */
#define PAREBEN_ST_CHOLESKY 2     /* Hessian not positive definite                   */
#define PAREBEN_ST_STALE    4     /* reference's stale-slot delete path was taken    */
#define PAREBEN_ST_ABORT    8     /* fit stopped early; its fold_err entry is NaN     */

/* number of int64 counters per fit in counters[] (order: n_outer, n_inner, n_add, n_del,
 * n_reest, n_fullstat, sum_m_action, sum_m_full, sum_m2_full, m_final, m_max, status, mfma_tiles =
 * 16x16x16 tile products (8192 flop each) executed on the FP64 matrix cores for this fit, sum_m_swept = Gram rows
 * the action sweeps really read: sum_m_action less the sweeps a following full-stat pass made unnecessary) */
#define PAREBEN_NCOUNTERS 14

typedef struct pareben_ctx pareben_ctx;

/* Library / device probes. */
const char *pareben_version(void);
const char *pareben_last_error(void);
int pareben_device_count(void);

/*
 * Stage one CV problem in HBM: BASIS (n x p, column-major), Target (n), fold ids (1..n_folds,
 * what R/AssignToFolds.R:6-19 returns).  prior: gaussian (elasticNetLinearNeMainEff.c /
 * elasticNetLinearNeFull2.c) or binomial (ElasticNetBinaryNEmainEff.c).  epis: 0 = main effects,
 * 1 = add the p(p-1)/2 pairwise columns x_i*x_j in the reference's order (elasticNetLinearNeFull2.c /
 * ElasticNetBinaryNeFull.c).
 * Active-set capacity.  The reference sizes its arrays for basisMax = min(p, 1e7/p) columns
 * (elasticNetLinearNeMainEff.c:68-69; elasticNetLinearNeFull2.c:67-80 with epistasis) and runs off them when
 * a fit grows past that (:605-611: it prints "out of Memory" and continues).  Here such a fit is FLAGGED
 * (PAREBEN_ST_OVERFLOW) and continues in a workspace of max(basisMax, min(N_train, 2048)) columns (bounded by
 * p and 2048); only there it is stopped (PAREBEN_ST_OVERFLOW | PAREBEN_ST_ABORT, score NaN).  max_active > 0
 * lowers both limits; <= 0 picks the defaults.  The context owns every device buffer.
 */
int pareben_ctx_create(pareben_ctx **out, int device,
                       const double *basis, int n, int p, const double *target,
                       const int32_t *fold_id, int n_folds,
                       int prior, int epis, int max_active);

/*
 * Evaluate n_cells (alpha, lambda) cells x n_folds folds on the context's GPU.
 *   fold_err [n_cells * n_folds], cell-major: gaussian -> held-out SSE (R/GetModelError.R:30-31),
 *                                             binomial -> mean held-out log-likelihood (:55)
 *   status   [n_cells * n_folds] or NULL
 *   counters [n_cells * n_folds * PAREBEN_NCOUNTERS] or NULL
 * Inputs are already resident; the call launches the per-fold preparation kernels (row split,
 * column statistics, Gram matrices) and the fit kernel on the context's stream, then copies the
 * small result arrays back.
 * Gaussian prior, large p: when n_folds p x p Gram matrices do not fit in HBM beside the fit
 * workspaces, the context keeps a pool of Gram rows per fold instead and the fit kernel fills it on
 * first use (plus a few private rows per workgroup once a pool is full): same results to rounding,
 * one sweep of the fold's design per new row.  The environment variable PAREBEN_GRAM_ROWS=<rows
 * per fold>, read by pareben_ctx_create, forces that mode (diagnostics).
 * In the tail of a launch (work queue drained) workgroups without a fit take over feature tiles of the
 * full-stat passes and action mat-vecs of the fits still running; results are bit-identical either way.
 * PAREBEN_SHARE=0 (read by pareben_ctx_run) turns that off, 1 keeps it to the tail, 2 shares from the
 * start; unset = automatic (diagnostics / A-B timing).  PAREBEN_HEAVY_M=<active-set size> (default 384) is
 * the size from which a fit shares its phases whatever the state of the queue.
 */
int pareben_ctx_run(pareben_ctx *ctx, int n_cells, const double *alpha, const double *lambda,
                    double *fold_err, int32_t *status, int64_t *counters);

/* Timings of the last pareben_ctx_run, measured with HIP events on the context's stream:
 * ms[0] preparation kernels, ms[1] fit kernel, ms[2] whole call including result copies. */
int pareben_ctx_last_timing(pareben_ctx *ctx, double ms[3]);

/* Geometry of the last launch: info[0] workgroups, info[1] threads per workgroup,
 * info[2] active-set capacity of the workspaces, info[3] workspace KiB per workgroup,
 * info[4] the reference's basisMax (fits growing past it carry PAREBEN_ST_OVERFLOW). */
int pareben_ctx_launch_info(pareben_ctx *ctx, int64_t info[5]);

/* Test hook: run the per-fold preparation kernels and copy the normalised Gram matrix of fold
 * `fold` (0-based) to `out` (K x K doubles, out[u*K + i] = x_i.(x_u/|x_u|)/|x_i| over the fold's training
 * rows: row u is the reference's BASIS_PHI row of basis u, elasticNetLinearNeMainEff.c:1608-1630), K = p or
 * p(p+1)/2 with epistasis.  PAREBEN_EINVAL for the binomial prior or when the context keeps Gram rows on
 * demand instead of whole matrices. */
int pareben_ctx_gram(pareben_ctx *ctx, int fold, double *out);

int pareben_ctx_destroy(pareben_ctx *ctx);

/* One-shot convenience: create + run + destroy. */
int pareben_cv_grid(const double *basis, int n, int p, const double *target,
                    const int32_t *fold_id, int n_folds,
                    const double *alpha, const double *lambda, int n_cells,
                    int epis, int prior, int device,
                    double *fold_err, int32_t *status, int64_t *counters);

/*
 * The pairwise pass of GetLambdaMax (R/BuildGrid.R:21-30, the O(n p^2) double loop that dominates BuildGrid() with
 * Epis = "yes"): *out = max over pairs i < j of  (x_i*x_j / |x_i*x_j|) . (Target - mean(Target)),  -inf when p < 2
 * or every pair column is zero.  The caller combines it with the main-effect pass and the log(1.1) floor (:9-19).
 */
int pareben_lambda_max_pairs(const double *basis, int n, int p, const double *target, int device, double *out);

/*
 * The same grid on n_gpu devices of one node from ONE host process (the caller is a single R session;
 * reference call site: the foreach over grid rows, R/CrossValidate.R:66-70, whose workers were separate R
 * processes): one host thread and context per device, BASIS / Target / fold ids replicated; the (cell, fold)
 * units are not dealt out in advance -- every device's persistent fit kernel pulls from ONE cost-sorted queue
 * whose head sits in pinned, coherent host memory (system-scope atomic), as foreach's workers took the next
 * row -- and the path's only exchange is one grouped ncclAllGather (RCCL over xGMI; ncclCommInitAll
 * communicators, created on first use for a GPU count and kept) of the per-device result tables, after which
 * every GPU holds all of them and the host merges them from the first.  n_gpu <= 0 uses every visible device.
 * Results are bit-identical for any n_gpu and any timing (a fit's arithmetic does not depend on what else
 * runs or where).  With one GPU nothing is exchanged and RCCL is not touched; RCCL is bound at run time:
 * without librccl.so a call with n_gpu > 1 returns PAREBEN_EUNSUPPORTED and everything else still works.
 * The caller's current HIP device is restored on return.
 */
int pareben_cv_grid_multi(const double *basis, int n, int p, const double *target,
                          const int32_t *fold_id, int n_folds,
                          const double *alpha, const double *lambda, int n_cells,
                          int epis, int prior, int n_gpu,
                          double *fold_err, int32_t *status, int64_t *counters);

/* Communicators of pareben_cv_grid_multi are created on first use for a GPU count and kept; this frees them. */
int pareben_multi_release(void);
/* Last pareben_cv_grid_multi call: out[0] = ranks of the RCCL communicator (1 when one GPU ran and nothing was
 * exchanged), out[1] / out[2] = units pulled from the shared queue by the busiest / idlest GPU, out[3] = GPUs. */
int pareben_multi_last_stats(int64_t out[4]);

/*
 * One fit on all rows, same argument tuple as the reference's .C entry
 * (EBEN_orig/R/EBelasticNet.Gaussian.R:39-51): Beta is K x 4 column-major
 * (loc1, loc2, beta, posterior variance), outputs written in place.
 */
int pareben_fit_gaussian(const double *basis, const double *target, double lambda, double alpha,
                         double *Beta, double *wald, double *intercept, int n, int k,
                         int verbose, double *residual, int device, int64_t *counters);

/*
 * Same for Epis = "yes" (EBEN_orig/R/EBelasticNet.Gaussian.R:16-28): Beta is k(k+1)/2 x 5
 * column-major (loc1, loc2, beta, posterior variance, 1-based column id where used); rows are the k
 * main effects followed by the pairs (1,2),(1,3)..(k-1,k) (elasticNetLinearNeFull2.c:115-134).
 */
int pareben_fit_gaussian_epis(const double *basis, const double *target, double lambda, double alpha,
                              double *Beta, double *wald, double *intercept, int n, int k,
                              int verbose, double *residual, int device, int64_t *counters);

/*
 * One binomial fit on all rows, same argument tuple as the reference's .C entry
 * (EBEN_orig/R/EBelasticNet.Binomial.R:32-46): Beta is k x 4 column-major, intercept[0] = posterior
 * mean of the intercept, intercept[1] = its posterior variance (ElasticNetBinaryNEmainEff.c:374-375).
 * bMax is accepted for signature compatibility; the active-set capacity is min(k + 1, 1024).
 */
int pareben_fit_binomial(const double *basis, const double *target, double lambda, double alpha,
                         double *logLikelihood, double *Beta, double *wald, double *intercept,
                         int n, int k, int verbose, int bMax, int device, int64_t *counters);

/*
 * Same for Epis = "yes" (EBEN_orig/R/EBelasticNet.Binomial.R:6-26, .C("ElasticNetBinaryNEfull"), i.e.
 * EBEN_orig/src/ElasticNetBinaryNeFull.c:52): the k main effects plus the k(k-1)/2 pairwise columns.  Beta is
 * bMax x 4 column-major with bMax = 2k as the R wrapper passes it, and -- unlike the tables above -- lists the
 * USED bases in model order: (locus1, locus2, effect, posterior variance), zero rows after them
 * (ElasticNetBinaryNeFull.c:154-211); a model may hold at most bMax bases.
 */
int pareben_fit_binomial_epis(const double *basis, const double *target, double lambda, double alpha,
                              double *logLikelihood, double *Beta, double *wald, double *intercept,
                              int n, int k, int verbose, int bMax, int device, int64_t *counters);

/*
 * The reference's own .C entry points: the symbol names, argument order and pointer-only convention of EBEN_orig/src,
 * so R's .C() binds them with nothing but PACKAGE changed:
 *   elasticNetLinearNeMainEff   EBEN_orig/src/elasticNetLinearNeMainEff.c:55-57   <- EBEN_orig/R/EBelasticNet.Gaussian.R:38-51
 *   elasticNetLinearNeEpisEff   EBEN_orig/src/elasticNetLinearNeFull2.c:57-58     <- EBEN_orig/R/EBelasticNet.Gaussian.R:16-29
 *   ElasticNetBinaryNEmainEff   EBEN_orig/src/ElasticNetBinaryNEmainEff.c:236-238 <- EBEN_orig/R/EBelasticNet.Binomial.R:32-46
 *   ElasticNetBinaryNEfull      EBEN_orig/src/ElasticNetBinaryNeFull.c:52-55      <- EBEN_orig/R/EBelasticNet.Binomial.R:10-24
 * Each forwards to the pareben_fit_* entry above on device PAREBEN_DEVICE (default 0) and writes its outputs in place.
 * `verb` is honoured (stdout): > 0 basisMax, > 1 start / finish lines, > 2 one line per outer iteration
 * (elasticNetLinearNeMainEff.c:70-71, :196, :205; ElasticNetBinaryNEmainEff.c:327, :342, :352), > 4 one line per inner
 * iteration (:405, Gaussian).  .C has no error channel: on failure the message goes to stderr and the scalar outputs are NaN.
 */
void elasticNetLinearNeMainEff(double *BASIS, double *y, double *a_lambda, double *b_Alpha, double *Beta,
                               double *wald, double *intercept, int *n, int *kdim, int *verb, double *residual);
void elasticNetLinearNeEpisEff(double *BASIS, double *y, double *a_lambda, double *b_Alpha, double *Beta,
                               double *wald, double *intercept, int *n, int *kdim, int *VB, double *residual);
void ElasticNetBinaryNEmainEff(double *BASIS, double *Targets, double *a_Lambda, double *b_Alpha, double *logLIKELIHOOD,
                               double *Beta, double *wald, double *intercept, int *n, int *kdim, int *VB, int *bMax);
void ElasticNetBinaryNEfull(double *BASIS, double *Targets, double *a_Lambda, double *b_Alpha, double *logLIKELIHOOD,
                            double *Beta, double *wald, double *intercept, int *n, int *kdim, int *VB, int *bMax);

/*
 * Diagnostics: decision trace of the following pareben_fit_gaussian[_epis] calls of the calling thread.  One record of
 * 16 64-bit words per inner iteration of elasticNetLinearNeMainEff.c:401-731 (the arg-max feature, its action and dML,
 * the runner-up, the block cut-off and the nearest dML to it, the noise precision and order-free XOR hashes of S_in,
 * Q_in, (Sigma, mu) afterwards; layout TR_* in pareben_amd/csrc/types.h).  buf[0] = records written, records start at
 * buf + 16; at most max_records are kept.  NULL switches it off.  oracle/eben_gm.c writes the same layout
 * (eben_set_trace), which is how tools/trace_divergence.py finds the first decision on which two builds part.
 */
int pareben_set_trace(uint64_t *buf, int64_t max_records);

#ifdef __cplusplus
}
#endif
#endif
