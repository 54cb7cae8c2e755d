/*
 * pareben_glue.c -- .Call glue between R and libpareben_hip.so (include/pareben_hip.h).
 * Compiled only where R is installed (R CMD INSTALL builds it from the package's src/ with the flags in
 * r/Makevars.snippet); this build container has no R, so this file is reviewed, not compiled, here.
 *
 * It replaces the body of the foreach loop in R/CrossValidate.R:66-70 (and :88-92): instead of
 * shipping BASIS to workers that each call EBEN's .C("elasticNetLinearNeMainEff"), R makes ONE
 * call and gets the nFolds x n_cells matrix of held-out errors back.
 *
 *   .Call(pareben_cv_grid_R, BASIS (double matrix n x p), Target (double n), foldId (int n),
 *         nFolds (int), alpha (double n_cells), lambda (double n_cells), epis (int), prior (int),
 *         nGPU (int: 1 = one device, `device`; 0 = all visible; k = the first k), device (int))
 *   -> list(fold_err = double matrix nFolds x n_cells, status = int matrix nFolds x n_cells)
 *
 * All entry points are registered (R_init_parEBEN), so the R side calls them as symbols, not strings.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include "pareben_hip.h"

static void check_design(SEXP basis, SEXP target)
{
    if (!isReal(basis) || !isMatrix(basis)) error("BASIS must be a double matrix (storage.mode(BASIS) <- \"double\")");
    if (!isReal(target) || LENGTH(target) != nrows(basis)) error("Target must be a double vector with nrow(BASIS) entries");
}

SEXP pareben_cv_grid_R(SEXP basis, SEXP target, SEXP fold_id, SEXP n_folds, SEXP alpha, SEXP lambda,
                       SEXP epis, SEXP prior, SEXP n_gpu, SEXP device)
{
    check_design(basis, target);
    if (!isInteger(fold_id) || !isReal(alpha) || !isReal(lambda)) error("bad argument types");
    const int n = nrows(basis), p = ncols(basis), nf = asInteger(n_folds), nc = LENGTH(alpha), ng = asInteger(n_gpu);
    if (LENGTH(fold_id) != n || LENGTH(lambda) != nc) error("length mismatch");

    SEXP err = PROTECT(allocMatrix(REALSXP, nf, nc));       /* column c = cell c, row f = fold f+1 */
    SEXP st = PROTECT(allocMatrix(INTSXP, nf, nc));
    int rc;
    if (ng == 1)
        rc = pareben_cv_grid(REAL(basis), n, p, REAL(target), INTEGER(fold_id), nf, REAL(alpha), REAL(lambda), nc,
                             asInteger(epis), asInteger(prior), asInteger(device), REAL(err), INTEGER(st), NULL);
    else      /* one host thread + context per GPU inside the library, one RCCL all-gather (R stays single-threaded) */
        rc = pareben_cv_grid_multi(REAL(basis), n, p, REAL(target), INTEGER(fold_id), nf, REAL(alpha), REAL(lambda), nc,
                                   asInteger(epis), asInteger(prior), ng, REAL(err), INTEGER(st), NULL);
    if (rc != PAREBEN_OK) { UNPROTECT(2); error("pareben_cv_grid failed (%d): %s", rc, pareben_last_error()); }
    SEXP out = PROTECT(allocVector(VECSXP, 2));
    SET_VECTOR_ELT(out, 0, err); SET_VECTOR_ELT(out, 1, st);
    SEXP nm = PROTECT(allocVector(STRSXP, 2));
    SET_STRING_ELT(nm, 0, mkChar("fold_err")); SET_STRING_ELT(nm, 1, mkChar("status"));
    setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(4);
    return out;
}

/* the Epis = "yes" double loop of GetLambdaMax (R/BuildGrid.R:21-30) -> max over pairs, -Inf if none */
SEXP pareben_lambda_max_pairs_R(SEXP basis, SEXP target, SEXP device)
{
    check_design(basis, target);
    double out = 0;
    const int rc = pareben_lambda_max_pairs(REAL(basis), nrows(basis), ncols(basis), REAL(target), asInteger(device), &out);
    if (rc != PAREBEN_OK) error("pareben_lambda_max_pairs failed (%d): %s", rc, pareben_last_error());
    return ScalarReal(out);
}

static SEXP named_list4(SEXP a, SEXP b, SEXP c, SEXP d, const char *na, const char *nb, const char *nc, const char *nd)
{
    SEXP out = PROTECT(allocVector(VECSXP, 4));
    SET_VECTOR_ELT(out, 0, a); SET_VECTOR_ELT(out, 1, b); SET_VECTOR_ELT(out, 2, c); SET_VECTOR_ELT(out, 3, d);
    SEXP nm = PROTECT(allocVector(STRSXP, 4));
    SET_STRING_ELT(nm, 0, mkChar(na)); SET_STRING_ELT(nm, 1, mkChar(nb)); SET_STRING_ELT(nm, 2, mkChar(nc)); SET_STRING_ELT(nm, 3, mkChar(nd));
    setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(2);
    return out;
}

/* drop-in for .C("elasticNetLinearNeMainEff", ...) (EBEN_orig/R/EBelasticNet.Gaussian.R:39-51): Beta K x 4;
 * epis != 0: .C("elasticNetLinearNeEpisEff", ...) (:16-28): Beta K(K+1)/2 x 5 */
SEXP pareben_fit_gaussian_R(SEXP basis, SEXP target, SEXP lambda, SEXP alpha, SEXP epis, SEXP device)
{
    check_design(basis, target);
    const int n = nrows(basis), k = ncols(basis), ep = asInteger(epis);
    SEXP beta = PROTECT(ep ? allocMatrix(REALSXP, k * (k + 1) / 2, 5) : allocMatrix(REALSXP, k, 4));
    double wald = 0, icpt = 0, resid = 0;
    const int rc = ep ? pareben_fit_gaussian_epis(REAL(basis), REAL(target), asReal(lambda), asReal(alpha), REAL(beta),
                                                  &wald, &icpt, n, k, 0, &resid, asInteger(device), NULL)
                      : pareben_fit_gaussian(REAL(basis), REAL(target), asReal(lambda), asReal(alpha), REAL(beta),
                                             &wald, &icpt, n, k, 0, &resid, asInteger(device), NULL);
    if (rc != PAREBEN_OK) { UNPROTECT(1); error("pareben_fit_gaussian failed (%d): %s", rc, pareben_last_error()); }
    SEXP w = PROTECT(ScalarReal(wald)), i = PROTECT(ScalarReal(icpt)), r = PROTECT(ScalarReal(resid));
    SEXP out = named_list4(beta, w, i, r, "Beta", "WaldScore", "Intercept", "residual");
    UNPROTECT(4);
    return out;
}

/* drop-in for .C("ElasticNetBinaryNEmainEff", ...) (EBEN_orig/R/EBelasticNet.Binomial.R:32-46): Beta K x 4;
 * epis != 0: .C("ElasticNetBinaryNEfull", ...) (:10-25): Beta 2K x 4, the used bases in model order */
SEXP pareben_fit_binomial_R(SEXP basis, SEXP target, SEXP lambda, SEXP alpha, SEXP epis, SEXP device)
{
    check_design(basis, target);
    const int n = nrows(basis), k = ncols(basis), ep = asInteger(epis);
    SEXP beta = PROTECT(allocMatrix(REALSXP, ep ? 2 * k : k, 4));
    SEXP icpt = PROTECT(allocVector(REALSXP, 2));
    double ll = 0, wald = 0;
    const int rc = ep ? pareben_fit_binomial_epis(REAL(basis), REAL(target), asReal(lambda), asReal(alpha), &ll, REAL(beta),
                                                  &wald, REAL(icpt), n, k, 0, 2 * k, asInteger(device), NULL)
                      : pareben_fit_binomial(REAL(basis), REAL(target), asReal(lambda), asReal(alpha), &ll, REAL(beta),
                                             &wald, REAL(icpt), n, k, 0, k, asInteger(device), NULL);
    if (rc != PAREBEN_OK) { UNPROTECT(2); error("pareben_fit_binomial failed (%d): %s", rc, pareben_last_error()); }
    SEXP l = PROTECT(ScalarReal(ll)), w = PROTECT(ScalarReal(wald));
    SEXP out = named_list4(beta, l, w, icpt, "Beta", "logLikelihood", "WaldScore", "Intercept");
    UNPROTECT(4);
    return out;
}

SEXP pareben_device_count_R(void) { return ScalarInteger(pareben_device_count()); }

static const R_CallMethodDef call_methods[] = {
    {"pareben_cv_grid_R",          (DL_FUNC)&pareben_cv_grid_R,          10},
    {"pareben_lambda_max_pairs_R", (DL_FUNC)&pareben_lambda_max_pairs_R,  3},
    {"pareben_fit_gaussian_R",     (DL_FUNC)&pareben_fit_gaussian_R,      6},
    {"pareben_fit_binomial_R",     (DL_FUNC)&pareben_fit_binomial_R,      6},
    {"pareben_device_count_R",     (DL_FUNC)&pareben_device_count_R,      0},
    {NULL, NULL, 0}
};

void R_init_parEBEN(DllInfo *dll)
{
    R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
    R_forceSymbols(dll, TRUE);          /* .Call(pareben_cv_grid_R, ...) with useDynLib(parEBEN, .registration = TRUE) */
}
