/*
 * pareben_glue.c -- .Call glue between R and libpareben_hip.so (include/pareben_hip.h).
 * Compiled only where R is installed (R CMD SHLIB / the package's src/); this build container
 * has no R, so this file is reviewed, not compiled, here.
 *
 * It replaces the body of the foreach loop in R/CrossValidate.R:66-70 (and :88-92): instead of
 * shipping BASIS to workers that each call EBEN's .C("elasticNetLinearNeMainEff"), R makes ONE
 * call and gets the nFolds x n_cells matrix of held-out errors back.
 *
 *   .Call("pareben_cv_grid_R", BASIS (double matrix n x p), Target (double n), foldId (int n),
 *         nFolds (int), alpha (double n_cells), lambda (double n_cells), epis (int), prior (int),
 *         device (int))
 *   -> list(fold_err = double matrix nFolds x n_cells, status = int matrix nFolds x n_cells)
 */
#include <R.h>
#include <Rinternals.h>
#include "pareben_hip.h"

SEXP pareben_cv_grid_R(SEXP basis, SEXP target, SEXP fold_id, SEXP n_folds, SEXP alpha, SEXP lambda,
                       SEXP epis, SEXP prior, SEXP device)
{
    if (!isReal(basis) || !isMatrix(basis)) error("BASIS must be a double matrix");
    if (!isReal(target) || !isInteger(fold_id) || !isReal(alpha) || !isReal(lambda)) error("bad argument types");
    const int n = nrows(basis), p = ncols(basis), nf = asInteger(n_folds), nc = LENGTH(alpha);
    if (LENGTH(target) != n || LENGTH(fold_id) != n || LENGTH(lambda) != nc) error("length mismatch");

    SEXP err = PROTECT(allocMatrix(REALSXP, nf, nc));       /* column c = cell c, row f = fold f+1 */
    SEXP st = PROTECT(allocMatrix(INTSXP, nf, nc));
    const int rc = pareben_cv_grid(REAL(basis), n, p, REAL(target), INTEGER(fold_id), nf,
                                   REAL(alpha), REAL(lambda), nc, asInteger(epis), asInteger(prior),
                                   asInteger(device), REAL(err), INTEGER(st), NULL);
    if (rc != PAREBEN_OK) { UNPROTECT(2); error("pareben_cv_grid failed (%d): %s", rc, pareben_last_error()); }
    SEXP out = PROTECT(allocVector(VECSXP, 2));
    SET_VECTOR_ELT(out, 0, err); SET_VECTOR_ELT(out, 1, st);
    SEXP nm = PROTECT(allocVector(STRSXP, 2));
    SET_STRING_ELT(nm, 0, mkChar("fold_err")); SET_STRING_ELT(nm, 1, mkChar("status"));
    setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(4);
    return out;
}

/* drop-in for .C("elasticNetLinearNeMainEff", ...) (EBEN_orig/R/EBelasticNet.Gaussian.R:39-51) */
SEXP pareben_fit_gaussian_R(SEXP basis, SEXP target, SEXP lambda, SEXP alpha, SEXP device)
{
    const int n = nrows(basis), k = ncols(basis);
    SEXP beta = PROTECT(allocMatrix(REALSXP, k, 4));
    double wald = 0, icpt = 0, resid = 0;
    const int rc = pareben_fit_gaussian(REAL(basis), REAL(target), asReal(lambda), asReal(alpha), REAL(beta),
                                        &wald, &icpt, n, k, 0, &resid, asInteger(device), NULL);
    if (rc != PAREBEN_OK) { UNPROTECT(1); error("pareben_fit_gaussian failed (%d): %s", rc, pareben_last_error()); }
    SEXP out = PROTECT(allocVector(VECSXP, 4));
    SET_VECTOR_ELT(out, 0, beta);
    SET_VECTOR_ELT(out, 1, ScalarReal(wald)); SET_VECTOR_ELT(out, 2, ScalarReal(icpt)); SET_VECTOR_ELT(out, 3, ScalarReal(resid));
    SEXP nm = PROTECT(allocVector(STRSXP, 4));
    SET_STRING_ELT(nm, 0, mkChar("Beta")); SET_STRING_ELT(nm, 1, mkChar("WaldScore"));
    SET_STRING_ELT(nm, 2, mkChar("Intercept")); SET_STRING_ELT(nm, 3, mkChar("residual"));
    setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(3);
    return out;
}

/* drop-in for .C("elasticNetLinearNeEpisEff", ...) (EBEN_orig/R/EBelasticNet.Gaussian.R:16-28): Beta K(K+1)/2 x 5 */
SEXP pareben_fit_gaussian_epis_R(SEXP basis, SEXP target, SEXP lambda, SEXP alpha, SEXP device)
{
    const int n = nrows(basis), k = ncols(basis);
    SEXP beta = PROTECT(allocMatrix(REALSXP, k * (k + 1) / 2, 5));
    double wald = 0, icpt = 0, resid = 0;
    const int rc = pareben_fit_gaussian_epis(REAL(basis), REAL(target), asReal(lambda), asReal(alpha), REAL(beta),
                                             &wald, &icpt, n, k, 0, &resid, asInteger(device), NULL);
    if (rc != PAREBEN_OK) { UNPROTECT(1); error("pareben_fit_gaussian_epis failed (%d): %s", rc, pareben_last_error()); }
    SEXP out = PROTECT(allocVector(VECSXP, 4));
    SET_VECTOR_ELT(out, 0, beta);
    SET_VECTOR_ELT(out, 1, ScalarReal(wald)); SET_VECTOR_ELT(out, 2, ScalarReal(icpt)); SET_VECTOR_ELT(out, 3, ScalarReal(resid));
    SEXP nm = PROTECT(allocVector(STRSXP, 4));
    SET_STRING_ELT(nm, 0, mkChar("Beta")); SET_STRING_ELT(nm, 1, mkChar("WaldScore"));
    SET_STRING_ELT(nm, 2, mkChar("Intercept")); SET_STRING_ELT(nm, 3, mkChar("residual"));
    setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(3);
    return out;
}

/* drop-in for .C("ElasticNetBinaryNEmainEff", ...) (EBEN_orig/R/EBelasticNet.Binomial.R:32-46) */
SEXP pareben_fit_binomial_R(SEXP basis, SEXP target, SEXP lambda, SEXP alpha, SEXP device)
{
    const int n = nrows(basis), k = ncols(basis);
    SEXP beta = PROTECT(allocMatrix(REALSXP, k, 4));
    SEXP icpt = PROTECT(allocVector(REALSXP, 2));
    double ll = 0, wald = 0;
    const int rc = pareben_fit_binomial(REAL(basis), REAL(target), asReal(lambda), asReal(alpha), &ll, REAL(beta),
                                        &wald, REAL(icpt), n, k, 0, k, asInteger(device), NULL);
    if (rc != PAREBEN_OK) { UNPROTECT(2); error("pareben_fit_binomial failed (%d): %s", rc, pareben_last_error()); }
    SEXP out = PROTECT(allocVector(VECSXP, 4));
    SET_VECTOR_ELT(out, 0, beta);
    SET_VECTOR_ELT(out, 1, ScalarReal(ll)); SET_VECTOR_ELT(out, 2, ScalarReal(wald)); SET_VECTOR_ELT(out, 3, icpt);
    SEXP nm = PROTECT(allocVector(STRSXP, 4));
    SET_STRING_ELT(nm, 0, mkChar("Beta")); SET_STRING_ELT(nm, 1, mkChar("logLikelihood"));
    SET_STRING_ELT(nm, 2, mkChar("WaldScore")); SET_STRING_ELT(nm, 3, mkChar("Intercept"));
    setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(4);
    return out;
}
