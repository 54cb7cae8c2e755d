/*
 * eben_linalg.h -- the few dense kernels the oracle needs, in plain sequential loops
 * (netlib reference-BLAS accumulation order).  TEST INFRASTRUCTURE ONLY.
 *
 * The reference reaches these through R's F77_CALL(ddot|dgemv|dgemm|dpotrf|dpotri|dgelsy)
 * (SURVEY.md 8(a) a23); BLAS/LAPACK are not vendored in the reference tree, so their
 * published (netlib) algorithms are restated here: unblocked Cholesky (dpotf2), triangular
 * inverse (dtrti2) and U*U' (dlauu2) for the 'U' storage the reference asks for.
 */
#ifndef EBEN_LINALG_H
#define EBEN_LINALG_H
#include <math.h>
#include <stddef.h>

static inline double dot_seq(int n, const double *a, const double *b)
{
#ifdef EBEN_BLAS_ORDER_UNROLL4
    /* Sensitivity experiment only (liboracle_blas4.so): the accumulation order of an optimised
     * BLAS ddot (four running partial sums) instead of netlib's single one -- what the
     * reference computes when R is linked against such a BLAS.  Never used as the checker. */
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = 0;
    for (; i + 3 < n; i += 4) { s0 += a[i] * b[i]; s1 += a[i + 1] * b[i + 1]; s2 += a[i + 2] * b[i + 2]; s3 += a[i + 3] * b[i + 3]; }
    for (; i < n; i++) s0 += a[i] * b[i];
    return (s0 + s1) + (s2 + s3);
#else
    double s = 0;
    for (int i = 0; i < n; i++) s = s + a[i] * b[i];
    return s;
#endif
}

/* unbiased variance: elasticNetLinearNeMainEff.c:1826-1838 (same routine in every kernel) */
static inline double var_unbiased(const double *v, int n)
{
    double m = 0, s = 0;
    for (int i = 0; i < n; i++) m = m + v[i];
    m = m / n;
    for (int i = 0; i < n; i++) s = s + pow(v[i] - m, 2);
    return s / (n - 1);
}

/* In-place inverse of the SPD n x n matrix a (column-major, leading dimension n) from its
 * upper triangle: a = U'U (dpotf2 'U'), U <- U^-1 (dtrti2), a <- U U' (dlauu2), then the lower
 * triangle is mirrored from the upper one as MatrixInverseGmNeEN does (:1363-1368).
 * Returns 1 when a pivot is not positive; the matrix is then left partly factorised, which is
 * what the reference carries on with. */
static inline int chol_inverse_upper(double *a, int n)
{
#define AT(r, c) a[(size_t)(c) * n + (r)]
    for (int j = 0; j < n; j++) {
        double d = AT(j, j);
        for (int k = 0; k < j; k++) d -= AT(k, j) * AT(k, j);
        if (!(d > 0)) return 1;
        d = sqrt(d);
        AT(j, j) = d;
        for (int c = j + 1; c < n; c++) {
            double s = AT(j, c);
            for (int k = 0; k < j; k++) s -= AT(k, j) * AT(k, c);
            AT(j, c) = s / d;
        }
    }
    for (int j = 0; j < n; j++) {
        AT(j, j) = 1.0 / AT(j, j);
        double ajj = -AT(j, j);
        /* x = T * x with T the inverted leading j x j block (upper, non-unit), x = column j */
        for (int c = 0; c < j; c++) {
            double xc = AT(c, j);
            if (xc != 0) {
                for (int r = 0; r < c; r++) AT(r, j) += xc * AT(r, c);
                AT(c, j) = xc * AT(c, c);
            }
        }
        for (int r = 0; r < j; r++) AT(r, j) *= ajj;
    }
    for (int i = 0; i < n; i++) {
        double aii = AT(i, i);
        if (i < n - 1) {
            double d = 0;
            for (int k = i; k < n; k++) d += AT(i, k) * AT(i, k);
            AT(i, i) = d;
            for (int r = 0; r < i; r++) {
                double s = 0;
                for (int k = i + 1; k < n; k++) s += AT(r, k) * AT(i, k);
                AT(r, i) = s + aii * AT(r, i);
            }
        } else {
            for (int r = 0; r <= i; r++) AT(r, i) *= aii;
        }
    }
    for (int i = 1; i < n; i++)
        for (int j = 0; j < i; j++) AT(i, j) = AT(j, i);
#undef AT
    return 0;
}

#endif
