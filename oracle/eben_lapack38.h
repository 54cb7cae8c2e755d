/*
 * eben_lapack38.h -- TEST INFRASTRUCTURE ONLY.  dpotrf + dpotri ('U') exactly as LAPACK 3.8.0 (the version R 3.5.0 bundles)
 * computes them on top of the reference BLAS: the BLOCKED algorithms (block size 64 from ILAENV) with the recursive dpotrf2 on
 * the diagonal blocks, and the reference BLAS-3 loop orders of dsyrk / dgemm / dtrsm / dtrmm.  The unblocked restatement in
 * eben_linalg.h sums in another order once the matrix is larger than 64, which is enough to decide last-bit ties on the
 * duplicated-column designs of the stored real-R tables differently from R ("Parity on chaotic fits").
 * LAPACK is not vendored in the reference tree (it links R's); the published algorithms are restated here.
 * All matrices column-major, A(i,j) = a[i + j*lda], 0-based.
 */
#ifndef EBEN_LAPACK38_H
#define EBEN_LAPACK38_H
#include <math.h>
#include <stddef.h>

#define L38_NB 64
#define L38(a, lda, i, j) (a)[(size_t)(i) + (size_t)(j) * (lda)]

/* DSYRK('U','T'): C(n x n, upper) := alpha A'A + beta C, A is k x n */
static void l38_dsyrk_ut(int n, int k, double alpha, const double *A, int lda, double beta, double *C, int ldc)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i <= j; i++) {
            double temp = 0;
            for (int l = 0; l < k; l++) temp = temp + L38(A, lda, l, i) * L38(A, lda, l, j);
            if (beta == 0) L38(C, ldc, i, j) = alpha * temp;
            else L38(C, ldc, i, j) = alpha * temp + beta * L38(C, ldc, i, j);
        }
}
/* DSYRK('U','N'): C(n x n, upper) := alpha A A' + beta C, A is n x k */
static void l38_dsyrk_un(int n, int k, double alpha, const double *A, int lda, double beta, double *C, int ldc)
{
    for (int j = 0; j < n; j++) {
        if (beta == 0) for (int i = 0; i <= j; i++) L38(C, ldc, i, j) = 0;
        else if (beta != 1) for (int i = 0; i <= j; i++) L38(C, ldc, i, j) = beta * L38(C, ldc, i, j);
        for (int l = 0; l < k; l++)
            if (L38(A, lda, j, l) != 0) {
                const double temp = alpha * L38(A, lda, j, l);
                for (int i = 0; i <= j; i++) L38(C, ldc, i, j) = L38(C, ldc, i, j) + temp * L38(A, lda, i, l);
            }
    }
}
/* DGEMM('T','N'): C(m x n) := alpha A'B + beta C, A is k x m, B is k x n */
static void l38_dgemm_tn(int m, int n, int k, double alpha, const double *A, int lda, const double *B, int ldb, double beta, double *C, int ldc)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++) {
            double temp = 0;
            for (int l = 0; l < k; l++) temp = temp + L38(A, lda, l, i) * L38(B, ldb, l, j);
            if (beta == 0) L38(C, ldc, i, j) = alpha * temp;
            else L38(C, ldc, i, j) = alpha * temp + beta * L38(C, ldc, i, j);
        }
}
/* DGEMM('N','T'): C(m x n) := alpha A B' + beta C, A is m x k, B is n x k */
static void l38_dgemm_nt(int m, int n, int k, double alpha, const double *A, int lda, const double *B, int ldb, double beta, double *C, int ldc)
{
    for (int j = 0; j < n; j++) {
        if (beta == 0) for (int i = 0; i < m; i++) L38(C, ldc, i, j) = 0;
        else if (beta != 1) for (int i = 0; i < m; i++) L38(C, ldc, i, j) = beta * L38(C, ldc, i, j);
        for (int l = 0; l < k; l++) {
            const double temp = alpha * L38(B, ldb, j, l);
            for (int i = 0; i < m; i++) L38(C, ldc, i, j) = L38(C, ldc, i, j) + temp * L38(A, lda, i, l);
        }
    }
}
/* DTRSM('L','U','T','N'): B(m x n) := alpha inv(A') B, A upper m x m */
static void l38_dtrsm_lutn(int m, int n, double alpha, const double *A, int lda, double *B, int ldb)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++) {
            double temp = alpha * L38(B, ldb, i, j);
            for (int k = 0; k < i; k++) temp = temp - L38(A, lda, k, i) * L38(B, ldb, k, j);
            temp = temp / L38(A, lda, i, i);
            L38(B, ldb, i, j) = temp;
        }
}
/* DTRSM('R','U','N','N'): B(m x n) := alpha B inv(A), A upper n x n */
static void l38_dtrsm_runn(int m, int n, double alpha, const double *A, int lda, double *B, int ldb)
{
    for (int j = 0; j < n; j++) {
        if (alpha != 1) for (int i = 0; i < m; i++) L38(B, ldb, i, j) = alpha * L38(B, ldb, i, j);
        for (int k = 0; k < j; k++)
            if (L38(A, lda, k, j) != 0)
                for (int i = 0; i < m; i++) L38(B, ldb, i, j) = L38(B, ldb, i, j) - L38(A, lda, k, j) * L38(B, ldb, i, k);
        const double temp = 1.0 / L38(A, lda, j, j);
        for (int i = 0; i < m; i++) L38(B, ldb, i, j) = temp * L38(B, ldb, i, j);
    }
}
/* DTRMM('L','U','N','N'): B(m x n) := alpha A B, A upper m x m */
static void l38_dtrmm_lunn(int m, int n, double alpha, const double *A, int lda, double *B, int ldb)
{
    for (int j = 0; j < n; j++)
        for (int k = 0; k < m; k++)
            if (L38(B, ldb, k, j) != 0) {
                double temp = alpha * L38(B, ldb, k, j);
                for (int i = 0; i < k; i++) L38(B, ldb, i, j) = L38(B, ldb, i, j) + temp * L38(A, lda, i, k);
                temp = temp * L38(A, lda, k, k);
                L38(B, ldb, k, j) = temp;
            }
}
/* DTRMM('R','U','T','N'): B(m x n) := alpha B A', A upper n x n */
static void l38_dtrmm_rutn(int m, int n, double alpha, const double *A, int lda, double *B, int ldb)
{
    for (int k = 0; k < n; k++) {
        for (int j = 0; j < k; j++)
            if (L38(A, lda, j, k) != 0) {
                const double temp = alpha * L38(A, lda, j, k);
                for (int i = 0; i < m; i++) L38(B, ldb, i, j) = L38(B, ldb, i, j) + temp * L38(B, ldb, i, k);
            }
        double temp = alpha;
        temp = temp * L38(A, lda, k, k);
        if (temp != 1) for (int i = 0; i < m; i++) L38(B, ldb, i, k) = temp * L38(B, ldb, i, k);
    }
}
/* DPOTRF2('U'), recursive */
static int l38_dpotrf2(int n, double *A, int lda)
{
    if (n == 0) return 0;
    if (n == 1) {
        if (!(A[0] > 0)) return 1;
        A[0] = sqrt(A[0]);
        return 0;
    }
    const int n1 = n / 2, n2 = n - n1;
    int info = l38_dpotrf2(n1, A, lda);
    if (info) return info;
    l38_dtrsm_lutn(n1, n2, 1.0, A, lda, &L38(A, lda, 0, n1), lda);
    l38_dsyrk_ut(n2, n1, -1.0, &L38(A, lda, 0, n1), lda, 1.0, &L38(A, lda, n1, n1), lda);
    info = l38_dpotrf2(n2, &L38(A, lda, n1, n1), lda);
    return info ? info + n1 : 0;
}
/* DPOTRF('U') */
static int l38_dpotrf(int n, double *A, int lda)
{
    const int nb = L38_NB;
    if (nb <= 1 || nb >= n) return l38_dpotrf2(n, A, lda);
    for (int j = 0; j < n; j += nb) {
        const int jb = nb < n - j ? nb : n - j;
        l38_dsyrk_ut(jb, j, -1.0, &L38(A, lda, 0, j), lda, 1.0, &L38(A, lda, j, j), lda);
        const int info = l38_dpotrf2(jb, &L38(A, lda, j, j), lda);
        if (info) return info + j;
        if (j + jb < n) {
            l38_dgemm_tn(jb, n - j - jb, j, -1.0, &L38(A, lda, 0, j), lda, &L38(A, lda, 0, j + jb), lda, 1.0, &L38(A, lda, j, j + jb), lda);
            l38_dtrsm_lutn(jb, n - j - jb, 1.0, &L38(A, lda, j, j), lda, &L38(A, lda, j, j + jb), lda);
        }
    }
    return 0;
}
/* DTRTI2('U','N') */
static void l38_dtrti2(int n, double *A, int lda)
{
    for (int j = 0; j < n; j++) {
        L38(A, lda, j, j) = 1.0 / L38(A, lda, j, j);
        const double ajj = -L38(A, lda, j, j);
        /* DTRMV('U','N','N', j, A, lda, A(:,j), 1) */
        for (int c = 0; c < j; c++) {
            if (L38(A, lda, c, j) != 0) {
                const double temp = L38(A, lda, c, j);
                for (int r = 0; r < c; r++) L38(A, lda, r, j) = L38(A, lda, r, j) + temp * L38(A, lda, r, c);
                L38(A, lda, c, j) = L38(A, lda, c, j) * L38(A, lda, c, c);
            }
        }
        for (int r = 0; r < j; r++) L38(A, lda, r, j) = ajj * L38(A, lda, r, j);      /* DSCAL */
    }
}
/* DTRTRI('U','N') */
static void l38_dtrtri(int n, double *A, int lda)
{
    const int nb = L38_NB;
    if (nb <= 1 || nb >= n) { l38_dtrti2(n, A, lda); return; }
    for (int j = 0; j < n; j += nb) {
        const int jb = nb < n - j ? nb : n - j;
        l38_dtrmm_lunn(j, jb, 1.0, A, lda, &L38(A, lda, 0, j), lda);
        l38_dtrsm_runn(j, jb, -1.0, &L38(A, lda, j, j), lda, &L38(A, lda, 0, j), lda);
        l38_dtrti2(jb, &L38(A, lda, j, j), lda);
    }
}
/* DLAUU2('U') */
static void l38_dlauu2(int n, double *A, int lda)
{
    for (int i = 0; i < n; i++) {
        const double aii = L38(A, lda, i, i);
        if (i < n - 1) {
            double d = 0;                                          /* DDOT over row i, columns i .. n-1 */
            for (int k = i; k < n; k++) d = d + L38(A, lda, i, k) * L38(A, lda, i, k);
            L38(A, lda, i, i) = d;
            /* DGEMV('N', i, n-i-1, 1, A(0,i+1), lda, A(i,i+1), lda, aii, A(0,i), 1): y := aii y first, then the columns in order */
            for (int r = 0; r < i; r++) L38(A, lda, r, i) = aii * L38(A, lda, r, i);
            for (int k = i + 1; k < n; k++) {
                const double temp = L38(A, lda, i, k);
                for (int r = 0; r < i; r++) L38(A, lda, r, i) = L38(A, lda, r, i) + temp * L38(A, lda, r, k);
            }
        } else {
            for (int r = 0; r <= i; r++) L38(A, lda, r, i) = aii * L38(A, lda, r, i);  /* DSCAL */
        }
    }
}
/* DLAUUM('U') */
static void l38_dlauum(int n, double *A, int lda)
{
    const int nb = L38_NB;
    if (nb <= 1 || nb >= n) { l38_dlauu2(n, A, lda); return; }
    for (int i = 0; i < n; i += nb) {
        const int ib = nb < n - i ? nb : n - i;
        l38_dtrmm_rutn(i, ib, 1.0, &L38(A, lda, i, i), lda, &L38(A, lda, 0, i), lda);
        l38_dlauu2(ib, &L38(A, lda, i, i), lda);
        if (i + ib < n) {
            l38_dgemm_nt(i, ib, n - i - ib, 1.0, &L38(A, lda, 0, i + ib), lda, &L38(A, lda, i, i + ib), lda, 1.0, &L38(A, lda, 0, i), lda);
            l38_dsyrk_un(ib, n - i - ib, 1.0, &L38(A, lda, i, i + ib), lda, 1.0, &L38(A, lda, i, i), lda);
        }
    }
}
/* dpotrf + dpotri ('U') + the mirror of MatrixInverseGmNeEN (:1346-1369); returns 1 when a pivot is not positive */
static inline int chol_inverse_upper_lapack38(double *a, int n)
{
    if (l38_dpotrf(n, a, n)) return 1;
    l38_dtrtri(n, a, n);
    l38_dlauum(n, a, n);
    for (int i = 1; i < n; i++)
        for (int j = 0; j < i; j++) a[(size_t)j * n + i] = a[(size_t)i * n + j];
    return 0;
}
#endif
