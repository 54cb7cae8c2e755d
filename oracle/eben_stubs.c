/* placeholders until the Gf / Bm restatements land (TEST INFRASTRUCTURE ONLY) */
#include "eben_oracle.h"
#ifndef HAVE_GF
int eben_gf_fit(const double *X, const double *y, int N, int K, double lambda, double alpha,
                double *Beta, double *wald, double *intercept, double *residual, eben_counters *cnt)
{ (void)X;(void)y;(void)N;(void)K;(void)lambda;(void)alpha;(void)Beta;(void)wald;(void)intercept;(void)residual;(void)cnt; return -1; }
#endif
#ifndef HAVE_BM
int eben_bm_fit(const double *X, const double *y, int N, int K, double lambda, double alpha,
                double *loglik, double *Beta, double *wald, double *intercept, eben_counters *cnt)
{ (void)X;(void)y;(void)N;(void)K;(void)lambda;(void)alpha;(void)loglik;(void)Beta;(void)wald;(void)intercept;(void)cnt; return -1; }
#endif
