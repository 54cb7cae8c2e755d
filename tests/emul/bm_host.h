// bm_host.h -- TEST INFRASTRUCTURE ONLY.  Plain-loop implementations of the phases of a binomial fit, with the signatures
// of pareben_amd/csrc/bm_dev.h, for the CPU harness (tests/emul/gm_emul.cpp).  Never part of the shipped library.
#pragma once
#define PHX2_BEGIN(v) do {} while (0)
#define PHX2_END(v, k) do {} while (0)

DEV void bm_phi_mu(const Blk &B, const FoldDev &F, const BmWork &W, int M, const double *mu, double *out) { bm_phi_mu_plain(B, F, W, M, mu, out); }

// t_i = sum_j BP[i][j] * vec[j] (j ascending) for every feature i, handed to f(i, t)
template <class Fn>
DEV void bm_rows_dot(const Blk &B, const BmWork &W, int K, int M, const double *vec, Fn f)
{
    PAR(i, K) {
        const double *bp = W.BP + (size_t)i * W.ld;
        double t = 0;
        for (int j = 0; j < M; j++) t += bp[j] * vec[j];
        f(i, t);
    }
}

// BP[i][p] = sum_h x_i[h] w[h] Phi_p[h] / |x_i| for all features i and model columns p < M;
// also bb-style single columns through `only` (>= 0: only that column, written to W.bb).

// want_stats (device build): also bb_i = x_i' diag(w) x_i -> W.bb[i] and ze_i = x_i' e -> W.aroot[i] (what the
// full-stat pass needs per feature, NEmainEff.c:1745-1757), taken from the same pass over the design columns.
// Returns 1 when it did (matrix-core path), 0 when the caller has to compute them.
DEVNI int bm_weighted_rows(const Blk &B, const FoldDev &F, const BmWork &W, int K, int M, bool want_stats = false, long long *phx = nullptr)
{
    (void)phx;
    const int N = F.N, ld = W.ld;
    (void)want_stats;
    for (int i = 0; i < K; i++)
        for (int p = 0; p < M; p++) {
            double a = 0;
            for (int h = 0; h < N; h++) a += (F.X[(size_t)i * N + h] * W.w[h]) * BM_PHI(p, h);
            W.BP[(size_t)i * ld + p] = a / F.scale[i];
        }
    blk_sync(B);
    return 0;
}


// S_in = bb_i / |x_i|^2 - BP_i' Sigma BP_i and Q_in = ze_i / |x_i| (NEmainEff.c:1732-1762)
DEV void bm_quad_features(const Blk &B, const FoldDev &F, const BmWork &W, int K, int M)
{
    const int ld = W.ld;
    PAR(i, K) {
        const double *bp = W.BP + (size_t)i * ld;
        double quad = 0;
        for (int p = 0; p < M; p++) {
            double t = 0;
            for (int q = 0; q < M; q++) t += W.Sig[(size_t)p * ld + q] * bp[q];
            quad += t * bp[p];
        }
        const double sc = F.scale[i];
        W.Sin[i] = W.bb[i] / (sc * sc) - quad;
        W.Qin[i] = W.aroot[i] / sc;
    }
    blk_sync(B);
}

DEV void bm_grad_hessian(const Blk &B, const FoldDev &F, const BmWork &W, int M, int N)
{
    const int ld = W.ld;
    (void)B;
        for (int j = 1; j < M; j++) {
            double ga = 0, ha = 0;
            for (int h = 0; h < N; h++) { const double ph = BM_PHI(j, h); ga += W.e[h] * ph; ha += W.w[h] * ph; }
            W.g[j] = ga - W.A[j - 1] * W.mu[j];
            W.H[j] = ha; W.H[(size_t)j * ld] = ha;
        }
        for (int j = 1; j < M; j++)
            for (int k = 1; k <= j; k++) {
                double a = 0;
                for (int h = 0; h < N; h++) a += BM_PHI(j, h) * W.w[h] * BM_PHI(k, h);
                if (j == k) a += W.A[k - 1];
                W.H[(size_t)k * ld + j] = a; W.H[(size_t)j * ld + k] = a;
            }
}

DEV void bm_feature_stats(const Blk &B, const FoldDev &F, const BmWork &W, int K, int N, int have_stats)
{
    (void)B; (void)have_stats;
    for (int i = 0; i < K; i++) {
        double bbq = 0, ze = 0;
        for (int h = 0; h < N; h++) { const double x = F.X[(size_t)i * N + h]; bbq += W.w[h] * (x * x); ze += x * W.e[h]; }
        W.bb[i] = bbq; W.aroot[i] = ze;        // scratch: aroot is rewritten by every dML pass
    }
}

DEV void bm_add_products(const Blk &B, const FoldDev &F, const BmWork &W, int K, int M, int N)
{
    (void)B;
    for (int i = 0; i < K; i++) { double a = 0; for (int h = 0; h < N; h++) a += F.X[(size_t)i * N + h] * W.bphi[h]; W.bb[i] = a / F.scale[i]; }
    for (int p = 0; p < M; p++) { double a = 0; for (int h = 0; h < N; h++) a += BM_PHI(p, h) * W.bphi[h]; W.tmp[p] = a; }
}
