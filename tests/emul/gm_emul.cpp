// gm_emul.cpp -- TEST INFRASTRUCTURE ONLY.  Compiles the device source of the Gaussian fit
// (pareben_amd/csrc/gm_fit.h) for the CPU with one "thread" per workgroup (PAREBEN_HOST_EMUL) so
// that tests can compare its control flow and results with the oracle without a GPU.  The
// shipped library never links this and has no CPU fallback.
#define PAREBEN_HOST_EMUL 1
#define GM_PHASES_H "../../tests/emul/gm_host.h"      // plain-loop phases instead of pareben_amd/csrc/gm_dev.h (paths relative to gm_fit.h)
#define BM_PHASES_H "../../tests/emul/bm_host.h"      // same for the binomial fit (bm_dev.h)
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <algorithm>
#include "../../pareben_amd/csrc/types.h"
#include "../../pareben_amd/csrc/blk.h"
#include "../../pareben_amd/csrc/gm_fit.h"
#include "../../pareben_amd/csrc/bm_fit.h"
#include "../../pareben_amd/csrc/gm_strict.h"

namespace {
struct Fold {
    int N, nte, n_main;
    std::vector<double> X, y, Xte, yte, scale, rscale, bt0, cs, G;
    double ymean, varY;
};

// CPU stand-ins for split_kernel / colstats_kernel / ystats_kernel / gram_kernel
void prepare(Fold &F, const double *basis, int n, int p, const double *y, const int *fold_id, int f, int n_main = -1)
{
    if (n_main < 0) n_main = p;
    F.n_main = n_main;
    std::vector<int> tr, te;
    for (int i = 0; i < n; i++) (fold_id[i] == f + 1 ? te : tr).push_back(i);
    F.N = (int)tr.size(); F.nte = (int)te.size();
    F.X.resize((size_t)F.N * p); F.Xte.resize((size_t)F.nte * p + 1); F.y.resize(F.N); F.yte.resize(F.nte + 1);
    for (int j = 0; j < p; j++) {
        for (int r = 0; r < F.N; r++) F.X[(size_t)j * F.N + r] = basis[(size_t)j * n + tr[r]];
        for (int r = 0; r < F.nte; r++) F.Xte[(size_t)j * F.nte + r] = basis[(size_t)j * n + te[r]];
    }
    for (int r = 0; r < F.N; r++) F.y[r] = y[tr[r]];
    for (int r = 0; r < F.nte; r++) F.yte[r] = y[te[r]];
    F.scale.resize(p); F.rscale.resize(p); F.bt0.resize(p); F.cs.resize(p); F.G.resize((size_t)p * p);
    for (int j = 0; j < p; j++) {
        double q = 0, xy = 0, x1 = 0;
        for (int h = 0; h < F.N; h++) { double v = F.X[(size_t)j * F.N + h]; q += v * v; xy += v * F.y[h]; x1 += v; }
        if (q == 0) q = 1;
        double s = std::sqrt(q);
        F.scale[j] = s; F.rscale[j] = 1 / s; F.bt0[j] = xy / s; F.cs[j] = x1 / s;
    }
    double s = 0; for (int h = 0; h < F.N; h++) s += F.y[h];
    F.ymean = s / F.N;
    double v = 0; for (int h = 0; h < F.N; h++) { double d = F.y[h] - F.ymean; v += d * d; }
    F.varY = v / (F.N - 1);
    for (int u = 0; u < p; u++)
        for (int i = 0; i < p; i++) {
            double a = 0;
            if (u < n_main) for (int h = 0; h < F.N; h++) a += F.X[(size_t)i * F.N + h] * (F.X[(size_t)u * F.N + h] * F.rscale[u]);
            else for (int h = 0; h < F.N; h++) a += F.X[(size_t)i * F.N + h] * (F.X[(size_t)u * F.N + h] / F.scale[u]);
            F.G[(size_t)u * p + i] = a / F.scale[i];
        }
}

FoldDev dev_view(const Fold &F)
{
    FoldDev D;
    D.X = F.X.data(); D.y = F.y.data(); D.Xte = F.Xte.data(); D.yte = F.yte.data();
    D.scale = F.scale.data(); D.rscale = F.rscale.data(); D.bt0 = F.bt0.data(); D.cs = F.cs.data();
    D.Xt = nullptr;
    D.G = F.G.data(); D.ymean = F.ymean; D.varY = F.varY; D.N = F.N; D.nte = F.nte; D.n_main = F.n_main;
    D.slot_of = nullptr; D.pool_next = nullptr; D.pool_base = 0; D.pool_rows = 0; D.lazy = 0;
    return D;
}

struct Work {
    std::vector<double> kd, sig, md; std::vector<int> ki, used, rowid, pfree; std::vector<signed char> act;
    GmWork W;
    Work(int K, int cap)
    {
        kd.assign((size_t)7 * K, 0); ki.assign((size_t)2 * K, 0); act.assign(K, 0);
        sig.assign((size_t)2 * cap * cap, 0); md.assign((size_t)7 * (cap + 1), 0); used.assign(cap + 1, 0);
        double *d = kd.data();
        W.Sin = d; d += K; W.Qin = d; d += K; W.Sout = d; d += K; W.Qout = d; d += K; W.dml = d; d += K; W.aroot = d; d += K; W.bt = d;
        W.upos = ki.data(); W.todo = ki.data() + K; W.act = act.data();
        W.Sig = sig.data(); W.H = sig.data() + (size_t)cap * cap;
        d = md.data(); int c1 = cap + 1;
        W.A = d; d += c1; W.mu = d; d += c1; W.gam = d; d += c1; W.v1 = d; d += c1; W.v2 = d; d += c1; W.v3 = d; d += c1; W.v4 = d;
        W.used = used.data(); rowid.assign(cap + 1, 0); W.rowid = rowid.data(); pfree.assign(cap + 2, 0); W.pfree = pfree.data(); W.priv_base = 0; W.priv_rows = 0; W.vb = nullptr; W.bsc = nullptr; W.e = nullptr; W.cap = cap; W.ld = cap; W.cap_flag = cap;
    }
};
}  // namespace

extern "C" int emul_default_cap(int K)
{
    long cap = (long)(1e7 / K);
    if (cap > K) cap = K;
    if (cap > 2048) cap = 2048;
    if (cap < 2) cap = 2;
    return (int)cap;
}

// same contract as pareben_cv_grid (gaussian; epis = 1 runs the epistasis rule set on the expanded design)
static int emul_gauss_grid(const double *basis_in, int n, int p_in, const double *y, const int *fold_id, int n_folds,
                           const double *alpha, const double *lambda, int n_cells, int epis,
                           double *fold_err, int *status, long long *counters)
{
    std::vector<double> Z;
    const double *basis = basis_in;
    int p = p_in;
    GmVariant variant{0, 0.9, 0.001, 1e-3, 1e2, 1e-10};
    int cap = emul_default_cap(p);
    if (epis) {                                    // expanded design, reference column order
        p = p_in * (p_in + 1) / 2;
        Z.resize((size_t)n * p);
        std::memcpy(Z.data(), basis_in, sizeof(double) * (size_t)n * p_in);
        size_t kk = p_in;
        for (int i = 0; i < p_in - 1; i++)
            for (int j = i + 1; j < p_in; j++, kk++)
                for (int h = 0; h < n; h++) Z[kk * n + h] = basis_in[(size_t)i * n + h] * basis_in[(size_t)j * n + h];
        basis = Z.data();
        variant = GmVariant{1, 0.99, 0.01, 0.1, 1e3, 0.0};
        cap = 0;
        for (int f = 0; f < n_folds; f++) {
            int N = 0; for (int i = 0; i < n; i++) if (fold_id[i] != f + 1) N++;
            cap = std::max(cap, N > p_in ? 2 * p_in : (N < 200 ? 4 * p_in : p_in));
        }
        cap = std::min(std::min(cap, p), 2048);
    }
    std::vector<Fold> folds(n_folds);
    for (int f = 0; f < n_folds; f++) prepare(folds[f], basis, n, p, y, fold_id, f, p_in);
    Work ws(p, cap);
    int ired[4];
    Blk B; B.tid = 0; B.nthr = 1; B.lane = 0; B.wave = 0; B.nwave = 1; B.red = nullptr; B.ired = ired; B.pool = nullptr; B.pool_n = 0;
    // PAREBEN_EMUL_LAZY=<rows per fold>[,<private rows>]: exercise the on-demand Gram-row pool (and the
    // per-workgroup private rows behind it) instead of the full matrix
    const char *lz = getenv("PAREBEN_EMUL_LAZY");
    int pool_rows = 0, priv_rows = cap;
    if (lz) { pool_rows = atoi(lz); if (const char *cm = strchr(lz, ',')) priv_rows = atoi(cm + 1); }
    std::vector<double> rows;
    std::vector<std::vector<int>> slot(n_folds);
    std::vector<int> next(n_folds, 0);
    if (pool_rows > 0) {
        rows.assign(((size_t)pool_rows * n_folds + priv_rows) * p, NAN);
        for (int f = 0; f < n_folds; f++) slot[f].assign(p, -1);
        ws.W.priv_base = pool_rows * n_folds; ws.W.priv_rows = priv_rows;
    }
    for (int c = 0; c < n_cells; c++)
        for (int f = 0; f < n_folds; f++) {
            FoldDev F = dev_view(folds[f]);
            if (pool_rows > 0) { F.G = rows.data(); F.slot_of = slot[f].data(); F.pool_next = &next[f]; F.pool_base = f * pool_rows; F.pool_rows = pool_rows; F.lazy = 1; }
            GmScalars S; FitCounters cnt; S.c = &cnt; S.ph = nullptr; S.v = variant;
            gm_fit(B, F, ws.W, p, lambda[c], alpha[c], S);
            const int u = c * n_folds + f;
            fold_err[u] = gm_fold_sse(B, F, ws.W, S);
            if (getenv("EMUL_DBG") && pool_rows > 0) {
                int nbad = 0, nrow = 0;
                for (int uu = 0; uu < p; uu++) { int sl = slot[f][uu]; if (sl < 0) continue; nrow++;
                    for (int i = 0; i < p; i++) if (rows[(size_t)sl * p + i] != folds[f].G[(size_t)uu * p + i]) nbad++; }
                fprintf(stderr, "c=%d f=%d rows=%d mismatching entries=%d M=%d\n", c, f, nrow, nbad, S.M);
            }
            if (status) status[u] = S.status;
            if (counters) std::memcpy(counters + (size_t)u * PAREBEN_NCOUNTERS, &cnt, sizeof cnt);
        }
    return 0;
}

extern "C" int emul_gm_cv_grid(const double *basis, int n, int p, const double *y, const int *fold_id, int n_folds,
                               const double *alpha, const double *lambda, int n_cells,
                               double *fold_err, int *status, long long *counters)
{ return emul_gauss_grid(basis, n, p, y, fold_id, n_folds, alpha, lambda, n_cells, 0, fold_err, status, counters); }

extern "C" int emul_gf_cv_grid(const double *basis, int n, int p, const double *y, const int *fold_id, int n_folds,
                               const double *alpha, const double *lambda, int n_cells,
                               double *fold_err, int *status, long long *counters)
{ return emul_gauss_grid(basis, n, p, y, fold_id, n_folds, alpha, lambda, n_cells, 1, fold_err, status, counters); }

// decision trace of the following emul_gm_fit calls (same layout as pareben_set_trace / eben_set_trace)
static unsigned long long *g_trace = nullptr;
static long long g_trace_cap = 0;
extern "C" void emul_set_trace(unsigned long long *buf, long long max_records) { g_trace = buf; g_trace_cap = max_records; if (buf) buf[0] = 0; }

// one fit on all rows; out = {intercept, beta(noise precision), M}; used/mu sized >= cap
extern "C" int emul_gm_fit(const double *X, const double *y, int n, int p, double lambda, double alpha,
                           double *out, int *used, double *mu, double *sigdiag, long long *counters)
{
    std::vector<int> fid(n, 2);
    Fold F; prepare(F, X, n, p, y, fid.data(), 0);
    const int cap = emul_default_cap(p);
    Work ws(p, cap);
    Blk B; B.tid = 0; B.nthr = 1; B.lane = 0; B.wave = 0; B.nwave = 1; B.red = nullptr; B.ired = nullptr; B.pool = nullptr; B.pool_n = 0;
    FoldDev D = dev_view(F);
    GmScalars S; FitCounters cnt; S.c = &cnt; S.ph = nullptr; S.v = GmVariant{0, 0.9, 0.001, 1e-3, 1e2, 1e-10};
    S.trace = g_trace; S.trace_cap = g_trace_cap;
    gm_fit(B, D, ws.W, p, lambda, alpha, S);
    out[0] = S.b; out[1] = S.beta; out[2] = S.M;
    for (int i = 0; i < S.M; i++) { used[i] = ws.W.used[i]; mu[i] = ws.W.mu[i] / F.scale[ws.W.used[i]]; sigdiag[i] = ws.W.Sig[(size_t)i * cap + i] / (F.scale[ws.W.used[i]] * F.scale[ws.W.used[i]]); }
    if (counters) std::memcpy(counters, &cnt, sizeof cnt);
    return S.status;
}


// the strict-order fit (gm_strict.h: the reference's own formulation and operation order) on all rows, with the decision
// trace when one is set; out = {intercept, beta, M, fold-style SSE on the training rows is not computed}
extern "C" int emul_gm_fit_strict(const double *X, const double *y, int n, int p, double lambda, double alpha,
                                  double *out, int *used, double *mu, long long *counters)
{
    std::vector<int> fid(n, 2);
    Fold F; prepare(F, X, n, p, y, fid.data(), 0);
    for (int j = 0; j < p; j++) {                     // sequential-order norms (strict_scale_kernel on the device)
        double q = 0;
        for (int h = 0; h < n; h++) q = q + X[(size_t)j * n + h] * X[(size_t)j * n + h];
        if (q == 0) q = 1;
        F.scale[j] = std::sqrt(q); F.rscale[j] = 1 / F.scale[j];
    }
    const int cap = emul_default_cap(p);
    Work ws(p, cap);
    const int nv = std::max(n, cap + 2);
    std::vector<double> ext((size_t)3 * nv + (size_t)cap * p + 2 * (size_t)cap * cap + 2 * (size_t)(cap + 2));
    GsExtra E;
    double *d = ext.data();
    E.t = d; d += nv; E.e = d; d += nv; E.phi = d; d += nv; E.BP = d; d += (size_t)cap * p; E.SigNew = d; d += (size_t)cap * cap;
    E.D = d; d += (size_t)cap * cap; E.w1 = d; d += cap + 2; E.w2 = d;
    int ired[4]; double red[4];
    // PAREBEN_EMUL_STRICT_POOL=1: an LDS-sized pool and the sample-major design copy, so that the staged code paths of
    // gm_strict.h (what the device runs) are the ones stepped here; default: the plain fallbacks
    std::vector<double> pool, xt;
    Blk B; B.tid = 0; B.nthr = 1; B.lane = 0; B.wave = 0; B.nwave = 1; B.red = red; B.ired = ired; B.pool = nullptr; B.pool_n = 0;
    FoldDev D = dev_view(F);
    if (getenv("PAREBEN_EMUL_STRICT_POOL")) {
        pool.assign(19456, 0.0); B.pool = pool.data(); B.pool_n = 19456;
        xt.resize((size_t)n * p);
        for (int i = 0; i < p; i++) for (int h = 0; h < n; h++) xt[(size_t)h * p + i] = X[(size_t)i * n + h];
        D.Xt = xt.data();
    }
    GmScalars S; FitCounters cnt; S.c = &cnt; S.ph = nullptr; S.v = GmVariant{0, 0.9, 0.001, 1e-3, 1e2, 1e-10};
    S.trace = g_trace; S.trace_cap = g_trace_cap;
    gs_fit(B, D, ws.W, E, p, lambda, alpha, S);
    out[0] = S.b; out[1] = S.beta; out[2] = S.M;
    for (int i = 0; i < S.M; i++) { used[i] = ws.W.used[i]; mu[i] = ws.W.mu[i] / F.scale[ws.W.used[i]]; }
    if (counters) std::memcpy(counters, &cnt, sizeof cnt);
    return S.status;
}

// same contract as pareben_cv_grid (binomial; epis = 1: the NeFull.c rule set on the expanded design): fold_err = mean
// held-out log-likelihood
static int bm_cv_impl(const double *basis_in, int n, int p_in, const double *y, const int *fold_id, int n_folds,
                      const double *alpha, const double *lambda, int n_cells, int epis,
                      double *fold_err, int *status, long long *counters)
{
    std::vector<double> Z;
    const double *basis = basis_in;
    int p = p_in;
    if (epis) {                                    // expanded design, reference column order (NeFull.c:90-105)
        p = p_in * (p_in + 1) / 2;
        Z.resize((size_t)n * p);
        std::memcpy(Z.data(), basis_in, sizeof(double) * (size_t)n * p_in);
        size_t kk = p_in;
        for (int i = 0; i < p_in - 1; i++)
            for (int j = i + 1; j < p_in; j++, kk++)
                for (int h = 0; h < n; h++) Z[kk * n + h] = basis_in[(size_t)i * n + h] * basis_in[(size_t)j * n + h];
        basis = Z.data();
    }
    std::vector<Fold> folds(n_folds);
    int nmax = 1;
    for (int f = 0; f < n_folds; f++) { prepare(folds[f], basis, n, p, y, fold_id, f); nmax = std::max(nmax, std::max(folds[f].N, folds[f].nte)); }
    const int bmax = epis ? 2 * p_in : p;
    int cap = std::min(p, bmax) + 1; if (cap > 1024) cap = 1024;
    const int ld = cap + 1;
    std::vector<double> kd((size_t)7 * p), sig((size_t)2 * ld * ld), md((size_t)9 * ld), nd((size_t)5 * nmax), bp((size_t)p * ld);
    std::vector<int> ki((size_t)2 * p), used(ld); std::vector<signed char> act(p);
    BmWork W;
    double *d = kd.data();
    W.Sin = d; d += p; W.Qin = d; d += p; W.Sout = d; d += p; W.Qout = d; d += p; W.dml = d; d += p; W.aroot = d; d += p; W.bb = d;
    W.upos = ki.data(); W.todo = ki.data() + p; W.act = act.data();
    W.Sig = sig.data(); W.H = sig.data() + (size_t)ld * ld;
    d = md.data();
    W.A = d; d += ld; W.mu = d; d += ld; W.g = d; d += ld; W.dmu = d; d += ld; W.mnew = d; d += ld; W.tmp = d; d += ld; W.tp = d; d += ld; W.v3 = d; d += ld; W.v4 = d;
    W.used = used.data();
    d = nd.data();
    W.w = d; d += nmax; W.pm = d; d += nmax; W.yv = d; d += nmax; W.e = d; d += nmax; W.bphi = d;
    W.BP = bp.data(); W.cap = cap; W.ld = ld; W.phi_div = epis; W.bmax = bmax;
    Blk B; B.tid = 0; B.nthr = 1; B.lane = 0; B.wave = 0; B.nwave = 1; B.red = nullptr; B.ired = nullptr; B.pool = nullptr; B.pool_n = 0;
    for (int c = 0; c < n_cells; c++)
        for (int f = 0; f < n_folds; f++) {
            FoldDev F = dev_view(folds[f]);
            GmScalars S; FitCounters cnt; S.c = &cnt; S.ph = nullptr; S.v.epis = epis;
            double ll;
            bm_fit(B, F, W, p, lambda[c], alpha[c], S, &ll);
            const int u = c * n_folds + f;
            fold_err[u] = bm_fold_loglik(B, F, W, S);
            if (status) status[u] = S.status;
            if (counters) std::memcpy(counters + (size_t)u * PAREBEN_NCOUNTERS, &cnt, sizeof cnt);
        }
    return 0;
}

extern "C" int emul_bm_cv_grid(const double *basis, int n, int p, const double *y, const int *fold_id, int n_folds,
                               const double *alpha, const double *lambda, int n_cells,
                               double *fold_err, int *status, long long *counters)
{
    return bm_cv_impl(basis, n, p, y, fold_id, n_folds, alpha, lambda, n_cells, 0, fold_err, status, counters);
}

extern "C" int emul_bf_cv_grid(const double *basis, int n, int p, const double *y, const int *fold_id, int n_folds,
                               const double *alpha, const double *lambda, int n_cells,
                               double *fold_err, int *status, long long *counters)
{
    return bm_cv_impl(basis, n, p, y, fold_id, n_folds, alpha, lambda, n_cells, 1, fold_err, status, counters);
}
