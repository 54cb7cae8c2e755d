// bm_fit.h -- one binomial (logistic) EBEN fit executed by ONE workgroup: main effects
// (ElasticNetBinaryNEmainEff.c, "Bm") or main effects + pairwise epistasis (ElasticNetBinaryNeFull.c, "Bf").
//
// Bf runs the same code on the expanded design (expand_kernel: the K main columns followed by the pair
// columns x_i*x_j in the reference's order) with the rule set in which NeFull.c differs from NEmainEff.c,
// each cited where `S.v.epis` / `W.phi_div` switches it: block cut-off 0.99 (:254), phi = column / scale by
// division (:437-450, :755), Newton step with y clamped to [1e-5, 1-1e-5] and weights < 1e-5 -> 1e-3,
// > 1e5 -> 1e3 (:1042-1048) that stops when ALL M gradient entries are small (:1085-1099), Sigma downdate of a
// delete associated as S - (s_i/s_jj) s_j (:1606), delete-priority only above 100 bases (:1805-1809), outer
// stopping sum over the M-1 precisions (:133-134), capacity bMax = 2K bases from the R wrapper
// (EBelasticNet.Binomial.R:7-9; NeFull.c:549-553).  The reference regenerates a pair column inside every
// sweep and associates its products as ((x_i*phi)*w)*x_j etc.; on the materialised column the same products
// round differently in the last bit for non-integer designs (exactly equal for genotype codes -1/0/1).
//
// What it computes is what EBEN_orig/src/ElasticNetBinaryNEmainEff.c computes for one
// (training fold, alpha, lambda): the outer loop (:329-344) around the inner
// add / re-estimate / delete ascent (:397-827) with a Laplace (IRLS) posterior mode
// (fEBCatPostModeBmNeEN :1808-2010) after every block of actions.
//
// The per-sample IRLS weights w = y(1-y) change at every mode update, so unlike the Gaussian
// kernel nothing can be cached as a Gram matrix: every action needs the weighted cross products
// BP[i][p] = x_i' diag(w) Phi_p / |x_i| of all K features with the M model columns.  They are
// computed once per (mode update / action) by the whole workgroup into a K x M scratch matrix in
// HBM -- a small GEMM on the FP64 matrix cores (bm_weighted_rows) -- and then reused by the
// per-feature quadratic forms (bm_quad_features, matrix cores too); the Newton step's Hessian
// Phi' diag(w) Phi is the third such product (bm_postmode).  Phi is never materialised: column 0
// is the intercept, column l+1 is the design column of used[l] times 1/|x|.
//
// Reference quirks kept: Q1 first basis = column 0, force-deleted once;
// Q12 the outer stopping sum covers M = N_used+1 precisions (one stale slot); Q15 add-priority
// never fires; the re-estimate S/Q update reads the already updated Sigma row; the Newton loop
// keeps the y of a rejected line search.
#pragma once
#include "blk.h"
#include "types.h"
#include "gm_fit.h"      // action codes, GmScalars, counters, gm_spd_inverse

struct BmWork {
    double *Sin, *Qin, *Sout, *Qout, *dml, *aroot, *bb;   // K each
    int *upos, *todo;                                      // K each
    signed char *act;                                      // K
    double *Sig, *H;                                       // ld x ld, column-major
    double *A, *mu, *g, *dmu, *mnew, *tmp, *tp, *v3, *v4;  // ld each
    int *used;                                             // ld
    double *w, *pm, *yv, *e, *bphi;                        // Nmax each
    double *BP;                                            // K x ld (row i = feature i)
    int cap, ld;                                           // cap = max model size M, ld = cap
    int phi_div;                                           // Bf: model column = x / |x| (division); Bm: x * (1/|x|)
    int bmax;                                              // Bf: most bases a model may hold (the R wrapper's bMax = 2K); Bm: unused (cap decides)
};

// design column u, normalised, at sample h: NEmainEff.c:644-646 multiplies by the reciprocal norm, NeFull.c:437-450 divides
DEV double bm_col(const FoldDev &F, const BmWork &W, int N, int u, int h)
{
    const double x = F.X[(size_t)u * N + h];
    return W.phi_div ? x / F.scale[u] : x * F.rscale[u];
}
// model column p at sample h
#define BM_PHI(p, h) ((p) == 0 ? 1.0 : bm_col(F, W, N, W.used[(p) - 1], (h)))

// a GmWork view so the Gaussian kernel's SPD inverse can be reused on (Sig, ld)
DEV GmWork bm_as_gm(const BmWork &W)
{
    GmWork G{};
    G.Sig = W.Sig; G.H = W.H; G.v3 = W.v3; G.v4 = W.v4; G.cap = W.cap; G.ld = W.ld;
    return G;
}

// pm[h] = sum_p Phi_p[h] * mu[p]
DEVNI void bm_phi_mu(const Blk &B, const FoldDev &F, const BmWork &W, int M, const double *mu, double *out)
{
    const int N = F.N;
#ifndef PAREBEN_HOST_EMUL
    if (3 * M + 8 <= B.pool_n) {
        // column ids, norms and coefficients staged in LDS first: the loop then has one coalesced design-column load per
        // term and nothing behind a used[] -> rscale[] address chain; same expression, same order
        double *lm = B.pool, *ls = B.pool + M;
        int *lu = (int *)(B.pool + 2 * M);
        blk_sync(B);
        PAR(p, M) { lm[p] = mu[p]; if (p >= 1) { const int u = W.used[p - 1]; lu[p] = u; ls[p] = W.phi_div ? F.scale[u] : F.rscale[u]; } }
        blk_sync(B);
        const bool dv = W.phi_div != 0;
        PAR(h, N) {
            double a = 0;
            a += 1.0 * lm[0];
#pragma unroll 4
            for (int p = 1; p < M; p++) {
                const double x = F.X[(size_t)lu[p] * N + h];
                a += (dv ? x / ls[p] : x * ls[p]) * lm[p];
            }
            out[h] = a;
        }
        blk_sync(B);
        return;
    }
#endif
    PAR(h, N) {
        double a = 0;
        for (int p = 0; p < M; p++) a += BM_PHI(p, h) * mu[p];
        out[h] = a;
    }
    blk_sync(B);
}

// y = sigmoid(pm); returns -sum t log y + (1-t) log(1-y)  (:2013-2032)
DEVNI double bm_data_error(const Blk &B, const FoldDev &F, const BmWork &W)
{
    const int N = F.N;
    double part = 0;
    PAR(h, N) {
        const double y = 1 / (1 + exp(-W.pm[h]));
        W.yv[h] = y;
        const double t = F.y[h];
        if (y != 0) part -= t * log(y);
        if (y != 1) part -= (1 - t) * log(1 - y);
    }
    const double s = blk_sum(B, part);
    return s;
}

// BP[i][p] = sum_h x_i[h] w[h] Phi_p[h] / |x_i| for all features i and model columns p < M;
// also bb-style single columns through `only` (>= 0: only that column, written to W.bb).
#ifndef PAREBEN_HOST_EMUL
// One 16-feature tile of bm_weighted_rows on the matrix cores, NCT column tiles of the staged block (compile-time: no
// guards around the matrix ops), EXT = the staged block carries the residual column (statistics wanted).
template <int NCT, int EXT>
DEV void bm_wr_tile(gptr_cd xa, lptr_d zb, lptr_d lw, int pitch, int Nu, int Nr, int l4, double (&out)[NCT][4], double &bbq_out)
{
    typedef double bd4 __attribute__((ext_vector_type(4)));
    constexpr int RS = 8;
    bd4 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) acc[ct] = bd4{0, 0, 0, 0};
    double bbq = 0;
    double an[RS];
#pragma unroll
    for (int u = 0; u < RS; u++) { const int h = 4 * u + l4; an[u] = xa[h < Nu ? h : Nu - 1]; }
    double bn[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) bn[ct] = zb[ct * 16 * pitch];
    for (int h0 = 0; h0 < Nr; h0 += 4 * RS) {
        double ac[RS];
#pragma unroll
        for (int u = 0; u < RS; u++) ac[u] = (h0 + 4 * u + l4 < Nu) ? an[u] : 0.0;
#pragma unroll
        for (int u = 0; u < RS; u++) { const int h = h0 + 4 * RS + 4 * u + l4; an[u] = xa[h < Nu ? h : Nu - 1]; }
#pragma unroll
        for (int u = 0; u < RS; u++) {
            const int hs = h0 + 4 * u;                             // wave-uniform
            if (hs < Nr) {
                double bc[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ct++) bc[ct] = bn[ct];
                const int hn = hs + 4 < Nr ? hs + 4 : hs;           // next step's B operands behind this step's matrix ops
#pragma unroll
                for (int ct = 0; ct < NCT; ct++) bn[ct] = zb[ct * 16 * pitch + hn];
                const double a = ac[u];
#pragma unroll
                for (int ct = 0; ct < NCT; ct++) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bc[ct], acc[ct], 0, 0, 0);
                if (EXT) bbq += lw[hs + l4] * (a * a);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) out[ct][r] = acc[ct][r];
    bbq_out = bbq;
}
#endif

// want_stats (device build): also bb_i = x_i' diag(w) x_i -> W.bb[i] and ze_i = x_i' e -> W.aroot[i] (what the
// full-stat pass needs per feature, NEmainEff.c:1745-1757), taken from the same pass over the design columns.
// Returns 1 when it did (matrix-core path), 0 when the caller has to compute them.
DEVNI int bm_weighted_rows(const Blk &B, const FoldDev &F, const BmWork &W, int K, int M, bool want_stats = false, long long *phx = nullptr)
{
    (void)phx;
    const int N = F.N, ld = W.ld;
    (void)want_stats;
#ifdef PAREBEN_HOST_EMUL
    for (int i = 0; i < K; i++)
        for (int p = 0; p < M; p++) {
            double a = 0;
            for (int h = 0; h < N; h++) a += (F.X[(size_t)i * N + h] * W.w[h]) * BM_PHI(p, h);
            W.BP[(size_t)i * ld + p] = a / F.scale[i];
        }
#else
    // A small GEMM, X' (diag(w) Phi), on the FP64 matrix cores when the weighted model columns fit in LDS 16 at a
    // time: Z_p = w .* Phi_p staged sample-contiguous with an odd pitch (zero-padded to whole 4-sample groups and
    // whole 16-column tiles); a wave owns 16 features and all column tiles, walks the samples four at a time --
    // A operand: lane l holds x[feature l & 15][sample h0 + (l >> 4)] straight from memory (the four lanes of a
    // feature read consecutive samples; the line stays in L1 for the next three steps; eight steps' loads are issued
    // together, a round ahead of the matrix ops that use them), B operand from LDS one step ahead -- and each
    // accumulator tile is one fma chain over the samples in ascending order.  D register r of lane l is
    // BP[feature (l >> 4) + 4 r][column l & 15]: rows of BP leave as 128-byte segments.
    // want_stats: one more staged column, e: x_i'e falls out of the same products (column pn); x_i' diag(w) x_i is summed
    // on the vector ALU from the operand already in registers and the weights staged behind the columns.
    // Everything the loop branches on is put in SGPRs (arguments of a non-inlined function arrive in VGPRs: a guard on
    // them becomes an exec mask around every matrix op) and LDS / global pointers carry their address space (a generic
    // pointer makes every operand read a flat load followed by s_waitcnt vmcnt(0), which drains the prefetch).
    {
        typedef double bd4 __attribute__((ext_vector_type(4)));
        constexpr int MAXCT = 4;
        const int Nu = uni(N), Ku = uni(K), Mu = uni(M), ldu = uni(ld), ws = uni(want_stats ? 1 : 0);
        const int Nr = (Nu + 3) & ~3, pitch = Nr + 1;
        int pcm = ((uni(B.pool_n) - (ws ? Nr : 0)) / pitch) & ~15;
        if (pcm > 16 * MAXCT) pcm = 16 * MAXCT;
        if (pcm >= 16) {
            const lptr_d Z = as_lds(uni_ptr(B.pool));          // [column][pitch]
            const gptr_cd gX = as_global(uni_ptr(F.X)), gw = as_global(uni_ptr(W.w)), ge = as_global(uni_ptr(W.e));
            const gptr_cd gsc = as_global(uni_ptr(F.scale));
            const gptr_d gBP = as_global_rw(uni_ptr(W.BP)), gbb = as_global_rw(uni_ptr(W.bb)), gze = as_global_rw(uni_ptr(W.aroot));
            const int lane = B.lane, wave = uni(B.wave), nwave = uni(B.nwave), tid = B.tid, nthr = uni(B.nthr);
            const int l15 = lane & 15, l4 = lane >> 4;
            for (int p0 = 0; p0 < Mu; p0 += 0) {
                const int ext = (ws && p0 == 0) ? 1 : 0;
                const int pn = Mu - p0 < pcm - ext ? Mu - p0 : pcm - ext;
                const int pn16 = (pn + ext + 15) & ~15, nct = pn16 >> 4;
                const lptr_d lw = Z + pcm * pitch;                     // the weights, zero beyond the last sample
                blk_sync(B);
                PHX_BEGIN(t_st);
                for (int e = tid; e < pn16 * pitch; e += nthr) {
                    const int pc = e / pitch, h = e - pc * pitch;
                    double v = 0.0;
                    if (h < Nu) {
                        if (pc < pn) v = gw[h] * BM_PHI(p0 + pc, h);
                        else if (ext && pc == pn) v = ge[h];
                    }
                    Z[e] = v;
                }
                if (ext) for (int h = tid; h < Nr; h += nthr) lw[h] = h < Nu ? gw[h] : 0.0;
                blk_sync(B);
                PHX_END(t_st, PH_HBUILD);
                PHX_BEGIN(t_mm);
                for (int ft = wave; ft * 16 < Ku; ft += nwave) {
                    const int il = ft * 16 + l15;
                    const gptr_cd xa = gX + (size_t)(il < Ku ? il : Ku - 1) * Nu;
                    const lptr_d zb = Z + l15 * pitch + l4;
                    double acc[MAXCT][4];
                    double bbq = 0;
#define BM_WR_CASE(n)                                                                                                   \
                    case n: {                                                                                           \
                        double o[n][4];                                                                                 \
                        if (ext) bm_wr_tile<n, 1>(xa, zb, lw, pitch, Nu, Nr, l4, o, bbq);                               \
                        else bm_wr_tile<n, 0>(xa, zb, lw, pitch, Nu, Nr, l4, o, bbq);                                   \
                        _Pragma("unroll") for (int ct = 0; ct < n; ct++) _Pragma("unroll") for (int r = 0; r < 4; r++) acc[ct][r] = o[ct][r]; \
                    } break;
                    switch (nct) { BM_WR_CASE(1) BM_WR_CASE(2) BM_WR_CASE(3) default: BM_WR_CASE(4) }
#undef BM_WR_CASE
#pragma unroll
                    for (int ct = 0; ct < MAXCT; ct++) {
                        if (ct >= nct) continue;
                        const int col = ct * 16 + l15;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int i = ft * 16 + l4 + 4 * r;
                            if (i < Ku) {
                                if (col < pn) gBP[(size_t)i * ldu + p0 + col] = acc[ct][r] / gsc[i];
                                else if (ext && col == pn) gze[i] = acc[ct][r];
                            }
                        }
                    }
                    if (ext) {                                 // the four sample groups of a feature sit 16 lanes apart
                        bbq += __shfl_xor(bbq, 16, 64); bbq += __shfl_xor(bbq, 32, 64);
                        if (l4 == 0 && il < Ku) gbb[il] = bbq;
                    }
                }
                PHX_END(t_mm, PH_MATVEC);
                p0 += pn;
            }
            blk_sync(B);
            return 1;
        }
    }
    // samples too many for a 16-column tile in LDS: vector-ALU version, as many columns as fit at a time
    int pcn = B.pool_n / N;
    if (pcn > M) pcn = M;
    if (pcn >= 1) {
        double *Z = B.pool;                                    // [pcn][N]
        for (int p0 = 0; p0 < M; p0 += pcn) {
            const int pn = M - p0 < pcn ? M - p0 : pcn;
            blk_sync(B);
            for (int e = B.tid; e < N * pn; e += B.nthr) {
                const int pc = e / N, h = e - pc * N, p = p0 + pc;
                Z[e] = W.w[h] * BM_PHI(p, h);
            }
            blk_sync(B);
            for (int i = B.wave; i < K; i += B.nwave) {
                const double *x = F.X + (size_t)i * N;
                const double rsc = 1.0 / F.scale[i];
                if (N <= 8 * 64) {                             // the design column lives in registers for all pn columns
                    double xr[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) xr[k] = (B.lane + 64 * k < N) ? x[B.lane + 64 * k] : 0.0;
                    for (int pc = 0; pc < pn; pc += 8) {       // eight columns per reduction tree (blk.h: wave_sum8)
                        double a[8];
#pragma unroll
                        for (int c = 0; c < 8; c++) {
                            const double *z = Z + (pc + c < pn ? pc + c : pn - 1) * N;
                            double t = 0;
#pragma unroll
                            for (int k = 0; k < 8; k++) if (B.lane + 64 * k < N) t += xr[k] * z[B.lane + 64 * k];
                            a[c] = t;
                        }
                        wave_sum8(a, B.lane);
                        const int c = B.lane >> 3;
                        if ((B.lane & 7) == 0 && pc + c < pn) W.BP[(size_t)i * ld + p0 + pc + c] = a[0] * rsc;
                    }
                } else {
                    for (int pc = 0; pc < pn; pc++) {
                        double a = 0;
                        for (int h = B.lane; h < N; h += 64) a += x[h] * Z[pc * N + h];
                        a = wave_sum(a);
                        if (B.lane == 0) W.BP[(size_t)i * ld + p0 + pc] = a * rsc;
                    }
                }
            }
        }
    } else {
        for (int i = B.wave; i < K; i += B.nwave) {            // samples do not fit in LDS: columns from memory
            const double *x = F.X + (size_t)i * N;
            for (int p = 0; p < M; p++) {
                double a = 0;
                if (p == 0) { for (int h = B.lane; h < N; h += 64) a += x[h] * W.w[h]; }
                else {
                    const int u = W.used[p - 1];
                    for (int h = B.lane; h < N; h += 64) a += (x[h] * W.w[h]) * bm_col(F, W, N, u, h);
                }
                a = wave_sum(a);
                if (B.lane == 0) W.BP[(size_t)i * ld + p] = a / F.scale[i];
            }
        }
    }
#endif
    blk_sync(B);
    return 0;
}

#ifndef PAREBEN_HOST_EMUL
// S_in = bb_i / |x_i|^2 - BP_i' Sigma BP_i and Q_in = ze_i / |x_i| for every feature (NEmainEff.c:1732-1762), the
// quadratic forms on the FP64 matrix cores: a wave owns 16 features; T = BP_tile * Sigma in 16 x 16 tiles (A operand:
// lane l holds BP[feature l & 15][k0 + (l >> 4)], B operand Sigma[k0 + (l >> 4)][16 ct + (l & 15)], 64 columns of Sigma
// per round), folded with BP on the fly: D register r of lane l is T[feature (l >> 4) + 4 r][column l & 15], multiplied
// by the same BP entry and summed over the 16 lanes of a row group.
DEVNI void bm_quad_features(const Blk &B, const FoldDev &F, const BmWork &W, int K, int M)
{
    typedef double bd4 __attribute__((ext_vector_type(4)));
    const int ld = W.ld, l15 = B.lane & 15, l4 = B.lane >> 4;
    const int nct = (M + 15) >> 4;
    for (int ft = B.wave; ft * 16 < K; ft += B.nwave) {
        const int ia = ft * 16 + l15;
        const double *bpa = W.BP + (size_t)(ia < K ? ia : K - 1) * ld;
        double q[4] = {0, 0, 0, 0};
        for (int c0 = 0; c0 < nct; c0 += 4) {
            bd4 acc[4];
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = bd4{0, 0, 0, 0};
            for (int k0 = 0; k0 < M; k0 += 4) {
                const int k = k0 + l4;
                const double a = (k < M && ia < K) ? bpa[k] : 0.0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int col = (c0 + c) * 16 + l15;
                    if (c0 + c < nct) {
                        const double b = (k < M && col < M) ? W.Sig[(size_t)k * ld + col] : 0.0;
                        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int col = (c0 + c) * 16 + l15;
                if (c0 + c < nct && col < M) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int i = ft * 16 + l4 + 4 * r;
                        if (i < K) q[r] += acc[c][r] * W.BP[(size_t)i * ld + col];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = q[r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            const int i = ft * 16 + l4 + 4 * r;
            if (l15 == 0 && i < K) {
                const double sc = F.scale[i];
                W.Sin[i] = W.bb[i] / (sc * sc) - v;
                W.Qin[i] = W.aroot[i] / sc;
            }
        }
    }
    blk_sync(B);
}
#endif

// posterior mode, :1808-2010.  Leaves w, Sig (= H^-1), H from its last Hessian evaluation.
DEVNI int bm_postmode(const Blk &B, const FoldDev &F, const BmWork &W, GmScalars &S)
{
    const int N = F.N, M = S.M, ld = W.ld;
    const double step_min = 1.0 / 256.0;
    bm_phi_mu(B, F, W, M, W.mu, W.pm);
    double derr = bm_data_error(B, F, W);
    double rp = 0;
    PAR(i, M) if (i >= 1) rp += W.A[i - 1] * W.mu[i] * W.mu[i] / 2;
    double total = blk_sum(B, rp) + derr;
    for (int it = 0; it < 25; it++) {
        const double elog = total;
        double gp = 0, hp = 0;
        PAR(h, N) {
            double y = W.yv[h];
            if (S.v.epis) {                                    // NeFull.c:1042-1043
                if (y < 1e-5) y = 1e-5;
                if (y > (1 - 1e-5)) y = 1 - 1e-5;
            }
            const double e = F.y[h] - y;
            W.e[h] = e;
            double b = y * (1 - y);
            if (S.v.epis) { if (b < 1e-5) b = 1e-3; if (b > 1e5) b = 1e3; }       // NeFull.c:1047-1048
            else { if (b < 1e-10) b = 1e-5; if (b > 1e10) b = 1e5; }             // NEmainEff.c:1878-1883
            W.w[h] = b;
            gp += e; hp += b;
        }
        const double g0 = blk_sum(B, gp), h0 = blk_sum(B, hp);
        // gradient and first Hessian row/column: one wavefront per model column
#ifdef PAREBEN_HOST_EMUL
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
#else
        for (int j = 1 + B.wave; j < M; j += B.nwave) {
            double ga = 0, ha = 0;
            for (int h = B.lane; h < N; h += 64) { const double ph = BM_PHI(j, h); ga += W.e[h] * ph; ha += W.w[h] * ph; }
            ga = wave_sum(ga); ha = wave_sum(ha);
            if (B.lane == 0) { W.g[j] = ga - W.A[j - 1] * W.mu[j]; W.H[j] = ha; W.H[(size_t)j * ld] = ha; }
        }
        if (M - 1 >= 8) {
            // Phi' diag(w) Phi on the FP64 matrix cores: 16 x 16 tiles over the model columns 1 .. M-1, lower-triangle tile
            // pairs dealt to the waves; A operand (phi_j w) and B operand phi_k straight from the design (lane l: column
            // l & 15 of the tile, sample h0 + (l >> 4)), one fma chain over the samples in ascending order -- the product is
            // associated as the reference does, (phi_j * w) * phi_k (:1911)
            typedef double bd4 __attribute__((ext_vector_type(4)));
            const int Mm = M - 1, nT = (Mm + 15) >> 4, l15 = B.lane & 15, l4 = B.lane >> 4;
            const int Nr = (N + 3) & ~3;
            for (int q = B.wave; q < nT * (nT + 1) / 2; q += B.nwave) {
                int tj = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
                while ((tj + 1) * (tj + 2) / 2 <= q) tj++;
                while (tj * (tj + 1) / 2 > q) tj--;
                const int tk = q - tj * (tj + 1) / 2;              // tk <= tj
                const int ja = tj * 16 + l15, kb = tk * 16 + l15;  // model columns (0-based among 1 .. M-1) of this lane's operands
                const int uj = W.used[ja < Mm ? ja : Mm - 1], uk = W.used[kb < Mm ? kb : Mm - 1];
                const double *xj = F.X + (size_t)uj * N, *xk = F.X + (size_t)uk * N;
                const double sj = W.phi_div ? F.scale[uj] : F.rscale[uj], sk = W.phi_div ? F.scale[uk] : F.rscale[uk];
                const bool dv = W.phi_div != 0;
                bd4 acc = bd4{0, 0, 0, 0};
                constexpr int RS = 8;                              // eight steps' operand loads issued together, one round ahead
                double xjn[RS], xkn[RS], wn[RS];
#pragma unroll
                for (int u = 0; u < RS; u++) { const int h = 4 * u + l4, hc = h < N ? h : N - 1; xjn[u] = xj[hc]; xkn[u] = xk[hc]; wn[u] = W.w[hc]; }
                for (int h0 = 0; h0 < Nr; h0 += 4 * RS) {
                    double xjc[RS], xkc[RS], wc[RS];
#pragma unroll
                    for (int u = 0; u < RS; u++) { xjc[u] = xjn[u]; xkc[u] = xkn[u]; wc[u] = wn[u]; }
#pragma unroll
                    for (int u = 0; u < RS; u++) { const int h = h0 + 4 * RS + 4 * u + l4, hc = h < N ? h : N - 1; xjn[u] = xj[hc]; xkn[u] = xk[hc]; wn[u] = W.w[hc]; }
#pragma unroll
                    for (int u = 0; u < RS; u++) {
                        const int h = h0 + 4 * u + l4;
                        if (h0 + 4 * u < Nr) {
                            const double pj = dv ? xjc[u] / sj : xjc[u] * sj, pk = dv ? xkc[u] / sk : xkc[u] * sk;
                            const double a = (h < N && ja < Mm) ? pj * wc[u] : 0.0;
                            const double b = (h < N && kb < Mm) ? pk : 0.0;
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int jm = tj * 16 + l4 + 4 * r, km = tk * 16 + l15;
                    if (jm < Mm && km < Mm && km <= jm) {
                        double v = acc[r];
                        if (jm == km) v += W.A[km];
                        W.H[(size_t)(km + 1) * ld + jm + 1] = v; W.H[(size_t)(jm + 1) * ld + km + 1] = v;
                    }
                }
            }
        } else {   // lower triangle of Phi' diag(w) Phi, one wavefront per (j, k) pair
            const int np = (M - 1) * M / 2;
            for (int q = B.wave; q < np; q += B.nwave) {
                int j = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
                while ((j + 1) * (j + 2) / 2 <= q) j++;
                while (j * (j + 1) / 2 > q) j--;
                const int k = q - j * (j + 1) / 2 + 1;
                j += 1;                                        // 1 <= k <= j <= M-1
                const int uj = W.used[j - 1], uk = W.used[k - 1];
                double a = 0;
                for (int h = B.lane; h < N; h += 64) a += bm_col(F, W, N, uj, h) * W.w[h] * bm_col(F, W, N, uk, h);
                a = wave_sum(a);
                if (B.lane == 0) {
                    if (j == k) a += W.A[k - 1];
                    W.H[(size_t)k * ld + j] = a; W.H[(size_t)j * ld + k] = a;
                }
            }
        }
#endif
        if (B.tid == 0) { W.g[0] = g0; W.H[0] = h0; }
        blk_sync(B);
        for (int j = B.wave; j < M; j += B.nwave)
            for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] = W.H[(size_t)j * ld + i];
        blk_sync(B);
        {
            const GmWork G = bm_as_gm(W);
            if (gm_spd_inverse(B, G, M)) return 1;
        }
        blk_sync(B);
        int cp = 0;
        PAR(j, M) if ((j >= 1 || S.v.epis) && fabs(W.g[j]) < 1e-6) cp++;   // NeFull.c:1085-1099 counts the intercept's entry too
        const int cnt = blk_isum(B, cp);
        if (cnt == (S.v.epis ? M : M - 1)) break;
        PAR(k, M) {
            double a = 0;
            for (int L = 0; L < M; L++) a += W.g[L] * W.Sig[(size_t)L * ld + k];
            W.dmu[k] = a;
        }
        blk_sync(B);
        double step = 1;
        while (step > step_min) {
            PAR(j, M) W.mnew[j] = W.mu[j] + step * W.dmu[j];
            blk_sync(B);
            bm_phi_mu(B, F, W, M, W.mnew, W.pm);
            derr = bm_data_error(B, F, W);
            double rq = 0;
            PAR(j, M) if (j >= 1) rq += W.A[j - 1] * W.mnew[j] * W.mnew[j] / 2;
            total = derr + blk_sum(B, rq);
            if (total >= elog) step = step / 2;
            else {
                PAR(j, M) W.mu[j] = W.mnew[j];
                blk_sync(B);
                step = 0;
            }
        }
    }
    blk_sync(B);
    return 0;
}

// full statistics, :1633-1803
DEVNI int bm_fullstat(const Blk &B, const FoldDev &F, const BmWork &W, int K, GmScalars &S)
{
    const int N = F.N, ld = W.ld;
    S.bp_ok = 0;                                               // the weights change: cached weighted rows are stale
    { PH_BEGIN(); const int bad = bm_postmode(B, F, W, S); PH_END(PH_INVERSE); if (bad) return 1; }
    const int M = S.M;
    PH_BEGIN();
    bm_phi_mu(B, F, W, M, W.mu, W.pm);
    PAR(h, N) { const double y = 1 / (1 + exp(-W.pm[h])); W.e[h] = F.y[h] - y; }
    blk_sync(B);
    const int have_stats = bm_weighted_rows(B, F, W, K, M, true, S.ph);
    S.bp_ok = M;
    PH_END(PH_FS_FEAT);
    // S_in = x_i' diag(w) x_i / |x_i|^2 - BP_i' Sigma BP_i ;  Q_in = x_i' e / |x_i|
    long long ph_t1_ = 0;
#if defined(PAREBEN_PHASE_TIMERS) && !defined(PAREBEN_HOST_EMUL)
    if (B.tid == 0) ph_t1_ = (long long)wall_clock64();
#endif
#ifdef PAREBEN_HOST_EMUL
    for (int i = 0; i < K; i++) {
        double bbq = 0, ze = 0;
        for (int h = 0; h < N; h++) { const double x = F.X[(size_t)i * N + h]; bbq += W.w[h] * (x * x); ze += x * W.e[h]; }
        W.bb[i] = bbq; W.aroot[i] = ze;        // scratch: aroot is rewritten by every dML pass
    }
#else
    if (!have_stats) {
        for (int i = B.wave; i < K; i += B.nwave) {
            const double *x = F.X + (size_t)i * N;
            double bbq = 0, ze = 0;
            for (int h = B.lane; h < N; h += 64) { const double xv = x[h]; bbq += W.w[h] * (xv * xv); ze += xv * W.e[h]; }
            bbq = wave_sum(bbq); ze = wave_sum(ze);
            if (B.lane == 0) { W.bb[i] = bbq; W.aroot[i] = ze; }
        }
    }
#endif
    blk_sync(B);
#ifdef PAREBEN_HOST_EMUL
    (void)have_stats;
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
#else
    bm_quad_features(B, F, W, K, M);
#endif
    PAR(i, K) {
        const double s = W.Sin[i], q = W.Qin[i];
        const int l = W.upos[i];
        if (l >= 0) { const double a = W.A[l]; W.Sout[i] = a * s / (a - s); W.Qout[i] = a * q / (a - s); }
        else { W.Sout[i] = s; W.Qout[i] = q; }
    }
    blk_sync(B);
#if defined(PAREBEN_PHASE_TIMERS) && !defined(PAREBEN_HOST_EMUL)
    if (B.tid == 0) S.ph[PH_FS_REST] += (long long)wall_clock64() - ph_t1_;
#endif
    (void)ph_t1_;
    CNT(c.n_fullstat++; c.sum_m_full += M; c.sum_m2_full += (int64_t)M * M);
    return 0;
}

// dML / action choice, :2063-2238 (Q15: only delete-priority can fire)
DEVNI int bm_delta_ml(const Blk &B, const BmWork &W, int K, int N, int NU, double lambda, double alpha,
                      int epis, int *any_del_out, double *best)
{
    const double l1 = lambda * alpha, l2 = lambda * (1 - alpha);
    int prio_add = 0, prio_del = 0;
    if (NU < 10) { prio_add = 1; prio_del = 0; }
    if (NU > 100 || (!epis && NU >= N)) { prio_add = 0; prio_del = 1; }   // NeFull.c:1805-1809 has no N clause
    int my_del = 0;
    PAR(i, K) {
        const int l = W.upos[i];
        if (l == UP_LOST) { W.act[i] = ACT_NONE; continue; }
        const double so = W.Sout[i], qo = W.Qout[i];
        double d_ml = 0;
        int act = ACT_NONE;
        const double a = so - qo * qo + 2 * l1 + l2;
        const double bq = (so + l2) * (so + 4 * l1 + l2);
        const double g = 2 * l1 * (so + l2) * (so + l2);
        const double disc = bq * bq - 4 * a * g;
        if (a < 0 && disc > 0) {
            const double r = (-bq - sqrt(disc)) / (2 * a);
            const double L = (log(r / (r + so + l2)) + qo * qo / (r + so + l2)) * 0.5 - l1 / r;
            if (L > 0) {
                W.aroot[i] = r + l2;
                if (l >= 0) {
                    act = ACT_REEST;
                    const double o = W.A[l] - l2;
                    d_ml = 0.5 * (log(r * (o + so + l2) / (o * (r + so + l2))) +
                                  qo * qo * (1 / (r + so + l2) - 1 / (o + so + l2))) -
                           l1 * (1 / r - 1 / o);
                } else { act = ACT_ADD; d_ml = L; }
            }
        } else if (l >= 0 && NU > 1) {
            my_del = 1;
            act = ACT_DEL;
            const double o = W.A[l] - l2;
            const double L = (log(o / (o + so + l2)) + qo * qo / (o + so + l2)) * 0.5 - l1 / o;
            d_ml = -L;
        }
        W.act[i] = (signed char)act;
        W.dml[i] = d_ml;
    }
    const int any_del = blk_or(B, my_del);
    *any_del_out = any_del;
    bool rescanned = false;
    if (any_del && prio_del) {
        PAR(i, K) {
            const int act = W.act[i];
            if (act == ACT_REEST) W.dml[i] = 0;
            else if (act == ACT_ADD) { if (!prio_add) W.dml[i] = 0; }
        }
        rescanned = true;
    }
    blk_sync(B);
    double v = 0; int idx = 0x7fffffff;
    PAR(i, K) {
        if (!rescanned && W.upos[i] == UP_LOST) continue;
        const double d = W.dml[i];
        if (d > v) { v = d; idx = i; }
    }
    double bv; int bi;
    blk_argmax(B, v, idx, &bv, &bi);
    if (!(bv > 0)) { bv = 0; bi = 0; }
    *best = bv;
    return bi;
}

DEVNI int bm_collect(const Blk &B, const BmWork &W, int K, double cutoff)
{
    int base = 0;
    for (int i0 = 0; i0 < K; i0 += B.nthr) {
        const int i = i0 + B.tid;
        const int f = (i < K && W.dml[i] >= cutoff) ? 1 : 0;
        int tot;
        const int off = blk_scan_excl(B, f, &tot);
        if (f) W.todo[base + off] = i;
        base += tot;
    }
    blk_sync(B);
    return base;
}

// add feature nu, :830-1003 + :701-711
DEVNI void bm_add(const Blk &B, const FoldDev &F, const BmWork &W, int K, GmScalars &S, int nu, double newA)
{
    const int N = F.N, M = S.M, ld = W.ld, NU = M - 1;
    PAR(h, N) W.bphi[h] = W.w[h] * bm_col(F, W, N, nu, h);
    blk_sync(B);
    // bb[i] = x_i' (w .* phi) / |x_i| ; tmp[p] = Phi_p' (w .* phi)
#ifdef PAREBEN_HOST_EMUL
    for (int i = 0; i < K; i++) { double a = 0; for (int h = 0; h < N; h++) a += F.X[(size_t)i * N + h] * W.bphi[h]; W.bb[i] = a / F.scale[i]; }
    for (int p = 0; p < M; p++) { double a = 0; for (int h = 0; h < N; h++) a += BM_PHI(p, h) * W.bphi[h]; W.tmp[p] = a; }
#else
    // eight features per wave and reduction tree: their loads are in flight together (a feature at a time paid a memory
    // round trip per feature); wave_sum8 pairs lanes exactly like wave_sum, so the sums are the same bits
    for (int i0 = B.wave * 8; i0 < K; i0 += B.nwave * 8) {
        double a[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const double *x = F.X + (size_t)(i0 + c < K ? i0 + c : K - 1) * N;
            double t = 0;
            for (int h = B.lane; h < N; h += 64) t += x[h] * W.bphi[h];
            a[c] = t;
        }
        wave_sum8(a, B.lane);
        const int i = i0 + (B.lane >> 3);
        if ((B.lane & 7) == 0 && i < K) W.bb[i] = a[0] / F.scale[i];
    }
    for (int p = B.wave; p < M; p += B.nwave) {
        double a = 0;
        for (int h = B.lane; h < N; h += 64) a += BM_PHI(p, h) * W.bphi[h];
        a = wave_sum(a);
        if (B.lane == 0) W.tmp[p] = a;
    }
#endif
    blk_sync(B);
    PAR(i, M) {
        double a = 0;
        for (int j = 0; j < M; j++) a += W.Sig[(size_t)j * ld + i] * W.tmp[j];
        W.tp[i] = a;
    }
    const double sii = 1.0 / (newA + W.Sin[nu]);
    const double mui = sii * W.Qin[nu];
    blk_sync(B);
    // rows against the OLD model columns: still current from the last full-stat pass / action unless the
    // weights changed in between (they only change in the posterior-mode step)
    if (S.bp_ok != M) bm_weighted_rows(B, F, W, K, M);
    PAR(i, K) {
        const double *bp = W.BP + (size_t)i * ld;
        double t = 0;
        for (int j = 0; j < M; j++) t += bp[j] * W.tp[j];
        const double mc = W.bb[i] - t;
        W.Sin[i] = W.Sin[i] - mc * mc * sii;
        W.Qin[i] = W.Qin[i] - mui * mc;
        W.BP[(size_t)i * ld + M] = W.bb[i];                    // the new model column's weighted row entry
    }
    S.bp_ok = M + 1;
    PAR(i, M) W.mu[i] += -mui * W.tp[i];
    for (int j = B.wave; j < M; j += B.nwave) {
        const double f = sii * W.tp[j];
        for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] += f * W.tp[i];
    }
    PAR(i, M) {
        const double si = -sii * W.tp[i];
        W.Sig[(size_t)M * ld + i] = si;
        W.Sig[(size_t)i * ld + M] = si;
    }
    if (B.tid == 0) {
        W.Sig[(size_t)M * ld + M] = sii;
        W.A[NU] = newA;
        W.mu[M] = mui;
        W.used[NU] = nu;
        W.upos[nu] = NU;
    }
    S.M = M + 1;
    blk_sync(B);
}

// delete used slot jj (feature nu), :1010-1121 + :728-744
DEVNI void bm_delete(const Blk &B, const FoldDev &F, const BmWork &W, int K, GmScalars &S, int jj, int nu)
{
    const int M = S.M, ld = W.ld, last = M - 1, j1 = jj + 1;
    PAR(i, M) W.tp[i] = W.Sig[(size_t)j1 * ld + i];
    blk_sync(B);
    const double sjj = W.tp[j1];
    const double mujj = W.mu[j1];
    const int gone = W.used[jj];
    if (S.bp_ok != M) bm_weighted_rows(B, F, W, K, M);
    PAR(i, K) {
        const double *bp = W.BP + (size_t)i * ld;
        double t = 0;
        for (int j = 0; j < M; j++) t += bp[j] * W.tp[j];
        W.Sin[i] = W.Sin[i] + t * t / sjj;
        W.Qin[i] = W.Qin[i] + t * mujj / sjj;
        W.BP[(size_t)i * ld + j1] = W.BP[(size_t)i * ld + last];   // the last model column moves into the freed slot
    }
    S.bp_ok = last;
    PAR(i, M) W.mu[i] = W.mu[i] - mujj * W.tp[i] / sjj;
    for (int j = B.wave; j < M; j += B.nwave) {
        const double vj = W.tp[j];
        if (S.v.epis) for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] -= W.tp[i] / sjj * vj;   // NeFull.c:1606
        else          for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] -= W.tp[i] * vj / sjj;   // NEmainEff.c:1069
    }
    blk_sync(B);
    if (j1 != last) {
        PAR(i, M) { W.v3[i] = W.Sig[(size_t)last * ld + i]; W.v4[i] = W.Sig[(size_t)i * ld + last]; }
        blk_sync(B);
        PAR(i, last) {
            if (i != j1) { W.Sig[(size_t)j1 * ld + i] = W.v3[i]; W.Sig[(size_t)i * ld + j1] = W.v4[i]; }
        }
        if (B.tid == 0) {
            W.Sig[(size_t)j1 * ld + j1] = W.v3[last];
            W.A[jj] = W.A[last - 1];
            W.mu[j1] = W.mu[last];
            W.used[jj] = W.used[last - 1];
            W.upos[W.used[last - 1]] = jj;
        }
    }
    if (B.tid == 0) W.upos[gone] = (gone == nu) ? UP_FREE : UP_LOST;
    S.M = last;
    blk_sync(B);
}

// re-estimate used slot jj, :1127-1203 (S/Q update reads the NEW Sigma row j1)
DEVNI void bm_reestimate(const Blk &B, const FoldDev &F, const BmWork &W, int K, GmScalars &S, int jj, double newA)
{
    const int M = S.M, ld = W.ld, j1 = jj + 1;
    PAR(i, M) W.tp[i] = W.Sig[(size_t)j1 * ld + i];
    blk_sync(B);
    const double oldA = W.A[jj];
    const double dinv = 1.0 / (newA - oldA);
    const double kappa = 1.0 / (W.tp[j1] + dinv);
    const double mujj = W.mu[j1];
    blk_sync(B);
    if (B.tid == 0) W.A[jj] = newA;
    PAR(i, M) W.mu[i] += (-mujj * kappa) * W.tp[i];
    for (int j = B.wave; j < M; j += B.nwave) {
        const double f = kappa * W.tp[j];
        for (int i = B.lane; i < M; i += BLK_LANES) W.Sig[(size_t)j * ld + i] -= f * W.tp[i];
    }
    blk_sync(B);
    PAR(i, M) W.tmp[i] = W.Sig[(size_t)j1 * ld + i];           // the updated row
    blk_sync(B);
    if (S.bp_ok != M) bm_weighted_rows(B, F, W, K, M);
    PAR(i, K) {
        const double *bp = W.BP + (size_t)i * ld;
        double t = 0;
        for (int j = 0; j < M; j++) t += bp[j] * W.tmp[j];
        W.Sin[i] = W.Sin[i] + t * t * kappa;
        W.Qin[i] = W.Qin[i] + mujj * kappa * t;
    }
    blk_sync(B);
}

// one call of the inner routine, :397-827.  *loglik = training log-likelihood at exit.
DEV int bm_inner(const Blk &B, const FoldDev &F, const BmWork &W, int K, double lambda, double alpha,
                 GmScalars &S, int iter, double *loglik)
{
    const int N = F.N;
    const bool first = iter <= 1;
    if (first) {                                               // :1225-1385
        S.M = 2;
        PAR(i, K) W.upos[i] = UP_FREE;
        blk_sync(B);
        double pa = 0, pb = 0, pc = 0, pd = 0;
        PAR(h, N) {
            const double tp = -1 + 2 * F.y[h];
            const double lo = log(((tp * 0.9 + 1) / 2) / (1 - (tp * 0.9 + 1) / 2));
            const double ph = bm_col(F, W, N, 0, h);
            pa += ph; pb += ph * ph; pc += lo; pd += ph * lo;
        }
        const double sa = blk_sum(B, pa), sb = blk_sum(B, pb), sc = blk_sum(B, pc), sd = blk_sum(B, pd);
        if (B.tid == 0) {
            W.used[0] = 0; W.upos[0] = 0;
            const double det = (double)N * sb - sa * sa;
            double m0, m1;
            if (fabs(det) > 1e-10 * N * (sb > 0 ? sb : 1)) { m0 = (sb * sc - sa * sd) / det; m1 = ((double)N * sd - sa * sc) / det; }
            else { const double c0 = sa / N, den = (double)N * (1 + c0 * c0); m0 = sc / den; m1 = c0 * sc / den; }
            W.mu[0] = m0; W.mu[1] = m1;
            double a0 = (m1 == 0) ? 1 : 1 / (m1 * m1);
            if (a0 < 1e-3) a0 = 1e-3;
            if (a0 > 1e3) a0 = 1e3;
            W.A[0] = a0;
        }
    } else {
        PAR(i, K) if (W.upos[i] == UP_LOST) W.upos[i] = UP_FREE;
    }
    blk_sync(B);
    const int initial = W.used[0];
    int ini_removed = first ? 0 : 1;
    if (bm_fullstat(B, F, W, K, S)) { S.status |= ST_CHOLESKY | ST_ABORT; return 1; }
    int sel = ACT_NONE, jj = -1, n_todo = 0, last_it = 0, i_iter = 0;
    const int it_max = iter == 1 ? 10 : 100;
    double ll = 1e-30, ll0;
    while (!last_it) {
        i_iter++;
        CNT(c.n_inner++);
        ll0 = ll;
        double best; int any_del;
        int nu;
        { PH_BEGIN(); nu = bm_delta_ml(B, W, K, N, S.M - 1, lambda, alpha, S.v.epis, &any_del, &best); PH_END(PH_DML); }
        int worthwhile;
        if (sel == ACT_TERM && !ini_removed && S.M > 2) nu = -1;
        if (nu == -1 && ini_removed) { worthwhile = 0; sel = ACT_TERM; }
        else if (nu == -1 && !ini_removed && S.M > 2) {
            worthwhile = 1;
            nu = initial;
            if (B.tid == 0) { W.act[nu] = ACT_DEL; W.todo[0] = initial; }
            blk_sync(B);
            n_todo = 1; ini_removed = 1; sel = ACT_DEL;
        } else {
            worthwhile = 1;
            const int act_nu = W.act[nu];
            double cutoff = best * (act_nu == ACT_ADD ? S.v.n_add : 1.0);     // NEmainEff.c:422 0.90 | NeFull.c:254 0.99
            if (cutoff < 0.001) cutoff = 0.001;
            n_todo = bm_collect(B, W, K, cutoff);
            if (act_nu == ACT_DEL && n_todo > 1) n_todo = 1;
            if (n_todo == 0) worthwhile = 0;
        }
        if (!worthwhile) sel = ACT_TERM;
        if (worthwhile) {
            for (int u = 0; u < n_todo; u++) {
                nu = W.todo[u];
                sel = W.act[nu];
                const double newA = W.aroot[nu];
                if (sel == ACT_REEST || sel == ACT_DEL) {
                    const int l = W.upos[nu];
                    if (l >= 0) jj = l;
                    else { S.status |= ST_STALE; if (jj < 0 || jj >= S.M - 1) { S.status |= ST_ABORT; return 1; } }
                }
                if (sel == ACT_REEST && fabs(log(newA) - log(W.A[jj])) <= 1e-3 && any_del == 0) sel = ACT_TERM;
                blk_sync(B);
                PH_BEGIN();
                if (sel == ACT_REEST) {
                    CNT(c.n_reest++; c.sum_m_action += S.M);
                    bm_reestimate(B, F, W, K, S, jj, newA);
                } else if (sel == ACT_ADD) {
                    if (S.M + 1 > W.cap) { S.status |= ST_OVERFLOW | ST_ABORT; return 1; }
                    // NeFull.c:549-553: more bases than the R wrapper's bMax (the reference tests this from the second outer
                    // iteration on and overruns its arrays in the first; here the fit is stopped either way)
                    if (S.v.epis && S.M > W.bmax) { S.status |= ST_OVERFLOW | ST_ABORT; return 1; }
                    CNT(c.n_add++; c.sum_m_action += S.M);
                    bm_add(B, F, W, K, S, nu, newA);
                } else if (sel == ACT_DEL) {
                    CNT(c.n_del++; c.sum_m_action += S.M);
                    bm_delete(B, F, W, K, S, jj, nu);
                    if (nu == initial) ini_removed = 1;
                }
                PH_END(PH_ACTION);
                CNT(if (S.M > c.m_max) c.m_max = S.M);
                if (u == n_todo - 1) {                         // :749-762
                    if (bm_fullstat(B, F, W, K, S)) { S.status |= ST_CHOLESKY | ST_ABORT; return 1; }
                }
            }
        }
        if (sel == ACT_TERM && ini_removed) last_it = 1;
        if ((i_iter == it_max && S.M == 2) || i_iter > it_max) last_it = 1;
        if (i_iter == it_max) sel = ACT_TERM;
        bm_phi_mu(B, F, W, S.M, W.mu, W.pm);
        double lp = 0;
        PAR(h, N) {
            const double ex = exp(W.pm[h]);
            lp += F.y[h] * log(ex / (1 + ex)) + (1 - F.y[h]) * log(1 / (1 + ex));
        }
        ll = blk_sum(B, lp);
        const double dL = fabs((ll - ll0) / ll0);
        if (dL < 1e-3) sel = ACT_TERM;
    }
    *loglik = ll;
    return 0;
}

// The whole fit, :236-389.  On return W.mu[0] is the intercept, W.mu[l+1] the weight of used[l]
// (normalised-column units), S.M the model size incl. the intercept.
DEV void bm_fit(const Blk &B, const FoldDev &F, const BmWork &W, int K, double lambda, double alpha,
                GmScalars &S, double *loglik)
{
    S.status = 0; S.M = 2; S.beta = 0; S.b = 0;
    S.v.n_add = S.v.epis ? 0.99 : 0.90;
    CNT(c = FitCounters{});
    PAR(i, W.ld) W.A[i] = 0;                                   // Calloc'd per fit, never cleared after (Q12)
    blk_sync(B);
    double vk = 1e-30, vk0, err = 1000, ll = 0;
    int iter = 0;
    while (iter < 100 && err > 1e-8) {
        iter++;
        vk0 = vk;
        if (bm_inner(B, F, W, K, lambda, alpha, S, iter, &ll)) break;
        double ap = 0;
        if (S.v.epis) { PAR(i, S.M - 1) ap += W.A[i]; }        // NeFull.c:133-134: the M-1 precisions
        else { PAR(i, S.M) ap += fabs(W.A[i]); }               // NEmainEff.c:340: dasum over M = N_used + 1 entries (Q12)
        vk = blk_sum(B, ap);
        err = fabs(vk - vk0) / S.M;
        if (S.outer_log && B.tid == 0) { double *o = S.outer_log + 3 * (iter - 1); o[0] = err; o[1] = 0; o[2] = 0; }   // NEmainEff.c:342
    }
    *loglik = ll;
    CNT(c.n_outer = iter; c.m_final = S.M - 1; c.status = S.status);
    blk_sync(B);
}

// fold score, R/GetModelError.R:34-57: mean Bernoulli log-likelihood of the held-out rows
DEV double bm_fold_loglik(const Blk &B, const FoldDev &F, const BmWork &W, const GmScalars &S)
{
    const int nte = F.nte, M = S.M;
    // eta -> exp(eta), with R's clamp applied only when max/min cross the thresholds
    double mx = -1e300, mn = 1e300;
    int any = 0;
    for (int j = 1; j < M; j++) if (W.mu[j] / F.scale[W.used[j - 1]] != 0) any = 1;
    if (!any) return 0.0;                                      // null model: logL <- 0 (:42-43)
    PAR(h, nte) {
        double eta = 0;
        for (int j = 1; j < M; j++) { const int u = W.used[j - 1]; eta += F.Xte[(size_t)u * nte + h] * (W.mu[j] / F.scale[u]); }
        const double t = exp(W.mu[0] + eta);
        W.pm[h] = t;
        if (t > mx) mx = t;
        if (t < mn) mn = t;
    }
    double dummy; int di;
    blk_argmax(B, mx, 0, &mx, &di);
    blk_argmax(B, -mn, 0, &dummy, &di); mn = -dummy;
    blk_sync(B);
    double part = 0;
    PAR(h, nte) {
        double t = W.pm[h];
        if (mx > 1e10 && t > 1e10) t = 1e5;
        if (mn < 1e-10 && t < 1e-10) t = 1e-5;
        part += F.yte[h] * log(t / (1 + t)) + (1 - F.yte[h]) * log(1 / (1 + t));
    }
    return blk_sum(B, part) / nte;
}
