// bm_dev.h -- the MI355X mapping of the phases of a binomial fit (bm_fit.h includes it after its prelude): the weighted
// rows X' diag(w) Phi, the per-feature quadratic forms BP Sigma BP' and the Newton Hessian Phi' diag(w) Phi on the FP64
// matrix cores, Phi mu with staged index pairs.  gfx950 only.
#pragma once

#if defined(PAREBEN_PHASE_TIMERS)
#define PHX2_BEGIN(v) long long v = (B.tid == 0) ? (long long)wall_clock64() : 0
#define PHX2_END(v, k) do { if (B.tid == 0) S.ph[k] += (long long)wall_clock64() - v; } while (0)
#else
#define PHX2_BEGIN(v) do {} while (0)
#define PHX2_END(v, k) do {} while (0)
#endif

// pm[h] = sum_p Phi_p[h] * mu[p]
DEVNI void bm_phi_mu(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int M, const double *mu, double *out)
{
    const int N = F.N;
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
    bm_phi_mu_plain(B, F, W, M, mu, out);
}

// t_i = sum_j BP[i][j] * vec[j] (j ascending: one fma chain per feature, the reference's loop order) for every feature i,
// handed to f(i, t): the K x M mat-vec every action ends with.  A thread walks its own row of BP; the vector is staged in
// LDS and the row entries are requested eight at a time from clamped addresses, so a row costs M / 8 memory round trips
// instead of M (element by element through generic pointers every term waited for its own two loads).
template <class Fn>
DEV void bm_rows_dot(const Blk &NOALIAS B, const BmWork &NOALIAS W, int K, int M, const double *vec, Fn f)
{
    const lptr_d lv = as_lds(B.pool);
    blk_sync(B);
    PAR(j, M) lv[j] = vec[j];
    blk_sync(B);
    const gptr_cd gBP = as_global(W.BP);
    const int ld = W.ld;
    PAR(i, K) {
        const gptr_cd bp = gBP + (size_t)i * ld;
        double t = 0;
        double nx[8];
#pragma unroll
        for (int c = 0; c < 8; c++) nx[c] = bp[c < M ? c : M - 1];
        for (int j0 = 0; j0 < M; j0 += 8) {
            double cur[8];
#pragma unroll
            for (int c = 0; c < 8; c++) cur[c] = nx[c];
            if (j0 + 8 < M) {
#pragma unroll
                for (int c = 0; c < 8; c++) { const int j = j0 + 8 + c; nx[c] = bp[j < M ? j : M - 1]; }
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                double v = cur[c];
                asm volatile("" : "+v"(v));
                if (j0 + c < M) t += v * lv[j0 + c];
            }
        }
        f(i, t);
    }
    blk_sync(B);
}

// BP[i][p] = sum_h x_i[h] w[h] Phi_p[h] / |x_i| for all features i and model columns p < M;
// also bb-style single columns through `only` (>= 0: only that column, written to W.bb).
// One 16-feature tile of bm_weighted_rows on the matrix cores, NCT column tiles of the staged block (compile-time: no
// guards around the matrix ops), EXT = the staged block carries the residual column (statistics wanted).
// xs: stride between consecutive samples of the lane's feature in the operand source -- 1 in the column-major design, K in
// the sample-major copy (F.Xt), where the 16 features of a tile are one 128-byte line per sample: a load then touches 4
// fully used lines instead of 16 quarter-used ones (the pass is bound by the CU's line requests, not by the matrix ops)
// The operand is addressed as a wave-uniform base (scalar registers) plus a 32-bit byte offset per lane (xo: the lane's
// feature, xs: bytes between consecutive samples), the form in which the memory pipe takes a load at full rate; the
// design is far below 4 GB.
// [hlo, Nr): the samples of this call (hlo a multiple of 4; Nr the end of the range, a multiple of 4): the staged block and
// the weights hold that range from index 0.  out / bbq_out come in with the values the chains start from (zero, or what an
// earlier range of the same pass left: the chain simply goes on) and leave with the results.
template <int NCT, int EXT>
DEV void bm_wr_tile(gptr_cc xb, unsigned xo, unsigned xs, lptr_d zb, lptr_d lw, int pitch, int Nu, int hlo, int Nr, int l4, double (&out)[NCT][4], double &bbq_out)
{
    typedef double bd4 __attribute__((ext_vector_type(4)));
    constexpr int RS = 8;                                      // steps per round (16 measured: no difference, config 3 261 vs 260 ms)
    bd4 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) acc[ct] = bd4{out[ct][0], out[ct][1], out[ct][2], out[ct][3]};
    double bbq = bbq_out;
    double an[RS];
#pragma unroll
    for (int u = 0; u < RS; u++) { const int h = hlo + 4 * u + l4; an[u] = *(gptr_cd)(xb + (xo + (unsigned)(h < Nu ? h : Nu - 1) * xs)); }
    double bn[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) bn[ct] = zb[ct * 16 * pitch];
    zb -= hlo; lw -= hlo;                                       // indexed by the absolute sample below
    for (int h0 = hlo; h0 < Nr; h0 += 4 * RS) {
        double ac[RS];
#pragma unroll
        for (int u = 0; u < RS; u++) ac[u] = (h0 + 4 * u + l4 < Nu) ? an[u] : 0.0;
#pragma unroll
        for (int u = 0; u < RS; u++) { const int h = h0 + 4 * RS + 4 * u + l4; an[u] = *(gptr_cd)(xb + (xo + (unsigned)(h < Nu ? h : Nu - 1) * xs)); }
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


// want_stats (device build): also bb_i = x_i' diag(w) x_i -> W.bb[i] and ze_i = x_i' e -> W.aroot[i] (what the
// full-stat pass needs per feature, NEmainEff.c:1745-1757), taken from the same pass over the design columns.
// Returns 1 when it did (matrix-core path), 0 when the caller has to compute them.
DEVNI int bm_weighted_rows(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, int M, bool want_stats = false, long long *phx = nullptr)
{
    (void)phx;
    const int N = F.N, ld = W.ld;
    (void)want_stats;
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
        // When the pool cannot hold all the columns of the pass over all the samples (half a pool: two fits per CU) the
        // SAMPLES are cut into ranges rather than the columns into blocks: the design -- what the pass is bound by -- is
        // still streamed once, the chains of a feature tile are parked in BP between two ranges (raw, exactly) and go on
        // where they stopped.  nseg ranges of Hs samples.
        const int want16 = (Mu + ws + 15) & ~15;
        int nseg = 1, Hs = Nr;
        if (want16 > pcm && want16 <= 16 * MAXCT && Mu + 8 <= ldu) {
            while (nseg < 8 && want16 * (Hs + 1) + (ws ? Hs : 0) > uni(B.pool_n)) { nseg++; Hs = (((Nr + nseg - 1) / nseg) + 3) & ~3; }
            if (want16 * (Hs + 1) + (ws ? Hs : 0) > uni(B.pool_n)) { nseg = 1; Hs = Nr; }
            else pcm = want16;
        }
        const int pitch_s = nseg > 1 ? Hs + 1 : pitch;
        if (pcm >= 16) {
            const lptr_d Z = as_lds(uni_ptr(B.pool));          // [column][pitch]
            const gptr_cd gX = as_global(uni_ptr(F.X)), gw = as_global(uni_ptr(W.w)), ge = as_global(uni_ptr(W.e));
            const gptr_cd gXt = as_global(uni_ptr(F.Xt));              // sample-major copy of the design (CV contexts), or null
            const gptr_cd gsc = as_global(uni_ptr(F.scale));
            const gptr_d gBP = as_global_rw(uni_ptr(W.BP)), gbb = as_global_rw(uni_ptr(W.bb)), gze = as_global_rw(uni_ptr(W.aroot));
            const int lane = B.lane, wave = uni(B.wave), nwave = uni(B.nwave), tid = B.tid, nthr = uni(B.nthr);
            const int l15 = lane & 15, l4 = lane >> 4;
            for (int p0 = 0; p0 < Mu; p0 += 0) {
                const int ext = (ws && p0 == 0) ? 1 : 0;
                const int pn = Mu - p0 < pcm - ext ? Mu - p0 : pcm - ext;
                const int pn16 = (pn + ext + 15) & ~15, nct = pn16 >> 4;
                const lptr_d lw = Z + pcm * pitch_s;                   // the weights, zero beyond the last sample
              for (int seg = 0; seg < nseg; seg++) {
                const int hlo = seg * Hs, hhi = hlo + Hs < Nr ? hlo + Hs : Nr;     // this range of samples (all of them when nseg == 1)
                const bool first = seg == 0, last = seg == nseg - 1;
                blk_sync(B);
                PHX_BEGIN(t_st);
                // a wave per staged column: the column's feature id and norm are wave-uniform, the design column is read in
                // whole lines (element by element every entry paid the chain used[] -> scale[] -> X and an integer division)
                for (int pc = wave; pc < pn16; pc += nwave) {
                    const int pm = p0 + pc;
                    const bool model = pc < pn, resid = ext && pc == pn;
                    const int u = (model && pm >= 1) ? uni(W.used[pm - 1]) : 0;
                    const bool dvz = W.phi_div != 0;
                    const double sc = dvz ? gsc[u] : F.rscale[u];
                    const gptr_cd x = gX + (size_t)u * Nu;
                    for (int hl = lane; hl < pitch_s; hl += 64) {
                        const int h = hlo + hl, hc = h < Nu ? h : Nu - 1;
                        double v = 0.0;
                        if (h < Nu && h < hhi) {
                            if (model) { const double xv = x[hc]; v = gw[hc] * (pm == 0 ? 1.0 : (dvz ? xv / sc : xv * sc)); }
                            else if (resid) v = ge[hc];
                        }
                        Z[pc * pitch_s + hl] = v;
                    }
                }
                if (ext) for (int hl = tid; hl < hhi - hlo; hl += nthr) lw[hl] = hlo + hl < Nu ? gw[hlo + hl] : 0.0;
                blk_sync(B);
                PHX_END(t_st, PH_HBUILD);
                PHX_BEGIN(t_mm);
                for (int ft = wave; ft * 16 < Ku; ft += nwave) {
                    const int il = ft * 16 + l15;
                    const int ilc = il < Ku ? il : Ku - 1;
                    const gptr_cc xb = (gptr_cc)(gXt ? gXt : gX);
                    const unsigned xo = gXt ? (unsigned)ilc * 8u : (unsigned)ilc * (unsigned)Nu * 8u;
                    const unsigned xs = gXt ? (unsigned)Ku * 8u : 8u;
                    const lptr_d zb = Z + l15 * pitch_s + l4;
                    double acc[MAXCT][4];
                    double bbq = 0;
                    // where the chains of this tile are parked between two ranges: the BP entries themselves (raw sums), the
                    // residual column in aroot, a lane's share of x' diag(w) x in the last four columns of the feature's row
#pragma unroll
                    for (int ct = 0; ct < MAXCT; ct++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            acc[ct][r] = 0.0;
                            if (!first && ct < nct) {
                                const int i = ft * 16 + l4 + 4 * r, col = ct * 16 + l15;
                                if (i < Ku) { if (col < pn) acc[ct][r] = gBP[(size_t)i * ldu + p0 + col]; else if (ext && col == pn) acc[ct][r] = gze[i]; }
                            }
                        }
                    if (!first && ext && il < Ku) bbq = gBP[(size_t)il * ldu + ldu - 4 + l4];
#define BM_WR_CASE(n)                                                                                                   \
                    case n: {                                                                                           \
                        double o[n][4];                                                                                 \
                        _Pragma("unroll") for (int ct = 0; ct < n; ct++) _Pragma("unroll") for (int r = 0; r < 4; r++) o[ct][r] = acc[ct][r]; \
                        if (ext) bm_wr_tile<n, 1>(xb, xo, xs, zb, lw, pitch_s, Nu, hlo, hhi, l4, o, bbq);               \
                        else bm_wr_tile<n, 0>(xb, xo, xs, zb, lw, pitch_s, Nu, hlo, hhi, l4, o, bbq);                   \
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
                                if (col < pn) gBP[(size_t)i * ldu + p0 + col] = last ? acc[ct][r] / gsc[i] : acc[ct][r];
                                else if (ext && col == pn) gze[i] = acc[ct][r];
                            }
                        }
                    }
                    if (ext) {
                        if (!last) { if (il < Ku) gBP[(size_t)il * ldu + ldu - 4 + l4] = bbq; }
                        else {                                 // the four sample groups of a feature sit 16 lanes apart
                            bbq += __shfl_xor(bbq, 16, 64); bbq += __shfl_xor(bbq, 32, 64);
                            if (l4 == 0 && il < Ku) gbb[il] = bbq;
                        }
                    }
                }
                PHX_END(t_mm, PH_MATVEC);
              }
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
    blk_sync(B);
    return 0;
}


// S_in = bb_i / |x_i|^2 - BP_i' Sigma BP_i and Q_in = ze_i / |x_i| for every feature (NEmainEff.c:1732-1762), the
// quadratic forms on the FP64 matrix cores: a wave owns 16 features; T = BP_tile * Sigma in 16 x 16 tiles (A operand:
// lane l holds BP[feature l & 15][k0 + (l >> 4)], B operand Sigma[k0 + (l >> 4)][16 ct + (l & 15)], 64 columns of Sigma
// per round), folded with BP on the fly: D register r of lane l is T[feature (l >> 4) + 4 r][column l & 15], multiplied
// by the same BP entry and summed over the 16 lanes of a row group.
DEVNI void bm_quad_features(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, int M)
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

// gradient entries 1 .. M-1 and the Hessian Phi' diag(w) Phi + diag(A) of one Newton step (NEmainEff.c:1890-1925)
// The same with the model columns staged in LDS first (Phi_p sample-contiguous, odd pitch, zero beyond the last sample; the
// weights behind them): one coalesced sweep of the M - 1 design columns by the whole workgroup, then the gradient sums and
// the chains of matrix ops read LDS.  Straight from the design a tile pair is a chain of N / 4 dependent matrix ops whose
// operands arrive eight steps at a time at 1 - 2 us a round (three waves busy, 38 us per call); from LDS it is bound by
// the chain itself.  Same products, same association ((phi_j * w) * phi_k), same order over the samples: same bits.
DEV bool bm_grad_hessian_lds(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int M, int N)
{
    typedef double bd4 __attribute__((ext_vector_type(4)));
    const int Mm = uni(M) - 1, Nu = uni(N), Nr = (Nu + 3) & ~3, pitch = Nr + 1, ld = uni(W.ld);
    if (Mm < 1 || Nu > 8 * 64 || Mm * pitch + Nr > uni(B.pool_n)) return false;
    const lptr_d Z = as_lds(uni_ptr(B.pool)), lw = Z + Mm * pitch;
    const gptr_cd gX = as_global(uni_ptr(F.X)), ge = as_global(uni_ptr(W.e)), gw = as_global(uni_ptr(W.w));
    const int lane = B.lane, wave = uni(B.wave), nwave = uni(B.nwave), tid = B.tid, nthr = uni(B.nthr);
    const bool dv = W.phi_div != 0;
    blk_sync(B);
    for (int c = wave; c < Mm; c += nwave) {                     // a wave per column: coalesced 512-byte loads
        const int u = uni(W.used[c]);
        const double sc = dv ? F.scale[u] : F.rscale[u];
        const gptr_cd x = gX + (size_t)u * Nu;
        for (int h = lane; h < pitch; h += 64) {
            const double v = x[h < Nu ? h : Nu - 1];
            Z[c * pitch + h] = h < Nu ? (dv ? v / sc : v * sc) : 0.0;
        }
    }
    for (int h = tid; h < Nr; h += nthr) lw[h] = h < Nu ? gw[h] : 0.0;
    blk_sync(B);
    {   // gradient entries and first row / column of the Hessian: a lane's sums over its samples in order, then the wave tree
        double er[8], wr[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int h = lane + 64 * k; er[k] = h < Nu ? ge[h] : 0.0; wr[k] = h < Nu ? lw[h < Nr ? h : 0] : 0.0; }
        for (int j = 1 + wave; j <= Mm; j += nwave) {
            double ga = 0, ha = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int h = lane + 64 * k;
                if (h < Nu) { const double ph = Z[(j - 1) * pitch + h]; ga += er[k] * ph; ha += wr[k] * ph; }
            }
            ga = wave_sum(ga); ha = wave_sum(ha);
            if (lane == 0) { W.g[j] = ga - W.A[j - 1] * W.mu[j]; W.H[j] = ha; W.H[(size_t)j * ld] = ha; }
        }
    }
    const int l15 = lane & 15, l4 = lane >> 4;
    if (Mm >= 8) {                                               // 16 x 16 tiles, lower-triangle tile pairs dealt to the waves
        const int nT = (Mm + 15) >> 4;
        for (int q = wave; q < nT * (nT + 1) / 2; q += nwave) {
            int tj = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
            while ((tj + 1) * (tj + 2) / 2 <= q) tj++;
            while (tj * (tj + 1) / 2 > q) tj--;
            const int tk = q - tj * (tj + 1) / 2;
            const int ja = tj * 16 + l15, kb = tk * 16 + l15;
            const lptr_d za = Z + (ja < Mm ? ja : Mm - 1) * pitch + l4, zb = Z + (kb < Mm ? kb : Mm - 1) * pitch + l4;
            const bool ona = ja < Mm, onb = kb < Mm;
            bd4 acc = bd4{0, 0, 0, 0};
            double pa = za[0], pb = zb[0], pw = lw[l4];
            for (int h0 = 0; h0 < Nr; h0 += 4) {
                const double ca = pa, cb = pb, cw = pw;
                const int hn = h0 + 4 < Nr ? h0 + 4 : h0;         // next step's operands behind this step's matrix op
                pa = za[hn]; pb = zb[hn]; pw = lw[hn + l4];
                const double a = ona ? ca * cw : 0.0, b = onb ? cb : 0.0;   // samples beyond N are zero in Z
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
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
    } else {                                                     // few columns: one wavefront per (j, k) pair, the reference's triple product
        const int np = Mm * (Mm + 1) / 2;
        for (int q = wave; q < np; q += nwave) {
            int j = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
            while ((j + 1) * (j + 2) / 2 <= q) j++;
            while (j * (j + 1) / 2 > q) j--;
            const int k = q - j * (j + 1) / 2;                   // 0 <= k <= j < Mm
            double a = 0;
            for (int h = lane; h < Nu; h += 64) a += Z[j * pitch + h] * lw[h] * Z[k * pitch + h];
            a = wave_sum(a);
            if (lane == 0) {
                if (j == k) a += W.A[k];
                W.H[(size_t)(k + 1) * ld + j + 1] = a; W.H[(size_t)(j + 1) * ld + k + 1] = a;
            }
        }
    }
    blk_sync(B);
    return true;
}

DEV void bm_grad_hessian(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int M, int N)
{
    const int ld = W.ld;
    if (bm_grad_hessian_lds(B, F, W, M, N)) return;
        if (N <= 8 * 64) {
            // a lane's share of e and w in registers; four model columns per trip, their 4 x 8 design loads issued together from
            // clamped addresses (a column at a time, element by element through BM_PHI, every term paid its own chain of
            // dependent loads: used[] -> scale[] -> X); a lane's sums run over its samples in order as before
            const gptr_cd gX = as_global(uni_ptr(F.X)), ge = as_global(uni_ptr(W.e)), gw = as_global(uni_ptr(W.w));
            const int lane = B.lane, wave = uni(B.wave), nwave = uni(B.nwave), Nu = uni(N), Mu = uni(M);
            const bool dv = W.phi_div != 0;
            double er[8], wr[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const int h = lane + 64 * k; er[k] = h < Nu ? ge[h] : 0.0; wr[k] = h < Nu ? gw[h] : 0.0; }
            for (int j0 = 1 + wave; j0 < Mu; j0 += 4 * nwave) {
                double xr[4][8], sc[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int j = j0 + c * nwave, u = uni(W.used[(j < Mu ? j : j0) - 1]);
                    sc[c] = dv ? F.scale[u] : F.rscale[u];
                    const gptr_cd x = gX + (size_t)u * Nu;
#pragma unroll
                    for (int k = 0; k < 8; k++) { const int h = lane + 64 * k; xr[c][k] = x[h < Nu ? h : Nu - 1]; }
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int j = j0 + c * nwave;
                    if (j < Mu) {
                        double ga = 0, ha = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            double v = xr[c][k];
                            asm volatile("" : "+v"(v));
                            if (lane + 64 * k < Nu) { const double ph = dv ? v / sc[c] : v * sc[c]; ga += er[k] * ph; ha += wr[k] * ph; }
                        }
                        ga = wave_sum(ga); ha = wave_sum(ha);
                        if (lane == 0) { W.g[j] = ga - W.A[j - 1] * W.mu[j]; W.H[j] = ha; W.H[(size_t)j * ld] = ha; }
                    }
                }
            }
        } else
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
}

// bb_i = x_i' diag(w) x_i and ze_i = x_i' e for every feature, when bm_weighted_rows did not deliver them
DEV void bm_feature_stats(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, int N, int have_stats)
{
    if (!have_stats) {
        for (int i = B.wave; i < K; i += B.nwave) {
            const double *x = F.X + (size_t)i * N;
            double bbq = 0, ze = 0;
            for (int h = B.lane; h < N; h += 64) { const double xv = x[h]; bbq += W.w[h] * (xv * xv); ze += xv * W.e[h]; }
            bbq = wave_sum(bbq); ze = wave_sum(ze);
            if (B.lane == 0) { W.bb[i] = bbq; W.aroot[i] = ze; }
        }
    }
}

// bb[i] = x_i' (w .* phi) / |x_i| for all features and tmp[p] = Phi_p' (w .* phi) for the model columns (bm_add)
DEV void bm_add_products(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const BmWork &NOALIAS W, int K, int M, int N)
{
    // eight features per wave and reduction tree (wave_sum8 pairs lanes exactly like wave_sum, so the sums are the same
    // bits).  Up to 512 samples a lane's share of w .* phi lives in registers and the 8 x 8 design loads of a trip are
    // issued together from clamped addresses before the first product (a lane's sum still runs over its samples in
    // order): a trip is one memory round trip instead of one per feature and sample group.
    if (N <= 8 * 64) {
        const gptr_cd gX = as_global(uni_ptr(F.X)), gbp = as_global(uni_ptr(W.bphi));
        const int lane = B.lane, wave = uni(B.wave), nwave = uni(B.nwave);
        N = uni(N); K = uni(K);
        double bp[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int h = lane + 64 * k; bp[k] = h < N ? gbp[h] : 0.0; }
        for (int i0 = wave * 8; i0 < K; i0 += nwave * 8) {
            double xr[8][8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const gptr_cd x = gX + (size_t)(i0 + c < K ? i0 + c : K - 1) * N;
#pragma unroll
                for (int k = 0; k < 8; k++) { const int h = lane + 64 * k; xr[c][k] = x[h < N ? h : N - 1]; }
            }
            double a[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                double t = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) { double v = xr[c][k]; asm volatile("" : "+v"(v)); if (lane + 64 * k < N) t += v * bp[k]; }
                a[c] = t;
            }
            wave_sum8(a, lane);
            const int i = i0 + (lane >> 3);
            if ((lane & 7) == 0 && i < K) W.bb[i] = a[0] / F.scale[i];
        }
    } else
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
    if (N <= 8 * 64) {                                          // the same for the model columns, four per trip
        const gptr_cd gX = as_global(uni_ptr(F.X)), gbp = as_global(uni_ptr(W.bphi));
        const int lane = B.lane, wave = uni(B.wave), nwave = uni(B.nwave), Mu = uni(M);
        const bool dv = W.phi_div != 0;
        double bp[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int h = lane + 64 * k; bp[k] = h < N ? gbp[h] : 0.0; }
        for (int p0 = wave; p0 < Mu; p0 += 4 * nwave) {
            double xr[4][8], sc[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int pp = p0 + c * nwave, pc = pp < Mu ? pp : p0;
                const int u = pc >= 1 ? uni(W.used[pc - 1]) : 0;
                sc[c] = dv ? F.scale[u] : F.rscale[u];
                const gptr_cd x = gX + (size_t)u * N;
#pragma unroll
                for (int k = 0; k < 8; k++) { const int h = lane + 64 * k; xr[c][k] = x[h < N ? h : N - 1]; }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int pp = p0 + c * nwave;
                if (pp < Mu) {
                    double a = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        double v = xr[c][k];
                        asm volatile("" : "+v"(v));
                        if (lane + 64 * k < N) { const double ph = pp == 0 ? 1.0 : (dv ? v / sc[c] : v * sc[c]); a += ph * bp[k]; }
                    }
                    a = wave_sum(a);
                    if (lane == 0) W.tmp[pp] = a;
                }
            }
        }
        return;
    }
    for (int p = B.wave; p < M; p += B.nwave) {
        double a = 0;
        for (int h = B.lane; h < N; h += 64) a += BM_PHI(p, h) * W.bphi[h];
        a = wave_sum(a);
        if (B.lane == 0) W.tmp[p] = a;
    }
}
