// gm_dev.h -- the MI355X mapping of the phases of a Gaussian fit (gm_fit.h includes it after its prelude): the
// full-stat pass on the FP64 matrix cores, the job board of the shared phases, the Gram-row sweeps, the in-place
// Sigma updates, on-demand Gram rows, runs of adds, the blocked inverse, the Hessian gather.  gfx950 only.
#pragma once

#define GM_TRACE(...)

// Optional per-phase tick accumulation (diagnostic build -DPAREBEN_PHASE_TIMERS only; the ticks go
// to a buffer of their own and never into a result).
#if defined(PAREBEN_PHASE_TIMERS)
#define PH_BEGIN() long long ph_t0_ = (B.tid == 0) ? (long long)wall_clock64() : 0
#define PH_END(k) do { if (B.tid == 0) S.ph[k] += (long long)wall_clock64() - ph_t0_; } while (0)
#define PHX_BEGIN(v) long long v = (B.tid == 0) ? (long long)wall_clock64() : 0
#define PHX_END(v, k) do { if (B.tid == 0 && phx) phx[k] += (long long)wall_clock64() - v; } while (0)
#else
#define PH_BEGIN() do {} while (0)
#define PH_END(k) do {} while (0)
#define PHX_BEGIN(v) do {} while (0)
#define PHX_END(v, k) do {} while (0)
#endif
// finer ticks inside the register form of the inverse, on slots the fit kernels use for other things: only in the
// single-phase diagnostic build (tools/ubench/inverse_rate.py with DIAG_LIB=libpareben_hip_diagprof.so)
#if defined(PAREBEN_PHASE_TIMERS) && defined(PAREBEN_DIAG)
#define PHD_BEGIN(v) PHX_BEGIN(v)
#define PHD_END(v, k) PHX_END(v, k)
#else
#define PHD_BEGIN(v) do {} while (0)
#define PHD_END(v, k) do {} while (0)
#endif

// address-space-qualified views (global_load / ds_read instead of flat_load) for the hot loops
typedef const double __attribute__((address_space(1))) *gptr_cd;
typedef const char __attribute__((address_space(1))) *gptr_cc;
typedef const int __attribute__((address_space(1))) *gptr_ci;
typedef double __attribute__((address_space(1))) *gptr_d;
typedef int __attribute__((address_space(3))) *lptr_i;
typedef double __attribute__((address_space(3))) *lptr_d;
typedef double d4 __attribute__((ext_vector_type(4)));
DEV gptr_cd as_global(const double *p) { return (gptr_cd)p; }
DEV gptr_ci as_global(const int *p) { return (gptr_ci)p; }
DEV lptr_d as_lds(double *p) { return (lptr_d)p; }
DEV lptr_i as_lds(int *p) { return (lptr_i)p; }
DEV gptr_d as_global_rw(double *p) { return (gptr_d)p; }
// Arguments of non-inlined device functions arrive in VGPRs; these put wave-uniform values back
// into SGPRs so addresses become "scalar base + 32-bit lane offset" (fewer VGPRs, saddr loads).
DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T> DEV T *uni_ptr(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
#ifndef FS_NB
#define FS_NB 2            // 16-feature column blocks per wave: every Sigma panel read from LDS feeds FS_NB matrix ops
#endif
#define FS_TPP (16 / FS_NB) // row tiles of Sigma per pass (FS_NB x FS_TPP accumulator tiles per wave)
#ifndef FS_NWAVES
#define FS_NWAVES 8        // wavefronts per fit workgroup (set by the kernel file from FIT_THREADS)
#endif
#define FS_FT (16 * FS_NWAVES * FS_NB)   // features per tile: every wave owns FS_NB 16-feature column blocks of it
#define SQ_FT 128          // feature tile of the shared action mat-vecs (job board granularity)
#ifndef FS_APRE
#define FS_APRE 1           // tiles by which the LDS reads of the A operands run ahead of the matrix ops
#endif
#define FS_PPW (FS_TPP / FS_NWAVES)   // Sigma panels each wave stages per step
#define FS_MAX_M 2048      // largest active set the pass is laid out for (LDS: row offsets + mu); the host bounds every capacity by it
static_assert(FS_TPP % FS_NWAVES == 0 && 16 % FS_NB == 0, "full-stat tiling");

// The full-stat pass is ONE software pipeline over all (feature tile, pass, k-block) steps of a call:
// a cursor names the step; what a step needs from memory is requested one (Gram operands) or two
// (Sigma panels) steps ahead, across pass and tile boundaries, so memory latency is exposed once per call.
struct FsCur { int i0, pass, h, last; };          // feature tile origin, row-tile pass, k-block, last k-block of the pass
// Row tiles are cut into passes of at most FS_TPP; a pass visits the k-blocks 0 .. (its last tile), so the number of
// steps of a feature tile is the sum of the passes' end indices: smallest when the LATER passes are full and the
// first one takes the remainder (19 tiles: 3 + 8 + 8 -> 3 + 11 + 19 = 33 steps; 7 + 6 + 6 would be 39, 8 + 8 + 3: 43).
// `first` = tiles of pass 0.
DEV int fs_pass_begin(int pass, int first) { return pass == 0 ? 0 : first + (pass - 1) * FS_TPP; }
DEV int fs_pass_end(int pass, int first) { return first + pass * FS_TPP; }          // exclusive; the last pass ends at nJ
DEV void fs_advance(FsCur &c, int n_pass, int nJ, int first)
{
    c.h++;
    if (c.h > c.last) {
        c.h = 0;
        c.pass++;
        if (c.pass == n_pass) { c.pass = 0; c.i0 += FS_FT; }
        (void)nJ;
        c.last = fs_pass_end(c.pass, first) - 1;
    }
}

// One step = k-block h (16 rows of the active set) of one pass (<= FS_TPP row tiles of Sigma) of one
// feature tile.  T = Sigma * Bt on the FP64 matrix cores (v_mfma_f64_16x16x4_f64) in 16 x 16 tiles:
//   A (16 rows of Sigma x 4 k):  lane l holds A[row = l & 15][k = l >> 4]
//   B (4 k x 16 features):       lane l holds B[k = l >> 4][col = l & 15]
//   D register r of lane l:      T[row = (l >> 4) + 4 r][col = l & 15]
// Work split: every wave owns FS_NB 16-feature column blocks of the tile and ALL row tiles of the pass
// (FS_NB x FS_TPP accumulator tiles), so at every step all waves do the same number of matrix ops -- the
// triangular schedule below costs no balance -- and every A operand read feeds FS_NB matrix ops.
// A operand: the Sigma panels of the step ((tile J, k-block h) = 16 x 16) are identical for all waves, so
// they are staged once per workgroup through LDS, in operand order (a wave reads one contiguous 512 B line
// per operand); each Sigma element is fetched from memory once per workgroup and step.
// B operand: lane l of k-group s holds G[row 16 h + 4 s + (l >> 4)][feature 16 blk + (l & 15)] -- 16
// consecutive features of one Gram row per 16 lanes, i.e. whole 128-byte lines -- and a wave needs only its
// own column blocks, so the Gram block is not shared through LDS at all: each wave loads its operands for
// step g+1 straight into a register ring during step g (loff[p] = byte offset of Gram row p of the active
// set).  Rows >= M are zeroed when they are used; the Sigma entries of the ragged 16-block beyond the active
// set are multiplied by those zeros, so they must be finite: gm_fullstat clears that band before every pass
// (whatever an earlier fit of this workgroup left there -- possibly NaN -- never reaches a result).
// Sigma is symmetric: row tile J only visits k-blocks h <= J and counts h < J twice (the panel is
// doubled when it is staged; doubling is exact).
// When h == J the k-block on the diagonal IS tile J's own rows, and the rows a lane holds as B operand
// (4 s + l4) are exactly the rows of its accumulator registers (l4 + 4 r): the wave folds
// sum_j T[j][i] b_j[i] and sum_j b_j[i] mu_j for its features right there from registers and clears the tile.
//   * panel ring slot CUR (= g & 1) receives the loads of step g+2 (cursor c2); slot CUR^1 holds step g+1's
//     panels (requested during step g-1), which this step writes to the other LDS buffer.
// Uniform branches guard matrix ops, folds and LDS writes only -- never a load -- so the load/wait
// bookkeeping is the same on every path (counted s_waitcnt vmcnt(N); the function must not spill and
// nothing may be pending at loop entry, or the compiler puts a static vmcnt(0) inside the loop).
typedef const unsigned long long __attribute__((address_space(3))) *lptr_cull;
template <int CUR>
DEV void fs_step(gptr_cc Sig, gptr_cc G, lptr_cull loff, lptr_d lmu, lptr_d acur, lptr_d anxt,
                   int ld, const FsCur &c0, const FsCur &c1, const FsCur &c2, int first, int nJ, int M, int K, int wave, int lane,
                   double (&pa)[2][FS_PPW][4], double (&bvr)[2][FS_NB][4], d4 (&acc)[FS_NB][FS_TPP], double (&qsum)[FS_NB],
                   double (&msum)[FS_NB])
{
    constexpr int NX = CUR ^ 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    // ---- matrix ops of step g; rows beyond the active set contribute zero
    const int jb = fs_pass_begin(c0.pass, first), je = fs_pass_end(c0.pass, first);
    double bv[FS_NB][4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const bool live = c0.h * 16 + 4 * s + l4 < M;
#pragma unroll
        for (int nb = 0; nb < FS_NB; nb++) bv[nb][s] = live ? bvr[CUR][nb][s] : 0.0;
    }
    // The A operands of tile t+1 are read from LDS before the matrix ops of tile t are issued (FS_APRE tiles ahead
    // through a small register ring), so no LDS latency sits between two tiles' matrix ops; the reads are
    // unconditional (every slot of the buffer is addressable), only the matrix ops are guarded.
    double an[FS_APRE][4];
#pragma unroll
    for (int p = 0; p < FS_APRE; p++)
#pragma unroll
        for (int s = 0; s < 4; s++) an[p][s] = acur[(p * 4 + s) * 64 + lane];
#pragma unroll
    for (int t = 0; t < FS_TPP; t++) {
        const int J = jb + t;
        double ac[4];
#pragma unroll
        for (int s = 0; s < 4; s++) ac[s] = an[t % FS_APRE][s];
        if (t + FS_APRE < FS_TPP) {
#pragma unroll
            for (int s = 0; s < 4; s++) an[t % FS_APRE][s] = acur[((t + FS_APRE) * 4 + s) * 64 + lane];
        }
        // ---- requests, placed behind work that is already queued so that their address arithmetic and the LDS read of
        // the row offsets do not hold up the first matrix ops of the step: this wave's share of the Sigma panels of
        // step g+2 in front of tile 0 ...
        if (t == 0) {
            const unsigned lane_off = (unsigned)((l4 * ld + l15) * 8);
#pragma unroll
            for (int pi = 0; pi < FS_PPW; pi++) {
                const int t2 = wave + pi * FS_NWAVES, J2 = fs_pass_begin(c2.pass, first) + t2;
                const bool on = J2 < fs_pass_end(c2.pass, first) && c2.h <= J2;
                const unsigned o = on ? (unsigned)((c2.h * 16 * ld + J2 * 16) * 8) + lane_off : 0u;
                const unsigned st = on ? (unsigned)(4 * ld * 8) : 0u;
#pragma unroll
                for (int s = 0; s < 4; s++) pa[CUR][pi][s] = *(gptr_cd)(Sig + (o + s * st));
            }
        }
        if (t == 1) {                                         // ... and its own Gram operands of step g+1 behind tile 0's matrix ops
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const unsigned long long ro = loff[c1.h * 16 + 4 * s + l4];
#pragma unroll
                for (int nb = 0; nb < FS_NB; nb++) {
                    const int i = c1.i0 + 16 * (wave + nb * FS_NWAVES) + l15;
                    bvr[NX][nb][s] = *(gptr_cd)(G + (ro + (unsigned)((i < K ? i : K - 1) * 8)));
                }
            }
        }
        if (J < je && c0.h <= J) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double a = ac[s];
#pragma unroll
                for (int nb = 0; nb < FS_NB; nb++) acc[nb][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[nb][s], acc[nb][t], 0, 0, 0);
            }
            if (c0.h == J) {                                  // tile J is complete: fold and clear
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double mj = lmu[c0.h * 16 + l4 + 4 * r];
#pragma unroll
                    for (int nb = 0; nb < FS_NB; nb++) {
                        qsum[nb] += acc[nb][t][r] * bv[nb][r];
                        msum[nb] += bv[nb][r] * mj;
                    }
                }
#pragma unroll
                for (int nb = 0; nb < FS_NB; nb++) acc[nb][t] = d4{0, 0, 0, 0};
            }
        }
    }
    // ---- step g+1's Sigma panels (requested during step g-1) -> the other LDS buffer
#pragma unroll
    for (int pi = 0; pi < FS_PPW; pi++) {
        const int t = wave + pi * FS_NWAVES, J = fs_pass_begin(c1.pass, first) + t;
        const double w = c1.h < J ? 2.0 : 1.0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            double v = pa[NX][pi][s] * w;
            asm volatile("" : "+v"(v));                        // consume the load on every path (see above)
            if (J < fs_pass_end(c1.pass, first) && c1.h <= J) anxt[(t * 4 + s) * 64 + lane] = v;
        }
    }
}


// S_in[i] = beta - beta^2 b_i' Sigma b_i,  Q_in[i] = beta (bt_i - b_i' mu)  for the features of tiles
// tile0 .. tile1-1 (FS_FT features each), b_i = G[used, i].  MainEff.c:1291-1319.  This is the K*M^2 contraction
// that dominates the run time, here T = Sigma * B' on the FP64 matrix cores as one software
// pipeline over all (feature tile, row-tile pass, k-block) steps: see fs_step for the work split (every wave owns
// FS_NB column blocks and all row tiles of a pass), the operand paths (Sigma panels staged through LDS once per
// workgroup and step, Gram operands straight from memory into a register ring) and the symmetric schedule.
DEVNI void gm_fullstat_features(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, int M,
                              double beta, int tile0, int tile1)
{
    // see fs_step.  B lives in memory (reference argument of a non-inlined function): take register
    // copies once, or every use in the loop becomes a flat load followed by s_waitcnt vmcnt(0); results
    // leave through global-address-space pointers for the same reason.
    const int lane = B.lane, wave = uni(B.wave), tid = B.tid, nthr = uni(B.nthr);
    const gptr_cc Sig = (gptr_cc)as_global(uni_ptr(W.Sig));
    const gptr_cc G = (gptr_cc)as_global(uni_ptr(F.G));
    const gptr_d gSin = as_global_rw(uni_ptr(W.Sin)), gQin = as_global_rw(uni_ptr(W.Qin));
    const gptr_cd gbt = as_global(uni_ptr(W.bt));
    double *pool = uni_ptr(B.pool);
    const lptr_d la = as_lds(pool);                                         // 2 x FS_TPP x 256 staged Sigma panels
    unsigned long long *loff_w = (unsigned long long *)(pool + 2 * FS_TPP * 256);          // Gram row byte offsets, M <= FS_MAX_M
    const lptr_cull loff = (lptr_cull)loff_w;
    const lptr_d lmu = as_lds(pool + 2 * FS_TPP * 256 + FS_MAX_M);          // mu, zero-padded to a k-block
    K = uni(K); M = uni(M);
    const int ld = uni(W.ld);
    const int nJ = (M + 15) >> 4;
    const int n_pass = (nJ + FS_TPP - 1) / FS_TPP;
    const int first = nJ - (n_pass - 1) * FS_TPP;     // row tiles of pass 0; the later passes are full (see fs_pass_end)
    const int n_ft = uni(tile1) - uni(tile0);                               // feature tiles tile0 .. tile1-1 of the call
    const int i_begin = uni(tile0) * FS_FT;
    int steps_per_tile = 0;
    for (int p = 0; p < n_pass; p++) steps_per_tile += fs_pass_end(p, first);
    const int total = n_ft * steps_per_tile;
    const int l15 = lane & 15, l4 = lane >> 4;
    __syncthreads();
    for (int p = tid; p < nJ * 16; p += nthr) {
        loff_w[p] = (unsigned long long)W.rowid[p < M ? p : 0] * (unsigned long long)K * 8ull;
        lmu[p] = p < M ? W.mu[p] : 0.0;
    }
    __syncthreads();
    FsCur c0, c1, c2;
    c0.i0 = i_begin; c0.pass = 0; c0.h = 0; c0.last = first - 1;
    c1 = c0; fs_advance(c1, n_pass, nJ, first);
    c2 = c1; fs_advance(c2, n_pass, nJ, first);
    d4 acc[FS_NB][FS_TPP];
#pragma unroll
    for (int nb = 0; nb < FS_NB; nb++)
#pragma unroll
        for (int t = 0; t < FS_TPP; t++) acc[nb][t] = d4{0, 0, 0, 0};
    double pa[2][FS_PPW][4], bvr[2][FS_NB][4];
    double qsum[FS_NB], msum[FS_NB];
#pragma unroll
    for (int nb = 0; nb < FS_NB; nb++) { qsum[nb] = 0; msum[nb] = 0; }
    {   // pipeline fill: step 0's Gram operands into ring slot 0, its Sigma panels straight to LDS buffer 0,
        // step 1's panels into ring slot 1
        const unsigned lane_off = (unsigned)((l4 * ld + l15) * 8);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const unsigned long long ro = loff[4 * s + l4];
#pragma unroll
            for (int nb = 0; nb < FS_NB; nb++) {
                const int i = i_begin + 16 * (wave + nb * FS_NWAVES) + l15;
                bvr[0][nb][s] = *(gptr_cd)(G + (ro + (unsigned)((i < K ? i : K - 1) * 8)));
            }
        }
#pragma unroll
        for (int pi = 0; pi < FS_PPW; pi++) {
            const int t = wave + pi * FS_NWAVES;
            const int J1 = fs_pass_begin(c1.pass, first) + t;
            const bool on0 = t < first, on1 = J1 < fs_pass_end(c1.pass, first) && c1.h <= J1;
            const unsigned o0 = on0 ? (unsigned)(t * 16 * 8) + lane_off : 0u, st0 = on0 ? (unsigned)(4 * ld * 8) : 0u;
            const unsigned o1 = on1 ? (unsigned)((c1.h * 16 * ld + J1 * 16) * 8) + lane_off : 0u, st1 = on1 ? (unsigned)(4 * ld * 8) : 0u;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const double v0 = *(gptr_cd)(Sig + (o0 + s * st0));
                pa[1][pi][s] = *(gptr_cd)(Sig + (o1 + s * st1));
                if (on0) la[(t * 4 + s) * 64 + lane] = v0 * (0 < t ? 2.0 : 1.0);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing pending at loop entry; the compiler tracks this form
    __syncthreads();
    // After the last step of a feature tile every wave holds the sums of its features, spread over
    // the four row groups of the accumulator layout: two xor-shuffles, then lanes 0..15 write them.
#define FS_FINISH_TILE(cc)                                                                                           \
        if ((cc).h == (cc).last && (cc).pass == n_pass - 1) {                                                        \
            _Pragma("unroll") for (int nb = 0; nb < FS_NB; nb++) {                                                   \
                double q = qsum[nb], m = msum[nb];                                                                   \
                q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);                                              \
                m += __shfl_xor(m, 16, 64); m += __shfl_xor(m, 32, 64);                                              \
                const int i = (cc).i0 + 16 * (wave + nb * FS_NWAVES) + l15;                                          \
                if (lane < 16 && i < K) { gSin[i] = beta - beta * q * beta; gQin[i] = beta * (gbt[i] - m); }         \
                qsum[nb] = 0; msum[nb] = 0;                                                                          \
            }                                                                                                        \
        }
#define FS_STEP(CURSLOT, gg)                                                                                         \
        {                                                                                                            \
            const int cb = (gg) & 1, nb_ = cb ^ 1;                                                                   \
            fs_step<CURSLOT>(Sig, G, loff, lmu, la + cb * (FS_TPP * 256), la + nb_ * (FS_TPP * 256), ld, c0, c1, c2, \
                               first, nJ, M, K, wave, lane, pa, bvr, acc, qsum, msum);                                 \
            FS_FINISH_TILE(c0)                                                                                       \
            __syncthreads();                                                                                         \
            c0 = c1; c1 = c2; fs_advance(c2, n_pass, nJ, first);                                                       \
        }
    // whole pairs of steps without any skip path inside the loop (a skipped step would leave its
    // predecessor's prefetch pending on the back edge: static s_waitcnt vmcnt(0)); then the odd tail
    int g = 0;
    for (; g + 1 < total; g += 2) {
        FS_STEP(0, g)
        FS_STEP(1, g + 1)
    }
    if (g < total) FS_STEP(0, g)
#undef FS_STEP
#undef FS_FINISH_TILE
    blk_sync(B);
}

// ---- shared phases ------------------------------------------------------------------------------
// A fit is one workgroup, and the heaviest fits of a grid take tens of times the median; once the
// work queue is drained the workgroups that are out of fits would sit idle while those finish.  The
// full-stat pass (half of a heavy fit's time) and the K x M mat-vec of every action are independent
// per feature, so in that phase of the launch an owner OPENS each of them (FsJob): idle workgroups
// claim chunks of feature tiles (FS_FT or SQ_FT features each) by compare-and-swap on (epoch, next tile), run the same code on
// the owner's state in HBM (Sigma / mu / row ids, or the action's vector) and write S_in / Q_in for
// their features; the owner works on its own job too and waits for the chunk count.  Results do not
// depend on who computed a feature (same code, same order).
// Visibility follows the guide's hand-off recipe both ways: stores drained by every wave, barrier,
// one agent-scope release, then a relaxed atomic; consumers read the atomic relaxed, then one
// agent-scope acquire + s_waitcnt vmcnt(0) + barrier before plain loads.  Nobody waits while holding
// a chunk, so every wait ends; the owner's wait is bounded anyway and flags the fit if it expires.
#define FS_CHUNK (4 / FS_NB) // tiles per claim, full-stat pass (512 features)
#define SQ_CHUNK 8         // tiles per claim, action mat-vec (1024 features = one pair per thread)
#define AT_LOAD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define AT_STORE(p, v) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define AT_ADD(p, v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
DEV bool fs_epoch_open(unsigned long long w) { return (w >> 32) & 1; }
// tiles per claim: a full-stat tile of a large active set is long enough (0.1 ms and up) to be claimed on its own,
// which is what lets a whole idle GPU work on the one heavy fit left on it
DEV int job_chunk(int kind, int M) { return kind != JOB_FULLSTAT ? SQ_CHUNK : FS_CHUNK; }
// thread 0 only: first tile of the claimed chunk, or -1 when the job is closed / fully handed out
DEV int fs_claim(FsJob *job, int n_tiles, int chunk)
{
    for (;;) {
        unsigned long long w = AT_LOAD(&job->word);
        if (!fs_epoch_open(w) || (int)(unsigned)w >= n_tiles) return -1;
        if (__hip_atomic_compare_exchange_strong(&job->word, &w, w + chunk, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            return (int)(unsigned)w;
    }
}
// Owner side.  Returns false when sharing is off / not worth it (the caller then does the whole phase);
// otherwise opens the job, runs work(tile0, tile1) on the chunks it claims itself, waits for the rest.
template <class Work>
DEV bool job_share(const Blk &NOALIAS B, GmScalars &NOALIAS S, int kind, int M, int n_tiles, double beta, int mode, int rid, int aux, double c1,
                   double c2, Work work)
{
    const FsShare *sh = S.share;
    const int chunk = job_chunk(kind, M);
    if (!(sh && sh->jobs) || n_tiles < 4 * chunk) return false;
    __syncthreads();
    // somebody may be free to help: always in the tail; from the start when the launch is small or the fit is a
    // heavy one (large active set: the few fits that decide the step time once the work is spread over GPUs)
    if (B.tid == 0) B.ired[0] = sh->early || M >= sh->heavy_m || AT_LOAD(sh->queue) >= sh->n_units;
    __syncthreads();
    const bool open = B.ired[0] != 0;
    __syncthreads();
    if (!open) return false;
    FsJob *job = sh->jobs + sh->self;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // what the phase reads: every wave drains its stores
    __syncthreads();
    if (B.tid == 0) {
        AT_STORE(&job->done, 0); AT_STORE(&job->fold, S.fold); AT_STORE(&job->M, M); AT_STORE(&job->n_tiles, n_tiles);
        AT_STORE(&job->kind, kind); AT_STORE(&job->mode, mode); AT_STORE(&job->rid, rid); AT_STORE(&job->pad, aux);
        __hip_atomic_store(&job->beta, beta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&job->c1, c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&job->c2, c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long w = AT_LOAD(&job->word);
        AT_STORE(&job->word, ((w >> 32) + 1) << 32);           // odd epoch, next tile 0: open
    }
    __syncthreads();
    int mine = 0;
    for (;;) {
        if (B.tid == 0) B.ired[0] = fs_claim(job, n_tiles, chunk);
        __syncthreads();
        const int c = B.ired[0];
        __syncthreads();
        if (c < 0) break;
        work(c, c + chunk < n_tiles ? c + chunk : n_tiles);
        mine++;
    }
    if (B.tid == 0) {
        const int total = (n_tiles + chunk - 1) / chunk;
        AT_ADD(&job->done, mine);
        long spins = 0;
#ifdef PAREBEN_PHASE_TIMERS
        const long long tw0 = (long long)wall_clock64();
#endif
        while (AT_LOAD(&job->done) < total && spins < 200000000L) { __builtin_amdgcn_s_sleep(4); spins++; }
#ifdef PAREBEN_PHASE_TIMERS
        const int fsj = kind == JOB_FULLSTAT;
        S.ph[fsj ? PH_FS_MINE : PH_SQ_MINE] += mine; S.ph[fsj ? PH_FS_CHUNKS : PH_SQ_CHUNKS] += total;
        S.ph[fsj ? PH_FS_WAIT : PH_SQ_WAIT] += (long long)wall_clock64() - tw0;
#endif
        B.ired[0] = AT_LOAD(&job->done) >= total;
        const unsigned long long w = AT_LOAD(&job->word);
        AT_STORE(&job->word, ((w >> 32) + 1) << 32);           // even epoch: closed
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the helpers' S_in / Q_in
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!B.ired[0]) S.status |= ST_ABORT;
    __syncthreads();
    return true;
}

// the whole pass: shared with idle workgroups when the job board is open
DEVNI void gm_fullstat_pass(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, int M, double beta, GmScalars &NOALIAS S)
{
    const int n_tiles = (K + FS_FT - 1) / FS_FT;
    if (M >= 48 && job_share(B, S, JOB_FULLSTAT, M, n_tiles, beta, 0, -1, -1, 0.0, 0.0,
                             [&](int t0, int t1) { gm_fullstat_features(B, F, W, K, M, beta, t0, t1); })) return;
    gm_fullstat_features(B, F, W, K, M, beta, 0, n_tiles);
}
// the band between the active block and the next multiple of 16 (see fs_step): exact zeros for the matrix-core pass
DEV void gm_fs_pad(const Blk &NOALIAS B, const GmWork &NOALIAS W, int M)
{
    const int ld = W.ld, Mp = ((M + 15) >> 4) << 4, pad = Mp - M;
    for (int e = B.tid; e < Mp * pad; e += B.nthr) {
        const int j = e / pad, i = M + e - j * pad;
        W.Sig[(size_t)j * ld + i] = 0.0;
        W.Sig[(size_t)i * ld + j] = 0.0;
    }
}
// matrix-core work of a pass: per 16-feature block the triangle of (row tile J, k-block h <= J) tile products
DEV void gm_fs_count(const Blk &NOALIAS B, GmScalars &NOALIAS S, int K, int M)
{
    CNT(const int64_t nJ = (M + 15) >> 4; c.mfma_tiles += (int64_t)((K + FS_FT - 1) / FS_FT) * (FS_FT / 16) * (nJ * (nJ + 1) / 2));
}
#if defined(PAREBEN_PHASE_TIMERS)
#define GM_FS_CLOCK_BEGIN() const long long ck0 = (B.tid == 0) ? (long long)clock64() : 0
#define GM_FS_CLOCK_END() do { if (B.tid == 0) S.ph[PH_FS_REST] += (long long)clock64() - ck0; } while (0)     // shader-clock ticks of the same span
#else
#define GM_FS_CLOCK_BEGIN() do {} while (0)
#define GM_FS_CLOCK_END() do {} while (0)
#endif

// Features [f0, f1) (f0 even, f1 <= K & ~1) of the K x M mat-vec over Gram rows + the S/Q update.  Row ids
// and `vec` are already staged in LDS.  Every thread owns SQ_Q PAIRS of adjacent features and fetches
// each pair with one 16-byte load (row base in SGPRs + 32-bit lane offset); NR rows per trip, so
// SQ_Q * NR independent coalesced 1 KB row segments per wave are in flight.  The sum of a feature runs
// over the rows in order whatever SQ_Q / NR are.
template <int SQ_Q, int NR>
DEV void gm_sq_core(gptr_cc G, lptr_d lvec, lptr_i lused, const GmWork &NOALIAS W, int K, int M, int mode, double beta, double c1,
                    double c2, const double *newrow, int f0, int f1, int tid, int nthr)
{
    typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));   // rows of an odd-K matrix start 8 bytes off
    typedef const d2 __attribute__((address_space(1))) *gptr_cd2;
    for (int ib = f0; ib < f1; ib += 2 * SQ_Q * nthr) {
        d2 accq[SQ_Q];
        unsigned off[SQ_Q];
#pragma unroll
        for (int q = 0; q < SQ_Q; q++) {
            accq[q] = d2{0, 0};
            const int i = ib + 2 * (q * nthr + tid);
            off[q] = (unsigned)((i < f1 ? i : f1 - 2) * 8);
        }
        int j = 0;
        for (; j + NR - 1 < M; j += NR) {
            d2 g[NR][SQ_Q];
            double v[NR];
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const gptr_cc row = G + (size_t)uni(lused[j + r]) * (size_t)K * 8;
                v[r] = lvec[j + r];
#pragma unroll
                for (int q = 0; q < SQ_Q; q++) g[r][q] = *(gptr_cd2)(row + off[q]);
            }
#pragma unroll
            for (int r = 0; r < NR; r++)
#pragma unroll
                for (int q = 0; q < SQ_Q; q++) accq[q] += g[r][q] * v[r];
        }
        for (; j < M; j++) {
            const gptr_cc row = G + (size_t)uni(lused[j]) * (size_t)K * 8;
            const double v0 = lvec[j];
#pragma unroll
            for (int q = 0; q < SQ_Q; q++) accq[q] += *(gptr_cd2)(row + off[q]) * v0;
        }
#pragma unroll
        for (int q = 0; q < SQ_Q; q++) {
            const int i = ib + 2 * (q * nthr + tid);
            if (i < f1) {
                gm_sq_apply(W, mode, beta, c1, c2, newrow, i, accq[q][0]);
                gm_sq_apply(W, mode, beta, c1, c2, newrow, i + 1, accq[q][1]);
            }
        }
    }
}
// The vector and the Gram row ids of the M rows of a sweep -> LDS.  A delete whose sweep was held back (gm_inner) runs
// after its slot shuffle: `del_jj` >= 0 names the freed slot and `del_row` the Gram row that sat there; the row that
// moved into it goes back to the end of the list.
DEV void gm_sq_stage(const Blk &NOALIAS B, const GmWork &NOALIAS W, int M, const double *vec, lptr_d lvec, lptr_i lused, int del_jj, int del_row)
{
    blk_sync(B);
    for (int j = B.tid; j < M; j += B.nthr) {
        lvec[j] = vec[j];
        int r = W.rowid[j];
        if (del_jj >= 0) { if (j == del_jj) r = del_row; else if (j == M - 1) r = W.rowid[del_jj]; }
        lused[j] = r;
    }
    blk_sync(B);
}
// stage row ids and the vector in LDS, then tiles [t0, t1) of 128 features with one pair per thread
// (the shape a claimed chunk has: owner and helpers)
DEV void gm_sq_tiles(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, int M, const double *vec, int mode, double beta,
                     double c1, double c2, const double *newrow, int t0, int t1, bool stage, int del_jj = -1, int del_row = -1)
{
    const gptr_cc G = (gptr_cc)as_global(uni_ptr(F.G));
    const lptr_d lvec = as_lds(B.pool);
    const lptr_i lused = as_lds((int *)(B.pool + ((M + 1) & ~1)));
    const int tid = B.tid, nthr = uni(B.nthr);
    if (stage) gm_sq_stage(B, W, M, vec, lvec, lused, del_jj, del_row);
    const int Kp = K & ~1, f0 = t0 * SQ_FT, f1 = t1 * SQ_FT < Kp ? t1 * SQ_FT : Kp;
    gm_sq_core<1, 8>(G, lvec, lused, W, uni(K), uni(M), mode, beta, c1, c2, newrow, f0, f1, tid, nthr);
}

// a[i] = sum_j G[used[j], i] * vec[j] for all features, fused with the S_in/Q_in update that
// consumes it.  `rid`: Gram row id of the new feature (mode 1), else -1.  del_jj / del_row: see gm_sq_stage.
DEVNI void gm_sq_update(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, int M, const double *vec,
                        int mode, double beta, double c1, double c2, int rid, GmScalars &NOALIAS S, int del_jj = -1, int del_row = -1)
{
    const double *newrow = rid >= 0 ? F.G + (size_t)rid * K : nullptr;
    CNT(c.sum_m_swept += M);
    PH_BEGIN();
    const gptr_cc G = (gptr_cc)as_global(uni_ptr(F.G));
    const lptr_d lvec = as_lds(B.pool);
    const lptr_i lused = as_lds((int *)(B.pool + ((M + 1) & ~1)));
    const int tid = B.tid, nthr = uni(B.nthr);
    K = uni(K); M = uni(M);
    gm_sq_stage(B, W, M, vec, lvec, lused, del_jj, del_row);
    const int Kp = K & ~1;                                    // pairs cover [0, Kp); an odd last feature is handled below
    const int n_tiles = (Kp + SQ_FT - 1) / SQ_FT;
    // the job carries (mode, rid | the deleted slot's Gram row, the deleted slot) so that a helper stages the same rows
    const bool shared = M >= 96 && job_share(B, S, JOB_SQ, M, n_tiles, beta, mode, mode == 2 ? del_row : rid, del_jj, c1, c2, [&](int t0, int t1) {
        gm_sq_tiles(B, F, W, K, M, vec, mode, beta, c1, c2, newrow, t0, t1, false);
    });
    if (!shared) gm_sq_core<5, 2>(G, lvec, lused, W, K, M, mode, beta, c1, c2, newrow, 0, Kp, tid, nthr);
    if ((K & 1) && tid == 0) {                                // the odd last feature
        double a = 0;
        for (int j = 0; j < M; j++) a += *(gptr_cd)(G + ((size_t)lused[j] * (size_t)K + (K - 1)) * 8) * lvec[j];
        gm_sq_apply(W, mode, beta, c1, c2, newrow, K - 1, a);
    }
    PH_END(PH_KSWEEP);
    blk_sync(B);
}

// Sigma[j][i] += a_j * b_i over the M x M block (the rank-1 update every action ends with), a_j = fa(j),
// b_i = fb(i).  On the device both vectors are staged in LDS (at `scr`, 2 M doubles) and each wave keeps four
// column chunks of b in registers while it walks its rows, so the only memory traffic is the coalesced
// read-modify-write of Sigma with four loads in flight per wave; one fma per element either way.
// one pass of gm_rank1 over CP column chunks of 64 starting at column i0 (compile-time chunk count: straight-line loads)
template <int CP>
DEV void gm_rank1_pass(const gptr_d Sg, int ld, int M, const lptr_d la, const lptr_d lb, int i0, int wave, int nwave)
{
    double br[CP];
#pragma unroll
    for (int c = 0; c < CP; c++) { const int i = i0 + c * BLK_LANES; br[c] = i < M ? lb[i] : 0.0; }
    for (int j = wave; j < M; j += 2 * nwave) {
        const int j2 = j + nwave;
        const bool two = j2 < M;
        const int j2c = two ? j2 : j;
        const double f0 = la[j], f1 = two ? la[j2] : 0.0;
        double s0[CP], s1[CP];
        // unguarded loads from clamped addresses (a guarded load becomes a branch and a full wait per element); the values
        // of lanes past the block are never stored
#pragma unroll
        for (int c = 0; c < CP; c++) {
            const int i = i0 + c * BLK_LANES, ic = i < M ? i : M - 1;
            s0[c] = Sg[(size_t)j * ld + ic];
            s1[c] = Sg[(size_t)j2c * ld + ic];
        }
#pragma unroll
        for (int c = 0; c < CP; c++) {
            const int i = i0 + c * BLK_LANES;
            double v0 = s0[c], v1 = s1[c];
            asm volatile("" : "+v"(v0), "+v"(v1));              // the loads stay where they were issued
            if (i < M) {
                Sg[(size_t)j * ld + i] = v0 + f0 * br[c];
                if (two) Sg[(size_t)j2 * ld + i] = v1 + f1 * br[c];
            }
        }
    }
}
template <class FA, class FB>
DEV void gm_rank1(const Blk &NOALIAS B, const GmWork &NOALIAS W, int M, double *scr, FA fa, FB fb)
{
    const int ld = W.ld;
    const lptr_d la = as_lds(scr), lb = as_lds(scr + M);
    blk_sync(B);
    PAR(i, M) { la[i] = fa(i); lb[i] = fb(i); }
    blk_sync(B);
    const gptr_d Sg = as_global_rw(W.Sig);
    const int lane = B.lane, wave = B.wave, nwave = B.nwave;
    // column chunks of 64, up to 8 per pass, two rows per trip (up to 16 loads in flight); one fma per element whatever
    // the split
    const int NC = (M + BLK_LANES - 1) / BLK_LANES;
    if (NC <= 1) gm_rank1_pass<1>(Sg, ld, M, la, lb, lane, wave, nwave);
    else if (NC == 2) gm_rank1_pass<2>(Sg, ld, M, la, lb, lane, wave, nwave);
    else if (NC == 3) gm_rank1_pass<3>(Sg, ld, M, la, lb, lane, wave, nwave);
    else if (NC == 4) gm_rank1_pass<4>(Sg, ld, M, la, lb, lane, wave, nwave);
    else if (NC <= 6) gm_rank1_pass<6>(Sg, ld, M, la, lb, lane, wave, nwave);
    else for (int c0 = 0; c0 < NC; c0 += 8) gm_rank1_pass<8>(Sg, ld, M, la, lb, c0 * BLK_LANES + lane, wave, nwave);
}

// Gram row of feature u = the reference's BASIS_PHI row for that basis (MainEff.c:1608-1630):
// G[i] = x_i . (x_u / scale_u) / scale_i.  Returns the row id r with the row at F.G + r*K, or -1.
//   full mode (F.lazy == 0): every row was computed by gram_kernel; r = u.
//   lazy mode: K x K does not fit in HBM.  Rows live in a per-fold pool shared by every workgroup
//   working on that fold and are computed on first use by sweeping the fold's design once (what
//   the reference does at every add of every fit).  slot_of[u]: -1 absent, -2 being computed,
//   >= 0 pool slot, published with an agent-scope release / consumed behind an agent-scope acquire.
//   A row is written once before it is published and never again, and its values do not depend on
//   who computed it (fixed summation order), so results stay independent of timing.  A waiting
//   workgroup waits only for one that is computing (the grid is fully resident and a computing
//   workgroup never waits); the wait is bounded anyway and falls back to a private copy.
//   When the pool is exhausted the row goes into one of the workgroup's private rows (released
//   again at delete / end of fit): the reference's own per-fit BASIS_PHI, one design sweep per add.
#define ROW_LOAD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ROW_CAS(p, e, d) __hip_atomic_compare_exchange_strong(p, &(e), d, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ROW_STORE(p, v) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ROW_FETCH_ADD(p, v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ROW_SPIN_MAX 400000       // x ~1 us: far beyond one row sweep
DEVNI int gm_row(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, int u)
{
    if (!F.lazy) return u;
    enum { R_OWNER = -3, R_PRIVATE = -4 };
    blk_sync(B);
    if (B.tid == 0) {
        int *st = F.slot_of + u;
        int s = ROW_LOAD(st);
        if (s == -1) {
            int expect = -1;
            if (ROW_CAS(st, expect, -2)) s = R_OWNER; else s = expect;
        }
        for (int spin = 0; s == -2 && spin < ROW_SPIN_MAX; spin++) {
            __builtin_amdgcn_s_sleep(32);
            s = ROW_LOAD(st);
        }
        if (s == -2 || s == -1) s = R_PRIVATE;                  // timed out / the owner found the pool full
        int my = -1;
        if (s == R_OWNER) {
            if (ROW_LOAD(F.pool_next) < F.pool_rows) my = ROW_FETCH_ADD(F.pool_next, 1);
            if (my >= 0 && my < F.pool_rows) my += F.pool_base;
            else { my = -1; ROW_STORE(st, -1); s = R_PRIVATE; }    // pool exhausted
        }
        if (s == R_PRIVATE && W.pfree[0] > 0) my = W.pfree[W.pfree[0]--];   // one of this fit's own rows
        B.ired[0] = s;
        B.ired[1] = my;
        if (s >= 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    blk_sync(B);
    const int s = B.ired[0], my = B.ired[1];
    blk_sync(B);
    if (s >= 0) return s;
    if (my < 0) return -1;
    const int N = F.N;
    double *row = const_cast<double *>(F.G) + (size_t)my * K;
    const double *xu = F.X + (size_t)u * N;
    // PHI as the reference forms it: a main-effect column times the reciprocal of its norm (:517-520), a pair column divided by it (Full2.c:544)
    const double su = u < F.n_main ? 1.0 : F.scale[u], ru = u < F.n_main ? F.rscale[u] : 1.0;
    const bool in_lds = N <= B.pool_n;
    if (in_lds) {
        PAR(h, N) B.pool[h] = xu[h] * ru / su;                  // one of the two factors is exactly 1
        blk_sync(B);
    }
    // eight features per wave and reduction tree (wave_sum8 pairs lanes exactly like wave_sum: same bits)
    for (int i0 = B.wave * 8; i0 < K; i0 += B.nwave * 8) {
        double a[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const double *xi = F.X + (size_t)(i0 + c < K ? i0 + c : K - 1) * N;
            double t = 0;
            if (in_lds) for (int h = B.lane; h < N; h += BLK_LANES) t += xi[h] * B.pool[h];
            else for (int h = B.lane; h < N; h += BLK_LANES) t += xi[h] * (xu[h] * ru / su);
            a[c] = t;
        }
        wave_sum8(a, B.lane);
        const int i = i0 + (B.lane >> 3);
        if ((B.lane & 7) == 0 && i < K) row[i] = a[0] / F.scale[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
    blk_sync(B);
    if (s == R_OWNER && B.tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ROW_STORE(F.slot_of + u, my);
    }
    blk_sync(B);
    return my;
}

// out[i] = sum_j Sigma[i][j] v[j], j ascending, one fma chain per row (the reference's loop order, so the
// bits do not depend on how the work is laid out).  A thread per row reading its own row would touch 64
// cache lines per wave-load; instead each wave takes 64 rows and moves them CW columns at a time through a
// private LDS tile: coalesced 16-byte loads (CW/2 lanes per row), transposed by the tile (pitch CW+1,
// conflict-free), then every lane walks its own row out of LDS.  The next chunk's loads are in flight while
// the current one is summed.  scr: (M rounded up to 16) + nwave * 64 * (CW+1) doubles of LDS.
#define MV_LDS(M, nwave, CW) ((((M) + 15) & ~15) + (nwave) * 64 * ((CW) + 1))
template <int CW>
DEV void gm_sigma_matvec(const Blk &NOALIAS B, const GmWork &NOALIAS W, int M, const double *v, double *out, double *scr, int scr_n, lptr_d out_lds)
{
    const int ld = W.ld;
    if (MV_LDS(M, B.nwave, CW) > scr_n) {                       // no room for the tiles: a thread per row
        PAR(i, M) {
            double a = 0;
            for (int j = 0; j < M; j++) a += W.Sig[(size_t)i * ld + j] * v[j];
            out[i] = a;
            if (out_lds) out_lds[i] = a;
        }
        return;
    }
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef const d2 __attribute__((address_space(1))) *gptr_cd2;
    constexpr int LPR = CW / 2, RPI = 64 / LPR, NI = 64 / RPI, TP = CW + 1;
    const int lane = B.lane, wave = B.wave, nwave = B.nwave;
    const lptr_d lv = as_lds(scr);
    const lptr_d tile = lv + ((M + 15) & ~15) + wave * 64 * TP;
    blk_sync(B);
    PAR(j, M) lv[j] = v[j];
    blk_sync(B);
    const gptr_cd Sg = as_global(W.Sig);
    const int lr = lane / LPR, lc = (lane % LPR) * 2;
    for (int r0 = wave * 64; r0 < M; r0 += nwave * 64) {
        size_t rowoff[NI];
#pragma unroll
        for (int q = 0; q < NI; q++) { int r = r0 + q * RPI + lr; if (r > M - 1) r = M - 1; rowoff[q] = (size_t)r * ld + lc; }
        d2 nx[NI];
#pragma unroll
        for (int q = 0; q < NI; q++) nx[q] = *(gptr_cd2)(Sg + rowoff[q]);
        double a = 0;
        for (int j0 = 0; j0 < M; j0 += CW) {
            d2 cur[NI];
#pragma unroll
            for (int q = 0; q < NI; q++) cur[q] = nx[q];
            if (j0 + CW < M) {
#pragma unroll
                for (int q = 0; q < NI; q++) nx[q] = *(gptr_cd2)(Sg + rowoff[q] + j0 + CW);
            }
#pragma unroll
            for (int q = 0; q < NI; q++) {
                tile[(q * RPI + lr) * TP + lc] = cur[q][0];
                tile[(q * RPI + lr) * TP + lc + 1] = cur[q][1];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): the tile is written
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (j0 + CW <= M) {
#pragma unroll
                for (int c = 0; c < CW; c++) a += tile[lane * TP + c] * lv[j0 + c];
            } else {
#pragma unroll
                for (int c = 0; c < CW; c++) if (j0 + c < M) a += tile[lane * TP + c] * lv[j0 + c];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_s_waitcnt(0xC07F);                 // the tile is consumed before it is overwritten
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (r0 + lane < M) { out[r0 + lane] = a; if (out_lds) out_lds[r0 + lane] = a; }
    }
}

// the new slot's column of the fit's copy of the Gram block (gm_hessian_build reads it)
DEV void gm_gc_add(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, const GmScalars &NOALIAS S, int M, int nu, int rid)
{
    const int ld = W.ld;
    if (S.gc_ok) {
        const gptr_cd G = as_global(F.G);
        PAR(j, M) W.Gc[(size_t)j * ld + M] = G[(size_t)W.rowid[j] * K + nu];
        if (B.tid == 0) W.Gc[(size_t)M * ld + M] = G[(size_t)rid * K + nu];
    }
}

// Lazy Gram mode: rows of several features in ONE sweep of the fold's design (gm_row does one row per
// sweep; a run of T adds needs T of them).  For every feature of `nus` that nobody has computed or is
// computing, this workgroup claims the row (same protocol as gm_row), takes a pool slot, then sweeps the
// design once with all claimed features' columns staged in LDS -- each design column is loaded once and
// dotted with every staged column (same lane assignment and reduction as gm_row: identical values) --
// and publishes the rows.  Anything it cannot claim (already there, in flight elsewhere, pool full) is
// simply left to the gm_row calls that follow.
#define ROWS_MAX 16
DEVNI void gm_rows_prefetch(const Blk &NOALIAS B, const FoldDev &NOALIAS F, int K, const int *nus, int T)
{
    if (!F.lazy || T < 2) return;
    const int N = F.N;
    int cmax = B.pool_n / (N > 0 ? N : 1);                      // staged columns that fit in the LDS pool
    if (cmax > ROWS_MAX) cmax = ROWS_MAX;
    if (cmax < 2) return;
    const lptr_i lfeat = as_lds(B.ired + 2 * BLK_MAX_WAVES), lslot = as_lds(B.ired + 2 * BLK_MAX_WAVES + ROWS_MAX);
    blk_sync(B);
    if (B.tid == 0) {
        int c = 0;
        for (int t = 0; t < T && c < cmax; t++) {
            int *st = F.slot_of + nus[t];
            int expect = -1;
            if (ROW_LOAD(st) != -1 || !ROW_CAS(st, expect, -2)) continue;            // present or in flight: not ours
            int my = -1;
            if (ROW_LOAD(F.pool_next) < F.pool_rows) my = ROW_FETCH_ADD(F.pool_next, 1);
            if (my < 0 || my >= F.pool_rows) { ROW_STORE(st, -1); break; }           // pool exhausted: leave it to gm_row
            lfeat[c] = nus[t]; lslot[c] = my + F.pool_base; c++;
        }
        B.ired[0] = c;
    }
    blk_sync(B);
    const int C = B.ired[0];
    blk_sync(B);
    if (C == 0) return;
    for (int c = 0; c < C; c++) {
        const double *xu = F.X + (size_t)lfeat[c] * N;
        const bool mainc = lfeat[c] < F.n_main;
        const double su = mainc ? 1.0 : F.scale[lfeat[c]], ru = mainc ? F.rscale[lfeat[c]] : 1.0;
        PAR(h, N) B.pool[c * N + h] = xu[h] * ru / su;
    }
    blk_sync(B);
    double *Gw = const_cast<double *>(F.G);
    for (int i = B.wave; i < K; i += B.nwave) {
        const double *xi = F.X + (size_t)i * N;
        double a[ROWS_MAX];
#pragma unroll
        for (int c = 0; c < ROWS_MAX; c++) a[c] = 0;
        for (int h = B.lane; h < N; h += BLK_LANES) {
            const double x = xi[h];
#pragma unroll
            for (int c = 0; c < ROWS_MAX; c++) if (c < C) a[c] += x * B.pool[c * N + h];
        }
        const double sc = F.scale[i];
#pragma unroll
        for (int c0 = 0; c0 < ROWS_MAX; c0 += 8) {              // eight staged columns per reduction tree
            if (c0 < C) {
                double v[8];
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] = a[c0 + c];
                wave_sum8(v, B.lane);
                const int c = c0 + (B.lane >> 3);
                if ((B.lane & 7) == 0 && c < C) Gw[(size_t)lslot[c] * K + i] = v[0] / sc;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its stores
    blk_sync(B);
    if (B.tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int c = 0; c < C; c++) ROW_STORE(F.slot_of + lfeat[c], lslot[c]);
    }
    blk_sync(B);
}

// S_in / Q_in update of one feature by one add (gm_sq_apply mode 1), with the products spelled out so
// that the K-space sweep and the M-space tracking below round identically.
DEV void gm_add_apply(double &sin, double &qin, double beta, double rowval, double a, double sii, double mui)
{
    const double mc = beta * rowval - beta * a;
    sin = sin - mc * mc * sii;
    qin = qin - mui * mc;
}

// The K-space half of a run of T consecutive adds: ONE sweep over the Gram rows of the (final) active
// set instead of T.  For add t (active-set size M0 + t when it is applied, vector vb[t]) feature i needs
// a_t[i] = sum_{j < M0+t} G[row_j][i] vb[t][j]; the vectors are staged zero-padded in LDS, so every thread
// loads each row element once and feeds TT accumulators; the new features' own rows (the rows M0 .. M0+T-1
// of the sweep) are picked up on the way.  Then the T updates are applied in order.  TT = T rounded up.
template <int TT>
DEV void gm_sq_batch(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, int M0, int T, double beta, int f0, int f1)
{
    typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
    typedef const d2 __attribute__((address_space(1))) *gptr_cd2;
    const gptr_cc G = (gptr_cc)as_global(uni_ptr(F.G));
    const int tid = B.tid, nthr = uni(B.nthr);
    const int Mt = M0 + T;                                      // rows of the sweep
    const int ldv = W.cap + 2;
    const lptr_d lvb = as_lds(B.pool);                          // [TT][Mt] zero-padded vectors
    const lptr_d lsc = as_lds(B.pool + TT * Mt);                // sii[TT], mui[TT]
    const lptr_i lused = as_lds((int *)(B.pool + TT * Mt + 2 * TT));
    blk_sync(B);
    for (int e = tid; e < TT * Mt; e += nthr) {
        const int t = e / Mt, j = e - t * Mt;
        lvb[e] = (t < T && j < M0 + t) ? W.vb[(size_t)t * ldv + j] : 0.0;
    }
    for (int t = tid; t < TT; t += nthr) { lsc[t] = t < T ? W.bsc[t] : 0.0; lsc[TT + t] = t < T ? W.bsc[ADD_TB + t] : 0.0; }
    for (int j = tid; j < Mt; j += nthr) lused[j] = W.rowid[j];
    blk_sync(B);
    K = uni(K);
    const int Kp = f1;                                          // features [f0, f1), both even, f1 <= K & ~1
    for (int ib = f0; ib < Kp; ib += 2 * nthr) {
        const int i = ib + 2 * tid;
        const unsigned off = (unsigned)((i < Kp ? i : Kp - 2) * 8);
        d2 acc[TT], rowv[TT];
#pragma unroll
        for (int t = 0; t < TT; t++) { acc[t] = d2{0, 0}; rowv[t] = d2{0, 0}; }
        int j = 0;
        for (; j + 1 < M0; j += 2) {                            // rows of the old active set, two per trip
            const d2 g0 = *(gptr_cd2)(G + (size_t)uni(lused[j]) * (size_t)K * 8 + off);
            const d2 g1 = *(gptr_cd2)(G + (size_t)uni(lused[j + 1]) * (size_t)K * 8 + off);
#pragma unroll
            for (int t = 0; t < TT; t++) { acc[t] += g0 * lvb[t * Mt + j]; acc[t] += g1 * lvb[t * Mt + j + 1]; }
        }
        if (j < M0) {
            const d2 g0 = *(gptr_cd2)(G + (size_t)uni(lused[j]) * (size_t)K * 8 + off);
#pragma unroll
            for (int t = 0; t < TT; t++) acc[t] += g0 * lvb[t * Mt + j];
        }
#pragma unroll
        for (int s2 = 0; s2 < TT; s2++) {                       // the new features' own rows: row M0+s2 feeds the adds after s2
            if (s2 < T) {
                const d2 g = *(gptr_cd2)(G + (size_t)uni(lused[M0 + s2]) * (size_t)K * 8 + off);
                rowv[s2] = g;
#pragma unroll
                for (int t = s2 + 1; t < TT; t++) acc[t] += g * lvb[t * Mt + M0 + s2];
            }
        }
        if (i < Kp) {
            double s0 = W.Sin[i], q0 = W.Qin[i], s1 = W.Sin[i + 1], q1 = W.Qin[i + 1];
#pragma unroll
            for (int t = 0; t < TT; t++) {
                if (t < T) {
                    gm_add_apply(s0, q0, beta, rowv[t][0], acc[t][0], lsc[t], lsc[TT + t]);
                    gm_add_apply(s1, q1, beta, rowv[t][1], acc[t][1], lsc[t], lsc[TT + t]);
                }
            }
            W.Sin[i] = s0; W.Qin[i] = q0; W.Sin[i + 1] = s1; W.Qin[i + 1] = q1;
        }
    }
    blk_sync(B);
}

// features [f0, f1) of the sweep of a run of T adds (the owner's whole range, or a claimed chunk of it)
DEV void gm_sq_batch_range(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, int M0, int T, double beta, int f0, int f1)
{
    if (T <= 4) gm_sq_batch<4>(B, F, W, K, M0, T, beta, f0, f1);
    else if (T <= 8) gm_sq_batch<8>(B, F, W, K, M0, T, beta, f0, f1);
    else gm_sq_batch<ADD_TB>(B, F, W, K, M0, T, beta, f0, f1);
}

// the whole sweep: shared with idle workgroups when the job board is open (same arithmetic per feature
// whoever runs it), the odd last feature by the owner
DEVNI void gm_sq_batch_all(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, int M0, int T, double beta, GmScalars &NOALIAS S)
{
    CNT(c.sum_m_swept += M0 + T);                               // one sweep of the final active set's rows for the whole run
    const int Kp = K & ~1;
    const int n_tiles = (Kp + SQ_FT - 1) / SQ_FT;
    const bool shared = M0 + T >= 96 && job_share(B, S, JOB_SQB, M0, n_tiles, beta, T, -1, -1, 0.0, 0.0, [&](int t0, int t1) {
        gm_sq_batch_range(B, F, W, K, M0, T, beta, t0 * SQ_FT, t1 * SQ_FT < Kp ? t1 * SQ_FT : Kp);
    });
    if (!shared) gm_sq_batch_range(B, F, W, K, M0, T, beta, 0, Kp);
    if ((K & 1) && B.tid == 0) {                                // the odd last feature
        const int i = K - 1, ldv = W.cap + 2;
        double s0 = W.Sin[i], q0 = W.Qin[i];
        for (int t = 0; t < T; t++) {
            double a = 0;
            for (int j = 0; j < M0 + t; j++) a += F.G[(size_t)W.rowid[j] * K + i] * W.vb[(size_t)t * ldv + j];
            gm_add_apply(s0, q0, beta, F.G[(size_t)W.rowid[M0 + t] * K + i], a, W.bsc[t], W.bsc[ADD_TB + t]);
        }
        W.Sin[i] = s0; W.Qin[i] = q0;
    }
    blk_sync(B);
}

// A run of T (2 .. ADD_TB) consecutive ADD actions of one block update = gm_add applied T times
// (MainEff.c:1585-1723 + :613-627), restructured: the M-space part of every add (Sigma border, mu) runs in
// sequence as before, but the K-space part -- each add's sweep over all Gram rows, which is what an action
// costs -- is deferred and done for the whole run in ONE sweep (gm_sq_batch).  The only K-space values the
// M-space part needs in between are S_in / Q_in of the run's own features (sii, mui of the later adds):
// those T values are tracked on the side with the same arithmetic (one lane per feature, rows in order).
// W.rowid[M0 .. M0+T) must already hold the Gram row ids of the run's features.
DEVNI void gm_add_batch(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S, const int *nus, int T, bool defer)
{
    const int M0 = S.M, ld = W.ld, ldv = W.cap + 2, Mt = M0 + T;
    const double beta = S.beta;
    double *sii_v = W.bsc, *mui_v = W.bsc + ADD_TB, *sin_v = W.bsc + 2 * ADD_TB, *qin_v = W.bsc + 3 * ADD_TB;
    // LDS: gb[u][j] = G[row_j][nus[u]], the Gram values of the run's own features at every row of the final
    // active set (gathered once, in parallel; the side tracking below then runs out of LDS), the current
    // add's vector, and the scratch of the rank-1 update.
    const lptr_d gb = as_lds(B.pool), lv2 = gb + (size_t)T * Mt;
    double *scr = B.pool + (size_t)T * Mt + Mt;
    const lptr_i lrow = as_lds((int *)scr);
    const gptr_cd G = as_global(F.G);
    blk_sync(B);
    if (B.tid < T) { sin_v[B.tid] = W.Sin[nus[B.tid]]; qin_v[B.tid] = W.Qin[nus[B.tid]]; }
    PAR(j, Mt) lrow[j] = W.rowid[j];
    blk_sync(B);
    for (int e = B.tid; e < T * Mt; e += B.nthr) {
        const int u = e / Mt, j = e - u * Mt;
        gb[e] = G[(size_t)lrow[j] * K + nus[u]];
    }
    blk_sync(B);
    for (int t = 0; t < T; t++) {
        const int M = M0 + t, nu = nus[t], rid = W.rowid[M];
        const double newA = W.aroot[nu];
        const double *row = F.G + (size_t)rid * K;
        double *v2 = W.vb + (size_t)t * ldv;
        PAR(l, M) W.v1[l] = beta * row[W.used[l]];
        blk_sync(B);
        { PH_BEGIN();
        gm_sigma_matvec<8>(B, W, M, W.v1, v2, scr, B.pool_n - T * Mt - Mt, lv2);
        PH_END(PH_MATVEC); }
        const double sii = 1.0 / (newA + sin_v[t]);
        const double mui = sii * qin_v[t];
        blk_sync(B);
        PAR(i, M) W.mu[i] += -mui * v2[i];
        // S_in / Q_in of the run's later features after this add: lane 0 of wave (u - t - 1) mod nwave walks
        // the rows in order (the same fma chain as the sweep in gm_sq_batch), operands from LDS
        { PH_BEGIN();
        for (int u = t + 1 + B.wave; u < T; u += B.nwave) {
            if (B.lane == 0) {
                const lptr_d gu = gb + (size_t)u * Mt;
                double a = 0;
#pragma unroll 8
                for (int j = 0; j < M; j++) a += gu[j] * lv2[j];
                gm_add_apply(sin_v[u], qin_v[u], beta, gu[M], a, sii, mui);
            }
        }
        PH_END(PH_TRACK); }
        { PH_BEGIN();
        gm_rank1(B, W, M, scr, [&](int j) { return sii * v2[j]; }, [&](int i) { return v2[i]; });
        PH_END(PH_RANK1); }
        PAR(i, M) {
            const double si = -sii * v2[i];
            W.Sig[(size_t)M * ld + i] = si;
            W.Sig[(size_t)i * ld + M] = si;
        }
        if (B.tid == 0) {
            W.Sig[(size_t)M * ld + M] = sii;
            W.A[M] = newA;
            W.mu[M] = mui;
            W.used[M] = nu;
            W.upos[nu] = M;
            sii_v[t] = sii; mui_v[t] = mui;
        }
        if (S.gc_ok) PAR(j, M + 1) W.Gc[(size_t)j * ld + M] = gb[(size_t)t * Mt + j];   // the new slot's column of the Gram block cache
        blk_sync(B);
    }
    if (defer) { S.pend.kind = 4; S.pend.M = M0; S.pend.T = T; S.pend.beta = beta; }
    else {
        PH_BEGIN();
        gm_sq_batch_all(B, F, W, K, M0, T, beta, S);
        PH_END(PH_KSWEEP);
    }
    S.M = M0 + T;
}

// A run of consecutive adds at todo[u ..]: one Gram-row sweep for all of them (gm_add_batch).  Returns the number of adds
// applied (0: no run here, the caller takes the single-action path; -1: the fit must stop, status set).
DEV int gm_add_run(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S, int u, int n_todo, bool defer_ok)
{
    int T = 1;
    while (u + T < n_todo && T < ADD_TB && W.act[W.todo[u + T]] == ACT_ADD) T++;
    if (S.M + T > W.cap) T = W.cap - S.M;       // the add that overflows is left to the single path
    for (;;) {                                    // LDS of the sweep: TT zero-padded vectors of M0 + T
        const int TT = T <= 4 ? 4 : (T <= 8 ? 8 : ADD_TB), Mt = S.M + T;
        if (T < 2 || (TT * Mt + 2 * TT + (Mt + 1) / 2 + 8 <= B.pool_n && T * Mt + Mt + MV_LDS(Mt, B.nwave, 8) <= B.pool_n)) break;
        T--;
    }
    if (T < 2) return 0;
    gm_rows_prefetch(B, F, K, W.todo + u, T);   // lazy Gram mode: the run's missing rows in one design sweep
    bool ok = true;
    for (int t = 0; t < T && ok; t++) {
        const int rid = gm_row(B, F, W, K, W.todo[u + t]);
        if (rid < 0) ok = false;
        else if (B.tid == 0) W.rowid[S.M + t] = rid;
    }
    if (!ok) { S.status |= ST_OVERFLOW | ST_ABORT; return -1; }
    CNT(c.n_add += T; c.sum_m_action += (int64_t)T * S.M + (int64_t)T * (T - 1) / 2);
    gm_add_batch(B, F, W, K, S, W.todo + u, T, defer_ok && u + T == n_todo);
    if (S.M > W.cap_flag) S.status |= ST_OVERFLOW;   // past the reference's basisMax (MainEff.c:605-611): flagged, not stopped
    return T;
}
// the held-back sweep of a run of adds (S.pend.kind == 4)
DEV void gm_flush_add_run(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S, int M0, int T, double beta)
{
    PH_BEGIN();
    gm_sq_batch_all(B, F, W, K, M0, T, beta, S);
    PH_END(PH_KSWEEP);
}

// Blocked form of the same elimination for the GPU: the symmetric sweep operator applied 16
// pivots at a time on the lower triangle,
//     A11 <- -A11^-1,   A21 <- A21 A11^-1,   A22 <- A22 - A21 A11^-1 A21'
// (after all blocks the matrix holds -H^-1; one last pass negates and mirrors).  The rank-16
// update of A22 runs on the FP64 matrix cores, one 16 x 16 tile per MFMA group, so the matrix
// makes M/16 round trips through L2 instead of M.  LDS: Tn = -A21 A11^-1 (M x 16, pitch 18) and
// the 16 x 16 pivot block.
#define INV_TP 18
#ifndef INV_TT
#define INV_TT 2           // tiles of the trailing update each wave keeps in flight
#endif
#ifndef INV_RUN
#define INV_RUN 8          // tiles per run (multiple of INV_TT)
#endif
// ---- two pivot blocks per trip through memory ---------------------------------------------------------------
// The trailing update of the blocked sweep above is bound by the CU's path to L2, not by the matrix cores: a tile is
// loaded, takes four matrix ops and is stored again, once per pivot block.  Here two consecutive pivot blocks k, k+1
// share one trip: after block k's panel product only the tiles that block k+1 will read as its pivot block and pivot
// column (those with tile index k+1: one tile row and one tile column) get block k's update; block k+1's pivot sweeps
// and panel product follow; then every other tile is loaded ONCE, takes block k's four matrix ops and then block
// k+1's, and is stored -- the same chain of operations per element as two separate steps (a store and a reload in
// between change nothing), so the result is bit-identical to gm_spd_inverse_blocked, with half the tile traffic.
// Operands: both panels Tn (block k, block k+1) in LDS; block k's ORIGINAL pivot column (the A21' operand of its
// update, overwritten in Sigma before block k+1's panel product needs the new values there) is kept in the fit's
// HBM scratch Pk[s][i] (16 x Mp, L2-resident), written on the way by block k's panel product.
DEV int inv_pivot(int tid, const gptr_d Sig, int ld, int M, int k0, const lptr_d nD, const lptr_d nD2)
{
    __syncthreads();
    if (tid < 256) {                                       // pivot block (identity-padded)
        const int r = tid & 15, c = tid >> 4, gi = k0 + r, gj = k0 + c;
        double v;
        if (gi < M && gj < M) { const int hi = gi > gj ? gi : gj, lo = gi > gj ? gj : gi; v = Sig[(size_t)lo * ld + hi]; }
        else v = (gi == gj) ? 1.0 : 0.0;
        nD[r * 17 + c] = v;
    }
    __syncthreads();
    // one division per element (numerator chosen first: the same operations as the four cases of the scalar sweep); a pivot
    // that is not positive is noticed by every thread alike and acted upon after the sweeps
    bool bad = false;
    const int r = tid & 15, c = (tid >> 4) & 15;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const lptr_d src = (s & 1) ? nD2 : nD, dst = (s & 1) ? nD : nD2;
        const double d = src[s * 17 + s], prs = src[r * 17 + s], psc = src[s * 17 + c], v = src[r * 17 + c];
        bad |= !(d > 0);
        const bool rs = r == s, cs = c == s;
        const double num = rs ? (cs ? -1.0 : psc) : (cs ? prs : prs * psc);
        const double t = num / d;
        if (tid < 256) dst[r * 17 + c] = (rs || cs) ? t : v - t;
        __syncthreads();
    }
    return bad ? 1 : 0;
}
// Tn[i][r] = sum_s A(i, k0+s) nD[s][r] for the rows outside the block (zero inside / padding); SAVE: the operand, i.e.
// the pivot column as it is now, goes to Pk[s][i] on the way
template <bool SAVE>
DEV void inv_panel(const Blk &NOALIAS B, const gptr_d Sig, int ld, int M, int nT, int k0, const lptr_d nD, const lptr_d Tn, const gptr_d Pk, int Mp)
{
    const int l15 = B.lane & 15, l4 = B.lane >> 4;
    for (int ti = B.wave; ti < nT; ti += B.nwave) {
        const int i = ti * 16 + l15;
        const bool live = i < M && (i < k0 || i >= k0 + 16);
        d4 acc = d4{0, 0, 0, 0};
        double av[4], bw[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int kc = k0 + kk * 4 + l4;
            double v = 0;
            if (live && kc < M) v = (i > kc) ? Sig[(size_t)kc * ld + i] : Sig[(size_t)i * ld + kc];
            av[kk] = v;
            bw[kk] = nD[(kk * 4 + l4) * 17 + l15];
        }
        if (SAVE) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) Pk[(size_t)(kk * 4 + l4) * Mp + i] = av[kk];
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bw[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) Tn[(size_t)(ti * 16 + l4 + 4 * r) * INV_TP + l15] = acc[r];
    }
}
// pivot column <- A21 A11^-1 = -Tn, pivot block <- -A11^-1
DEV void inv_write_panel(const Blk &NOALIAS B, const gptr_d Sig, int ld, int M, int Mp, int k0, const lptr_d Tn, const lptr_d nD)
{
    for (int e = B.tid; e < Mp * 16; e += B.nthr) {
        const int i = e >> 4, r = e & 15, kc = k0 + r;
        if (i >= M || kc >= M || (i >= k0 && i < k0 + 16)) continue;
        const double v = -Tn[(size_t)i * INV_TP + r];
        if (i > kc) Sig[(size_t)kc * ld + i] = v; else Sig[(size_t)i * ld + kc] = v;
    }
    if (B.tid < 256) {
        const int r = B.tid & 15, c = B.tid >> 4;
        if (k0 + r < M && k0 + c < M) Sig[(size_t)(k0 + c) * ld + k0 + r] = nD[r * 17 + c];
    }
}
// the A21' operand of tile column tj for the pivot block at kb, from Sigma (the pivot column as it is now) ...
DEV void inv_av_sig(const gptr_d Sig, int ld, int M, int tj, int kb, int l15, int l4, double (&av)[4])
{
    const int jrow = tj * 16 + l15;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int kc = kb + kk * 4 + l4;
        double a_ = 0;
        if (jrow < M && kc < M) a_ = (jrow > kc) ? Sig[(size_t)kc * ld + jrow] : Sig[(size_t)jrow * ld + kc];
        av[kk] = a_;
    }
}
// ... or from the saved copy
DEV void inv_av_pk(const gptr_d Pk, int Mp, int tj, int l15, int l4, double (&av)[4])
{
#pragma unroll
    for (int kk = 0; kk < 4; kk++) av[kk] = Pk[(size_t)(kk * 4 + l4) * Mp + tj * 16 + l15];
}
// One run of the trailing update: the tiles (ti = rows(a), tj) for a in [a0, a1), INV_TT of them per trip.
// NS = 2: A += Tn0 av0' then A += Tn1 av1' (two pivot blocks in one trip); NS = 1: the second pair only.
// The tiles of the NEXT trip are requested before the matrix ops of this one are issued (a second set of registers), as
// raw values from clamped addresses -- an unguarded load per element: a guarded one becomes a branch and a full wait
// each -- and the padding is masked to zero where the values are taken over, one trip later.
// (Measured, tools/ubench/inverse_rate.py: the phase runs at ~21 % of the matrix peak at 640 - 1000 columns whatever the
// number of tiles per trip (2, 4), the issue order of stores and loads or the width of the tile loads; with 8 workgroups on
// the chip it takes the same time with neither tile loads nor stores (-DINV_DIAG_NOLOAD -DINV_DIAG_NOSTORE: 3.77 vs 3.87 ms
// at 640 columns), i.e. it is bound by the waves' own instruction streams; with 256 inverting at once memory comes on top
// (6.7 vs 4.4 ms).  The request-ahead is worth 6 %, two pivot blocks per trip 25 - 30 %; the register form removes the
// phase.  profiles/r03/inverse_ablation.txt)
// timing-only ablations of the trailing update (tools/ubench/inverse_rate.py with a -DINV_DIAG_* build; results are wrong)
#ifdef INV_DIAG_NOLOAD
#define INV_DIAG_LOAD(expr, jc, ic) (1e-3 * (double)((jc) + (ic)))
#else
#define INV_DIAG_LOAD(expr, jc, ic) (expr)
#endif
template <int NS, class Rows, class TP>
DEV void inv_run(const gptr_d Sig, int ld, int M, int tj, int a0, int a1, Rows rows, const double (&av0)[4], const double (&av1)[4],
                 const TP Tn0, const TP Tn1, int l15, int l4)
{
    d4 nxt[INV_TT];
#define INV_LOAD_TILES(at)                                                                                           \
    _Pragma("unroll") for (int z = 0; z < INV_TT; z++) {                                                             \
        const int ti = rows((at) + z < a1 ? (at) + z : a1 - 1);                                                      \
        const int icol = ti * 16 + l15, ic = icol < M ? icol : M - 1;                                                \
        _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                              \
            const int j = tj * 16 + l4 + 4 * r, jc = j < M ? j : M - 1;                                              \
            nxt[z][r] = INV_DIAG_LOAD(Sig[(size_t)jc * ld + ic], jc, ic);                                            \
        }                                                                                                            \
    }
    INV_LOAD_TILES(a0)
    for (int a = a0; a < a1; a += INV_TT) {
        d4 acc[INV_TT];
        double bv0[INV_TT][4], bv1[INV_TT][4];
#pragma unroll
        for (int z = 0; z < INV_TT; z++) {
            const int icol = rows(a + z < a1 ? a + z : a1 - 1) * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double v = nxt[z][r];
                asm volatile("" : "+v"(v));                        // the load stays where it was issued
                acc[z][r] = (icol < M && tj * 16 + l4 + 4 * r < M) ? v : 0.0;
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                if (NS == 2) bv0[z][kk] = Tn0[(size_t)icol * INV_TP + kk * 4 + l4];
                bv1[z][kk] = Tn1[(size_t)icol * INV_TP + kk * 4 + l4];
            }
        }
        if (a + INV_TT < a1) { INV_LOAD_TILES(a + INV_TT) }
        if (NS == 2) {
#pragma unroll
            for (int z = 0; z < INV_TT; z++)
#pragma unroll
                for (int kk = 0; kk < 4; kk++) acc[z] = __builtin_amdgcn_mfma_f64_16x16x4f64(av0[kk], bv0[z][kk], acc[z], 0, 0, 0);
        }
#pragma unroll
        for (int z = 0; z < INV_TT; z++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc[z] = __builtin_amdgcn_mfma_f64_16x16x4f64(av1[kk], bv1[z][kk], acc[z], 0, 0, 0);
#pragma unroll
        for (int z = 0; z < INV_TT; z++) {
            if (a + z < a1) {
                const int icol = rows(a + z) * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int j = tj * 16 + l4 + 4 * r;
#ifdef INV_DIAG_NOSTORE
                    if (icol < M && j < M && acc[z][r] == 1.2345e300) Sig[(size_t)j * ld + icol] = acc[z][r];
#else
                    if (icol < M && j < M) Sig[(size_t)j * ld + icol] = acc[z][r];
#endif
                }
            }
        }
    }
#undef INV_LOAD_TILES
}
// TP: where the M x 16 panel Tn lives -- LDS while it fits (M <= 1040 with the 152 KB pool), else the fit's own
// scratch in HBM (W.Tn, L2-resident: 2048 x 18 doubles = 295 KB); the two 16 x 17 pivot blocks are always in LDS.
template <class TP>
DEVNI int gm_spd_inverse_blocked(const Blk &NOALIAS B_, const GmWork &NOALIAS W, int M, long long *phx, const TP Tn, const lptr_d nD)
{
    // B_ is a reference argument of a non-inlined function, i.e. memory: a use inside a loop would be a flat load with its
    // wait (the sixteen sweeps of a pivot tile each paid one) -- a register copy, taken once
    const Blk B{B_.tid, uni(B_.nthr), B_.lane, uni(B_.wave), uni(B_.nwave), B_.red, B_.ired, uni_ptr(B_.pool), uni(B_.pool_n)};
    const int ld = W.ld;
    const gptr_d Sig = as_global_rw(W.Sig);
    const int nT = (M + 15) >> 4, Mp = nT * 16;
    const lptr_d nD2 = nD + 16 * 17;                         // second copy for the pivot sweeps
    const int l15 = B.lane & 15, l4 = B.lane >> 4;
    for (int tk = 0; tk < nT; tk++) {
        const int k0 = tk * 16;
        __syncthreads();
        PHX_BEGIN(t_piv);
        if (B.tid < 256) {                                   // pivot block (identity-padded)
            const int r = B.tid & 15, c = B.tid >> 4, gi = k0 + r, gj = k0 + c;
            double v;
            if (gi < M && gj < M) { const int hi = gi > gj ? gi : gj, lo = gi > gj ? gj : gi; v = Sig[(size_t)lo * ld + hi]; }
            else v = (gi == gj) ? 1.0 : 0.0;
            nD[r * 17 + c] = v;
        }
        __syncthreads();
        {   // scalar sweeps inside the block, ping-pong between two copies: one barrier per sweep, one division per
            // element (numerator chosen first: the operations of the four cases of gm_spd_inverse_scalar); a pivot that
            // is not positive is noticed by every thread alike and acted upon after the sweeps
            bool bad = false;
            const int r = B.tid & 15, c = (B.tid >> 4) & 15;
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const lptr_d src = (s & 1) ? nD2 : nD, dst = (s & 1) ? nD : nD2;
                const double d = src[s * 17 + s], prs = src[r * 17 + s], psc = src[s * 17 + c], v = src[r * 17 + c];
                bad |= !(d > 0);
                const bool rs = r == s, cs = c == s;
                const double num = rs ? (cs ? -1.0 : psc) : (cs ? prs : prs * psc);
                const double t = num / d;
                if (B.tid < 256) dst[r * 17 + c] = (rs || cs) ? t : v - t;
                __syncthreads();
            }
#ifndef INV_DIAG_NOEXIT
            if (bad) return 1;
#endif
        }                                                     // 16 sweeps: the result is back in nD
        PHX_END(t_piv, PH_INV_PIVOT);
        PHX_BEGIN(t_tn);
        // Tn[i][r] = sum_s A(i, k0+s) * nD[s][r] for rows outside the block (zero inside / padding): one 16 x 16 x 16
        // product per row tile on the matrix cores (four chained ops = the same k-ascending fma chain a scalar loop
        // runs, tools/ubench/mfma_f64_order.hip); rows <-> i, columns <-> r
        for (int ti = B.wave; ti < nT; ti += B.nwave) {
            const int i = ti * 16 + l15;
            const bool live = i < M && (i < k0 || i >= k0 + 16);
            d4 acc = d4{0, 0, 0, 0};
            double av[4], bw[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int kc = k0 + kk * 4 + l4;
                double v = 0;
                if (live && kc < M) v = (i > kc) ? Sig[(size_t)kc * ld + i] : Sig[(size_t)i * ld + kc];
                av[kk] = v;
                bw[kk] = nD[(kk * 4 + l4) * 17 + l15];
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bw[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) Tn[(size_t)(ti * 16 + l4 + 4 * r) * INV_TP + l15] = acc[r];
        }
        __syncthreads();
        PHX_END(t_tn, PH_INV_TN);
        // A22 += Tn * A21'  on the lower-triangle tiles that do not touch the pivot block.
        // MFMA rows <-> j (column of A), MFMA columns <-> i (row of A): stores are contiguous in i.
        // The tiles form a triangle over the nT - 1 non-pivot tile indices, cut into runs of up to INV_RUN tiles down one
        // tile column b (same 16 columns j of A, consecutive row tiles a >= b): the A21' operand depends on b only and is
        // loaded once per run.  Runs are dealt to the waves round-robin (inv_run: INV_TT tiles per trip, the next trip's
        // tiles requested ahead).
        {
            const int n1 = nT - 1;
            const auto skip1 = [tk](int c) { return c < tk ? c : c + 1; };   // skip the pivot tile row / column
            const double zero4[4] = {0, 0, 0, 0};
            int cnt = 0;
            for (int b = 0; b < n1; b++) {
                const int tj = skip1(b);
                for (int a0 = b; a0 < n1; a0 += INV_RUN) {
                    if ((cnt++) % B.nwave != B.wave) continue;
                    double av[4];
                    inv_av_sig(Sig, ld, M, tj, k0, l15, l4, av);
                    inv_run<1>(Sig, ld, M, tj, a0, a0 + INV_RUN < n1 ? a0 + INV_RUN : n1, skip1, zero4, av, Tn, Tn, l15, l4);
                }
            }
        }
        __syncthreads();
        // pivot column panel <- A21 A11^-1 = -Tn, pivot block <- -A11^-1
        for (int e = B.tid; e < Mp * 16; e += B.nthr) {
            const int i = e >> 4, r = e & 15, kc = k0 + r;
            if (i >= M || kc >= M || (i >= k0 && i < k0 + 16)) continue;
            const double v = -Tn[(size_t)i * INV_TP + r];
            if (i > kc) Sig[(size_t)kc * ld + i] = v; else Sig[(size_t)i * ld + kc] = v;
        }
        if (B.tid < 256) {
            const int r = B.tid & 15, c = B.tid >> 4;
            if (k0 + r < M && k0 + c < M) Sig[(size_t)(k0 + c) * ld + k0 + r] = nD[r * 17 + c];
        }
    }
    __syncthreads();
    for (int j = B.wave; j < M; j += B.nwave)
        for (int i = j + B.lane; i < M; i += 64) {
            const double v = -Sig[(size_t)j * ld + i];
            Sig[(size_t)j * ld + i] = v;
            Sig[(size_t)i * ld + j] = v;
        }
    __syncthreads();
    return 0;
}

DEVNI int gm_spd_inverse_paired(const Blk &NOALIAS B_, const GmWork &NOALIAS W, int M, long long *phx)
{
    // B_ is a reference argument of a non-inlined function, i.e. memory: a use inside a loop would be a flat load with its
    // wait (the sixteen sweeps of a pivot tile each paid one) -- a register copy, taken once
    const Blk B{B_.tid, uni(B_.nthr), B_.lane, uni(B_.wave), uni(B_.nwave), B_.red, B_.ired, uni_ptr(B_.pool), uni(B_.pool_n)};
    const int ld = W.ld;
    const gptr_d Sig = as_global_rw(W.Sig), Pk = as_global_rw(W.Tn);
    const int nT = (M + 15) >> 4, Mp = nT * 16;
    const lptr_d Tn0 = as_lds(B.pool), Tn1 = Tn0 + (size_t)Mp * INV_TP, nD = Tn1 + (size_t)Mp * INV_TP, nD2 = nD + 16 * 17;
    const int l15 = B.lane & 15, l4 = B.lane >> 4;
    const double zero4[4] = {0, 0, 0, 0};
    int tk = 0;
    for (; tk + 1 < nT; tk += 2) {
        const int k0 = tk * 16, k1 = k0 + 16, t1 = tk + 1;
        // ---- block k: pivot sweeps, panel product (the pivot column saved on the way)
        PHX_BEGIN(t_piv);
        if (inv_pivot(B.tid, Sig, ld, M, k0, nD, nD2)) return 1;
        PHX_END(t_piv, PH_INV_PIVOT);
        PHX_BEGIN(t_tn);
        inv_panel<true>(B, Sig, ld, M, nT, k0, nD, Tn0, Pk, Mp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // Pk: every storing wave drains its stores
        __syncthreads();
        PHX_END(t_tn, PH_INV_TN);
        // ---- block k's update of the tiles block k+1 reads: tile column t1 (rows t1 ..) and tile row t1 (columns < tk)
        {
            int cnt = 0;
            for (int a0 = t1; a0 < nT; a0 += INV_RUN) {
                if ((cnt++) % B.nwave != B.wave) continue;
                double av[4];
                inv_av_pk(Pk, Mp, t1, l15, l4, av);
                inv_run<1>(Sig, ld, M, t1, a0, a0 + INV_RUN < nT ? a0 + INV_RUN : nT, [](int a) { return a; }, zero4, av, Tn0, Tn0, l15, l4);
            }
            for (int b = 0; b < tk; b++) {
                if ((cnt++) % B.nwave != B.wave) continue;
                double av[4];
                inv_av_pk(Pk, Mp, b, l15, l4, av);
                inv_run<1>(Sig, ld, M, b, t1, t1 + 1, [](int a) { return a; }, zero4, av, Tn0, Tn0, l15, l4);
            }
        }
        __syncthreads();
        inv_write_panel(B, Sig, ld, M, Mp, k0, Tn0, nD);
        // ---- block k+1: pivot sweeps (on the tile just updated), panel product (reads block k's new pivot column)
        PHX_BEGIN(t_piv2);
        if (inv_pivot(B.tid, Sig, ld, M, k1, nD, nD2)) return 1;
        PHX_END(t_piv2, PH_INV_PIVOT);
        PHX_BEGIN(t_tn2);
        inv_panel<false>(B, Sig, ld, M, nT, k1, nD, Tn1, Pk, Mp);
        __syncthreads();
        PHX_END(t_tn2, PH_INV_TN);
        // ---- every other tile, one trip: the tiles with neither index in {tk, t1} take both updates, the tiles of
        // block k's pivot column / pivot block (tile index tk) block k+1's only
        {
            const int n2 = nT - 2;
            const auto skip2 = [tk](int c) { return c < tk ? c : c + 2; };
            int cnt = 0;
            for (int b = 0; b < n2; b++) {
                const int tj = skip2(b);
                for (int a0 = b; a0 < n2; a0 += INV_RUN) {
                    if ((cnt++) % B.nwave != B.wave) continue;
                    double av0[4], av1[4];
                    inv_av_pk(Pk, Mp, tj, l15, l4, av0);
                    inv_av_sig(Sig, ld, M, tj, k1, l15, l4, av1);
                    inv_run<2>(Sig, ld, M, tj, a0, a0 + INV_RUN < n2 ? a0 + INV_RUN : n2, skip2, av0, av1, Tn0, Tn1, l15, l4);
                }
            }
            // tile column tk: rows tk, tk+2, ...
            const auto rows_k = [tk](int c) { return c == 0 ? tk : tk + 1 + c; };
            const int nk = nT - tk - 1;                          // tk itself + the tiles below t1
            for (int a0 = 0; a0 < nk; a0 += INV_RUN) {
                if ((cnt++) % B.nwave != B.wave) continue;
                double av1[4];
                inv_av_sig(Sig, ld, M, tk, k1, l15, l4, av1);
                inv_run<1>(Sig, ld, M, tk, a0, a0 + INV_RUN < nk ? a0 + INV_RUN : nk, rows_k, zero4, av1, Tn1, Tn1, l15, l4);
            }
            // tile row tk: columns b < tk
            for (int b = 0; b < tk; b++) {
                if ((cnt++) % B.nwave != B.wave) continue;
                double av1[4];
                inv_av_sig(Sig, ld, M, b, k1, l15, l4, av1);
                inv_run<1>(Sig, ld, M, b, tk, tk + 1, [](int a) { return a; }, zero4, av1, Tn1, Tn1, l15, l4);
            }
        }
        __syncthreads();
        inv_write_panel(B, Sig, ld, M, Mp, k1, Tn1, nD);
    }
    if (tk < nT) {                                               // an odd last block on its own
        const int k0 = tk * 16;
        if (inv_pivot(B.tid, Sig, ld, M, k0, nD, nD2)) return 1;
        inv_panel<false>(B, Sig, ld, M, nT, k0, nD, Tn0, Pk, Mp);
        __syncthreads();
        const int n1 = nT - 1;                                   // the last block: the other tiles are 0 .. n1-1
        int cnt = 0;
        for (int b = 0; b < n1; b++)
            for (int a0 = b; a0 < n1; a0 += INV_RUN) {
                if ((cnt++) % B.nwave != B.wave) continue;
                double av[4];
                inv_av_sig(Sig, ld, M, b, k0, l15, l4, av);
                inv_run<1>(Sig, ld, M, b, a0, a0 + INV_RUN < n1 ? a0 + INV_RUN : n1, [](int a) { return a; }, zero4, av, Tn0, Tn0, l15, l4);
            }
        __syncthreads();
        inv_write_panel(B, Sig, ld, M, Mp, k0, Tn0, nD);
    }
    __syncthreads();
    for (int j = B.wave; j < M; j += B.nwave)
        for (int i = j + B.lane; i < M; i += 64) {
            const double v = -Sig[(size_t)j * ld + i];
            Sig[(size_t)j * ld + i] = v;
            Sig[(size_t)i * ld + j] = v;
        }
    __syncthreads();
    return 0;
}

// ---- the whole triangle in registers ---------------------------------------------------------------------------
// At the sizes most inversions of a grid have (a few hundred columns at most) the blocked sweep above is bound by
// neither the matrix cores nor bandwidth but by memory round trips: every pivot block costs a chain of them (pivot
// tile in, panel in, each wave's tiles in and out, panel out) at 1 - 2 us each with 256 workgroups on the chip, ~20 us
// per block at M = 133 against 1 us of matrix-pipe time (tools/ubench/inverse_rate.py).  Here the matrix makes ONE
// trip: tile q = a (a + 1) / 2 + b of the lower triangle (row tile a >= column tile b) belongs to wave q mod nwave
// for the whole inversion and lives in that wave's registers as the accumulator of its matrix ops (S tiles per wave,
// statically indexed: the slot loops are unrolled, the branches inside are wave-uniform).  Per pivot block the
// owners put the pivot column (P, in operand order) and the pivot tile into LDS, the pivot tile is swept there, the
// panel product goes LDS -> matrix cores -> LDS, and every wave updates its own tiles from LDS operands; the tiles of
// the pivot column / pivot tile take their new values from the panel / the swept tile.  Element by element these
// are the operations of gm_spd_inverse_blocked in the same order -- the same bits -- with no memory access inside
// the loop.  Up to S * nwave tiles: S = 24 -> 19 row tiles, M <= 304.
#define INV_PP 17          // pitch of a pivot-column row group in P: [row tile][pivot s][row in tile]
template <int S>
DEVNI int gm_spd_inverse_regs(const Blk &NOALIAS B, const GmWork &NOALIAS W, int M, long long *phx)
{
    const int ld = W.ld;
    const gptr_d Sig = as_global_rw(W.Sig);
    const int nT = (M + 15) >> 4, Mp = nT * 16, nQ = nT * (nT + 1) / 2;
    // B is a reference argument of a non-inlined function, i.e. memory: every use inside a loop would be a flat load
    // (with its wait) -- register copies, taken once
    const int tid = B.tid, lane = tid & 63;
    const int wave = uni(B.wave), nw = uni(B.nwave);
    const int l15 = lane & 15, l4 = lane >> 4;
    double *const pool = uni_ptr(B.pool);
    const lptr_d Tn = as_lds(pool), P = Tn + (size_t)Mp * INV_TP, nD = P + (size_t)nT * 16 * INV_PP, nD2 = nD + 16 * 17;
    // (a << 8 | b) of slot s in a scalar register, -1 = no tile.  Inside the loops the value is passed through an empty
    // asm so that everything derived from it (LDS addresses, guards) is recomputed where it is used: hoisted out of
    // the pivot loop those would take ~20 vector registers per slot and push the tiles themselves into scratch.
    int tab[S];
    d4 acc[S];
    PHD_BEGIN(t_ld);
#define INV_SLOT(s) int ab_ = tab[s]; asm volatile("" : "+s"(ab_)); const int a_ = ab_ >> 8, b_ = ab_ & 255
#pragma unroll
    for (int s = 0; s < S; s++) {                               // this wave's tiles
        const int q = s * nw + wave;
        int a = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
        while (a * (a + 1) / 2 > q) a--;
        while ((a + 1) * (a + 2) / 2 <= q) a++;
        const bool on = q < nQ;
        tab[s] = uni(on ? (a << 8) | (q - a * (a + 1) / 2) : -1);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int icol = a * 16 + l15, j = (q - a * (a + 1) / 2) * 16 + l4 + 4 * r;
            acc[s][r] = (on && icol < M && j < M) ? Sig[(size_t)j * ld + icol] : 0.0;
        }
    }
    PHD_END(t_ld, 16);
    for (int tk = 0; tk < nT; tk++) {
        const int k0 = tk * 16;
        // slots holding a tile of the pivot row / column (scalar, branch-free: the slot loops below then test one bit per
        // slot instead of walking a chain of compares and branches -- at these sizes branch latency is what a slot costs)
        unsigned mk = 0, mv = 0;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int ab = tab[s];
            mv |= (ab >= 0 ? 1u : 0u) << s;
            mk |= ((ab >= 0 && ((ab >> 8) == tk || (ab & 255) == tk)) ? 1u : 0u) << s;
        }
        PHX_BEGIN(t_piv);
        __syncthreads();                                         // the previous block's readers of P / Tn / nD are done
        PHD_BEGIN(t_ex);
        // ---- pivot column and pivot tile -> LDS
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (!((mk >> s) & 1u)) continue;
            INV_SLOT(s);
            if (a_ == tk && b_ == tk) {                           // lanes of the stored triangle write both mirror images, identity-padded
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int rr = l15, cc = l4 + 4 * r;          // icol = k0 + rr >= j = k0 + cc
                    if (rr >= cc) {
                        const bool in = k0 + rr < M;              // then k0 + cc < M too
                        const double v = in ? acc[s][r] : (rr == cc ? 1.0 : 0.0);
                        nD[rr * 17 + cc] = v; nD[cc * 17 + rr] = v;
                    }
                }
            } else if (b_ == tk) {                                // tile (a, tk): A(icol, k0 + l4 + 4r)
#pragma unroll
                for (int r = 0; r < 4; r++) P[(a_ * 16 + l4 + 4 * r) * INV_PP + l15] = acc[s][r];
            } else if (a_ == tk) {                                // tile (tk, b): A(j, k0 + l15) by symmetry
#pragma unroll
                for (int r = 0; r < 4; r++) P[(b_ * 16 + l15) * INV_PP + l4 + 4 * r] = acc[s][r];
            }
        }
        __syncthreads();
        PHD_END(t_ex, 17);
        // ---- the sixteen sweeps inside the pivot tile: one division per element (numerator chosen first); a pivot that is
        // not positive is noticed by every thread alike and acted upon after the sweeps (what follows it is never used)
        bool bad = false;
        {
            const int r = tid & 15, c = (tid >> 4) & 15;
#pragma unroll
            for (int sw = 0; sw < 16; sw++) {
                const lptr_d src = (sw & 1) ? nD2 : nD, dst = (sw & 1) ? nD : nD2;
                const double d = src[sw * 17 + sw], prs = src[r * 17 + sw], psc = src[sw * 17 + c], v = src[r * 17 + c];
                bad |= !(d > 0);
                const bool rs = r == sw, cs = c == sw;
                const double num = rs ? (cs ? -1.0 : psc) : (cs ? prs : prs * psc);
                const double t = num / d;
                if (tid < 256) dst[r * 17 + c] = (rs || cs) ? t : v - t;
                __syncthreads();
            }
        }
        if (bad) return 1;
        PHX_END(t_piv, PH_INV_PIVOT);
        PHX_BEGIN(t_tn);
        // ---- panel product Tn = (pivot column) x (swept tile), rows outside the block
        for (int ti = wave; ti < nT; ti += nw) {
            const int i = ti * 16 + l15;
            const bool live = i < M && ti != tk;
            d4 t4 = d4{0, 0, 0, 0};
            double av[4], bw[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int kc = k0 + kk * 4 + l4;
                const double v = P[(size_t)(ti * 16 + kk * 4 + l4) * INV_PP + l15];
                av[kk] = (live && kc < M) ? v : 0.0;
                bw[kk] = nD[(kk * 4 + l4) * 17 + l15];
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) t4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bw[kk], t4, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) Tn[(size_t)(ti * 16 + l4 + 4 * r) * INV_TP + l15] = t4[r];
        }
        __syncthreads();
        PHX_END(t_tn, PH_INV_TN);
        PHD_BEGIN(t_tr);
        // ---- every wave's own tiles (the branches are wave-uniform)
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (!((mv >> s) & 1u)) continue;
            INV_SLOT(s);
            if (!((mk >> s) & 1u)) {                              // A += Tn[a] * (pivot column rows of b)'
                double av[4], bv[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int kc = k0 + kk * 4 + l4, jrow = b_ * 16 + l15;
                    const double v = P[(b_ * 16 + kk * 4 + l4) * INV_PP + l15];
                    av[kk] = (jrow < M && kc < M) ? v : 0.0;
                    bv[kk] = Tn[(a_ * 16 + l15) * INV_TP + kk * 4 + l4];
                }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc[s], 0, 0, 0);
            } else if (a_ == tk && b_ == tk) {                    // pivot tile <- swept tile
#pragma unroll
                for (int r = 0; r < 4; r++) if (k0 + l15 < M && k0 + l4 + 4 * r < M) acc[s][r] = nD[l15 * 17 + l4 + 4 * r];
            } else if (b_ == tk) {                                // pivot column below the block <- -Tn
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = a_ * 16 + l15, kc = k0 + l4 + 4 * r;
                    if (i < M && kc < M) acc[s][r] = -Tn[i * INV_TP + l4 + 4 * r];
                }
            } else {                                              // ... and left of it (row tile tk, column tile b)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = b_ * 16 + l4 + 4 * r, kc = k0 + l15;
                    if (i < M && kc < M) acc[s][r] = -Tn[i * INV_TP + l15];
                }
            }
        }
        PHD_END(t_tr, 20);
    }
    PHD_BEGIN(t_st);
    // ---- out: negated, both mirror images (the diagonal tiles only from their stored triangle)
#pragma unroll
    for (int s = 0; s < S; s++) {
        INV_SLOT(s);
        if (ab_ < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int icol = a_ * 16 + l15, j = b_ * 16 + l4 + 4 * r;
            if (icol < M && j < M && icol >= j) {
                const double v = -acc[s][r];
                Sig[(size_t)j * ld + icol] = v;
                Sig[(size_t)icol * ld + j] = v;
            }
        }
    }
    __syncthreads();
    PHD_END(t_st, 21);
    return 0;
#undef INV_SLOT
}
// LDS of the register form: Tn (Mp x 18) + P (Mp x 17) + two pivot tiles
DEV bool inv_regs_fits(const Blk &NOALIAS B, int M, int slots)
{
    const int nT = (M + 15) >> 4, Mp = nT * 16;
    return nT * (nT + 1) / 2 <= slots * B.nwave && Mp * (INV_TP + INV_PP) + 2 * 16 * 17 <= B.pool_n;
}

DEV int gm_spd_inverse(const Blk &NOALIAS B, const GmWork &NOALIAS W, int M, long long *phx = nullptr, int pair = 3)
{
    (void)phx;
    const int Mp = ((M + 15) >> 4) * 16;
    if (pair & 2) {                                              // bit 1: the register form while the triangle fits
        if (M > 16 && inv_regs_fits(B, M, 4)) return gm_spd_inverse_regs<4>(B, W, M, phx);
        if (M > 16 && inv_regs_fits(B, M, 8)) return gm_spd_inverse_regs<8>(B, W, M, phx);
        if (M > 16 && inv_regs_fits(B, M, 12)) return gm_spd_inverse_regs<12>(B, W, M, phx);
        if (M > 16 && inv_regs_fits(B, M, 16)) return gm_spd_inverse_regs<16>(B, W, M, phx);
        if (M > 16 && inv_regs_fits(B, M, 20)) return gm_spd_inverse_regs<20>(B, W, M, phx);
        if (M > 16 && inv_regs_fits(B, M, 22)) return gm_spd_inverse_regs<22>(B, W, M, phx);
    }
    if ((pair & 1) && M > 32 && W.Tn && 2 * Mp * INV_TP + 2 * 16 * 17 <= B.pool_n) return gm_spd_inverse_paired(B, W, M, phx);
    if (M > 16 && Mp * INV_TP + 2 * 16 * 17 <= B.pool_n) {
        const lptr_d Tn = as_lds(B.pool);
        return gm_spd_inverse_blocked(B, W, M, phx, Tn, Tn + (size_t)Mp * INV_TP);
    }
    if (M > 16 && W.Tn) return gm_spd_inverse_blocked(B, W, M, phx, as_global_rw(W.Tn), as_lds(B.pool));
    return gm_spd_inverse_scalar(B, W, M);
}

// H = beta G[used, used] + diag(A) into W.H and W.Sig (MainEff.c:1841-1876)
DEV void gm_hessian_build(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int K, GmScalars &NOALIAS S)
{
    const int M = S.M, ld = W.ld;
    const double beta = S.beta;
    {   // feature ids, Gram row ids and A staged in LDS: the M^2 gathers then depend on nothing but LDS
        PH_BEGIN();
        int *lu = (int *)B.pool, *lr = lu + M;
        double *la = B.pool + M + 1;
        blk_sync(B);
        PAR(l, M) { lu[l] = W.used[l]; lr[l] = W.rowid[l]; la[l] = W.A[l]; }
        blk_sync(B);
        const gptr_cd G = as_global(F.G);
        const gptr_d H = as_global_rw(W.H), Sg = as_global_rw(W.Sig), Gc = as_global_rw(W.Gc);
        const bool cached = S.gc_ok != 0;
        // the blocked inverse reads and writes only the triangle [j][i >= j] (and mirrors Sigma itself at the end);
        // the scattered mirror stores are needed only in front of the scalar inverse (M <= 16, or no LDS room)
        const bool mirror = !(M > 16 && (((M + 15) >> 4) * 16 * INV_TP + 2 * 16 * 17 <= B.pool_n || W.Tn));
        for (int j = B.wave; j < M; j += B.nwave) {
            const size_t rj = (size_t)lr[j] * K;
            // Phi_i.Phi_j from the Gram matrix: element (j, i >= j) is G[row_j][used_i] -- gathered from Gram row j
            // (one row per wave trip, not a column walk across M rows) the first time and after a delete has
            // reshuffled the slots, otherwise taken from the fit's own copy of that block, which every add extends
            // by one column (gm_add, gm_add_batch): a coalesced read instead of M^2/2 gathers.  One triangle stands for
            // both, so H is exactly symmetric.  Four loads are issued before the first store so that they overlap.
            for (int i0 = j + B.lane; i0 < M; i0 += 4 * BLK_LANES) {
                double h[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int i = i0 + c * BLK_LANES;
                    h[c] = i < M ? (cached ? Gc[(size_t)j * ld + i] : G[rj + lu[i]]) : 0.0;
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int i = i0 + c * BLK_LANES;
                    if (i < M) {
                        if (!cached) Gc[(size_t)j * ld + i] = h[c];
                        double v = h[c] * beta;
                        if (i == j) v += la[i];
                        H[(size_t)j * ld + i] = v;
                        Sg[(size_t)j * ld + i] = v;
                        if (mirror && i != j) { H[(size_t)i * ld + j] = v; Sg[(size_t)i * ld + j] = v; }
                    }
                }
            }
        }
        blk_sync(B);
        S.gc_ok = 1;
        PH_END(PH_HBUILD);
    }
}

// blocked inverse: per 16-pivot step one panel product per row tile + the triangle of trailing tiles
DEV void gm_inverse_count(const Blk &NOALIAS B, GmScalars &NOALIAS S, int M)
{
    if (M > 16) CNT(const int64_t nT = (M + 15) >> 4; c.mfma_tiles += nT * (nT + (nT - 1) * nT / 2));
}
// mu = beta Sigma v1 (v1 = Phi't), one fma chain per row over the columns in order
DEV void gm_mu_update(const Blk &NOALIAS B, const GmWork &NOALIAS W, int M, double beta)
{
    const int ld = W.ld;
    {   // the vector from LDS, Sigma through a global pointer: the row loads of a thread pipeline
        const lptr_d lv = as_lds(B.pool);
        PAR(j, M) lv[j] = W.v1[j];
        blk_sync(B);
        const gptr_cd Sg = as_global(W.Sig);
        PAR(i, M) {
            double a = 0;
#pragma unroll 8
            for (int j = 0; j < M; j++) a += lv[j] * Sg[(size_t)j * ld + i];
            W.mu[i] = a * beta;
        }
    }
}

// out_h = sum_j vec[j] * (X[used[j]][h] * rscale[used[j]])  for one sample h, the model columns in order.
// On the device the (column id, coefficient, 1/|x|) triples are staged in LDS first (gm_stage_model), so
// the only memory access per term is the coalesced design-column load and nothing sits behind a dependent
// used[] -> rscale[] address chain; the arithmetic is the same expression in the same order either way.
DEV void gm_stage_model(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &NOALIAS W, int M, const double *vec)
{
    double *lc = B.pool, *lr = B.pool + M;
    int *lu = (int *)(B.pool + 2 * M);
    blk_sync(B);
    PAR(j, M) { const int uj = W.used[j]; lu[j] = uj; lc[j] = vec[j]; lr[j] = F.rscale[uj]; }
    blk_sync(B);
}
DEV double gm_model_at(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &, int M, const double *, int N, int h)
{
    const lptr_d lc = as_lds(B.pool), lr = as_lds(B.pool + M);
    const lptr_i lu = as_lds((int *)(B.pool + 2 * M));
    const gptr_cd X = as_global(F.X);
    double v = 0;
#pragma unroll 4
    for (int j = 0; j < M; j++) v += lc[j] * (X[(size_t)lu[j] * N + h] * lr[j]);
    return v;
}
// two samples at once (twice the design-column loads in flight; each sum is the same chain as above)
DEV void gm_model_at2(const Blk &NOALIAS B, const FoldDev &NOALIAS F, const GmWork &, int M, const double *, int N, int h0, int h1, double &v0, double &v1)
{
    const lptr_d lc = as_lds(B.pool), lr = as_lds(B.pool + M);
    const lptr_i lu = as_lds((int *)(B.pool + 2 * M));
    const gptr_cd X = as_global(F.X);
    double a0 = 0, a1 = 0;
#pragma unroll 8
    for (int j = 0; j < M; j++) {
        const size_t col = (size_t)lu[j] * N;
        const double c = lc[j], r = lr[j];
        a0 += c * (X[col + h0] * r);
        a1 += c * (X[col + h1] * r);
    }
    v0 = a0; v1 = a1;
}

// XOR over the workgroup (decision trace: order-free hash of bit patterns)
DEV unsigned long long blk_xor64(const Blk &NOALIAS B, unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
    unsigned long long *r = (unsigned long long *)B.red;
    __syncthreads();
    if (B.lane == 0) r[B.wave] = v;
    __syncthreads();
    v = 0;
    for (int w = 0; w < B.nwave; w++) v ^= r[w];
    __syncthreads();
    return v;
}

