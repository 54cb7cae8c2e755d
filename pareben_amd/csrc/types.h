// types.h -- plain structs shared by the host API and the device code.
#pragma once
#include <stdint.h>

// Per-fold read-only data in HBM.  All matrices are column-major with the training (or test)
// rows of that fold compacted in their original order, so every column is one contiguous,
// coalesced run.
struct FoldDev {
    const double *X;        // N   x K  training design
    const double *y;        // N        training target
    const double *Xte;      // nte x K  held-out design
    const double *yte;      // nte
    const double *scale;    // K  |x_i| (1 where the column is all zero)     MainEff.c:87-99
    const double *rscale;   // K  1/scale
    const double *bt0;      // K  x_i.y / scale_i
    const double *cs;       // K  x_i.1 / scale_i
    const double *G;        // normalised Gram rows, G[r*K+i] = x_i.(x_u/scale_u)/scale_i: all K rows
                            // (r = u), or in lazy mode a pool of pool_rows rows filled on demand
    int *slot_of;           // lazy mode: K ints, row id of feature u (-1 absent, -2 in flight)
    int *pool_next;         // lazy mode: slots handed out so far from this fold's pool
    int pool_base;          // lazy mode: row id of this fold's first pool row (G is common to all folds)
    int pool_rows, lazy;
    double ymean;           // sum(y)/N                                       MainEff.c:145-147
    double varY;            // unbiased variance of y                         MainEff.c:152
    int N, nte;
    const double *Xt;       // strict-order mode and binomial CV contexts (else null): the training design once more, sample-major
                            // (N x K, Xt[h*K + i]), so that lanes holding consecutive features read whole lines (gm_strict.h; bm_dev.h: the
                            // matrix-core operand of the weighted-rows pass)
    int n_main;             // columns 0 .. n_main-1 are main effects: PHI = x * (1/scale) (MainEff.c:1062-1065, :517-520); the pair
                            // columns behind them (epistasis) are formed by division, PHI = x / scale (Full2.c:544, :913)
};

// The two Gaussian reference kernels share their skeleton; these are the constants and rules in
// which elasticNetLinearNeFull2.c (epis = 1) differs from elasticNetLinearNeMainEff.c (epis = 0).
struct GmVariant {
    int epis = 0;              // selects the priority / initial-beta rules
    double n_add = 0.9;        // block cut-off factor        MainEff.c:275  0.9   | Full2.c:293  0.99
    double ml_delta = 1e-3;    // minimum dML                 MainEff.c:277  1e-3  | Full2.c:295  1e-2
    double reest_tol = 1e-3;   // |dlog alpha| termination    MainEff.c:543  1e-3  | Full2.c:558  0.1
    double alpha_max = 1e2;    // initial precision clamp     MainEff.c:995  1e2   | Full2.c:849  1e3
    double b_eps = 1e-10;      // intercept denominator guard MainEff.c:188  1e-10 | Full2.c:202  none
};
// The binomial kernels (bm_fit.h) read only `epis` (main effects: NEmainEff.c rules, epistasis: NeFull.c rules) and set n_add themselves.

// Per-fit event counters (SURVEY.md 8(d) accounting).
struct FitCounters {
    int64_t n_outer, n_inner, n_add, n_del, n_reest, n_fullstat;
    int64_t sum_m_action, sum_m_full, sum_m2_full, m_final, m_max, status;
    int64_t mfma_tiles;   // 16 x 16 x 16 tile products (4 x v_mfma_f64_16x16x4_f64, 8192 flop) the fit's matrix-core passes execute
    int64_t sum_m_swept;  // Gram rows the action sweeps actually read (sum_m_action minus the sweeps a following full-stat pass made unnecessary)
};
#define PAREBEN_NCOUNTERS 14

enum {
    ST_OVERFLOW = 1,      // active set exceeded the reference's basisMax (and, with ST_ABORT, the workspace capacity)
    ST_CHOLESKY = 2,      // Hessian not positive definite
    ST_STALE = 4,         // reference's stale-index delete path taken (SURVEY.md hard parts)
    ST_ABORT = 8          // fit stopped early (a state the reference leaves undefined)
};

// Decision trace (diagnostics; pareben_set_trace, tools/trace_divergence.py): one record of TR_NSLOT 64-bit words per
// inner iteration -- what was decided, by what margin, and order-free XOR hashes of the state afterwards.  The oracle
// writes the same layout (oracle/eben_gm.c), so two builds can be compared record by record.  buf[0] = records
// written; records start at buf + TR_NSLOT.
enum { TR_ITER, TR_IITER, TR_MBEFORE, TR_NU, TR_ACT, TR_NTODO, TR_SEL, TR_MAFTER,
       TR_BEST, TR_SECOND, TR_CUTOFF, TR_NEAREST, TR_BETA, TR_HSIN, TR_HQIN, TR_HSIG, TR_NSLOT };

#define ADD_TB 16         // consecutive ADD actions of one block update applied with ONE sweep of the Gram rows

// Per-workgroup scratch in HBM (one slot per resident workgroup).
struct GmWork {
    double *Sin, *Qin, *Sout, *Qout, *dml, *aroot, *bt;   // K each
    int *upos, *todo;                                      // K each
    signed char *act;                                      // K
    double *Sig, *H;                                       // cap x cap, column-major, ld = cap
    double *Gc;        // device only: Gram block of the active set, Gc[j][i] = G[row_j][used_i] for slots j <= i (gm_final_update)
    double *Tn = nullptr;   // device only: ld x 18 scratch for the blocked inverse's pivot-column panel when it does not fit in LDS (M > 1040)
    double *A, *mu, *gam, *v1, *v2, *v3, *v4;              // cap+1 each
    double *vb, *bsc;                                      // batched adds: ADD_TB vectors of cap+2, 4*ADD_TB scalars
    int *used;                                             // cap+1  feature of each active slot
    int *rowid;                                            // cap+1  Gram row id of each active slot
    int *pfree;                                            // lazy mode: [0] = n free, [1..] free private row ids
    int priv_base, priv_rows;                              // this workgroup's private rows (pool exhausted)
    double *e;                                             // max(N) scratch
    int cap, ld;
    int cap_flag;      // the reference's basisMax: an active set growing past it is flagged (ST_OVERFLOW), the fit goes on up to cap
};

// One feature-parallel phase of a fit offered to idle workgroups (gm_fit.h, "shared phases").  `word` packs
// (epoch << 32) | next_tile: odd epoch = open; tiles are claimed by compare-and-swap on the whole word,
// so a claim always belongs to the pass that was open when it was made.
struct FsJob {
    unsigned long long word;
    int done;              // chunks finished (by anyone)
    int fold, M, n_tiles;  // n_tiles: feature tiles of the job (256 features for a full-stat pass, 128 for a sweep)
    int kind, mode, rid;   // JOB_FULLSTAT | JOB_SQ; for JOB_SQ the update mode and the new feature's Gram row id
    int pad;
    double beta, c1, c2;   // one 64-byte line per job
};
enum { JOB_FULLSTAT = 0, JOB_SQ = 1, JOB_SQB = 2 };   // JOB_SQB: the sweep of a run of adds (M = size before the run, mode = run length)
struct FsShare {           // null jobs = sharing off
    FsJob *jobs;           // one per workgroup of the launch
    int *active;           // workgroups that currently own a fit
    const int *queue;      // work-queue head (>= n_units: drained)
    int n_units, n_blocks, self;
    char *ws;              // workspace base / stride of the launch: a helper carves the owner's slot
    size_t ws_stride, offK, offSig, offM;
    const FoldDev *folds;
    int cap;
    int early;             // few fits per workgroup: phases are shared from the start, not only once the queue is drained
    int heavy_m;           // fits whose active set reaches this size share their phases from then on
};
