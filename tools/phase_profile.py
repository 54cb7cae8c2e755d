"""Diagnostic: per-phase share of the fit kernel's time, from the -DPAREBEN_PHASE_TIMERS build
(libpareben_hip_prof.so).  Usage: python tools/phase_profile.py [--p 10000 --nlambda 10]"""
import argparse, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "pareben_amd", "lib", os.environ.get("PAREBEN_PROF_LIB", "libpareben_hip_prof.so"))
import pareben_amd
from pareben_amd.grid import BuildGrid, AssignToFolds
from pareben_amd.synth import synthetic_gaussian

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000); ap.add_argument("--p", type=int, default=10000)
ap.add_argument("--nfolds", type=int, default=5); ap.add_argument("--nalpha", type=int, default=20)
ap.add_argument("--nlambda", type=int, default=10)
ap.add_argument("--world", type=int, default=1); ap.add_argument("--rank", type=int, default=0)   # one rank's share of the grid
a = ap.parse_args()
X, y, _, _ = synthetic_gaussian(a.n, a.p)
alpha, lam = BuildGrid(X, y, a.nfolds, nAlpha=a.nalpha, nLambda=a.nlambda)
fid = AssignToFolds(X, a.nfolds)
if a.world > 1:
    from pareben_amd.dist import shard_cells
    mine = shard_cells(alpha, lam, a.rank, a.world)
    alpha, lam = alpha[mine], lam[mine]
path = os.path.join(tempfile.gettempdir(), "pareben_phase.bin")
os.environ["PAREBEN_PHASE_DUMP"] = path
with pareben_amd.Context(X, y, fid, a.nfolds) as ctx:
    E, st, cnt = ctx.run(alpha, lam)
    print("timing", ctx.last_timing(), ctx.launch_info())
ph = np.fromfile(path, dtype=np.int64).reshape(-1, 24).astype(np.float64)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "phase_dump.npz"), ph=ph, cnt=cnt, alpha=alpha, lam=lam)
names = ["fullstat_features", "fullstat_rest(incl in total only)", "delta_ml+collect", "actions", "noise", "spd_inverse", "action_ksweep", "total",
         "act_matvec", "act_rank1", "act_refresh", "h_build", "mu_after_inv", "batch_track", "inv_pivot", "inv_tn"]
tot = ph[:, 7].sum()
print("sum of per-fit wall ticks (100 MHz): %.3f s over %d fits" % (tot / 1e8, len(ph)))
cc = cnt.reshape(-1, cnt.shape[-1]).astype(np.float64)
print("aggregate full-stat rate per CU: %.1f GFLOP/s (peak 307); action Gram-row rate per CU: %.1f GB/s" % (
    2.0 * a.p * cc[:, 8].sum() / (ph[:, 0].sum() / 1e8) / 1e9, 8.0 * a.p * cc[:, 6].sum() / (ph[:, 3].sum() / 1e8) / 1e9))
print("shader clock during the full-stat pass: %.0f MHz" % (ph[:, 1].sum() / ph[:, 0].sum() * 100.0))
for k in (0, 2, 3, 6, 8, 9, 10, 13, 4, 5, 14, 15, 11, 12):
    print("  %-22s %6.2f %%" % (names[k], 100 * ph[:, k].sum() / tot))
print("  %-22s %6.2f %%" % ("other", 100 * (tot - ph[:, [0, 2, 3, 4, 5, 11, 12]].sum()) / tot))
print("  shared jobs: full-stat chunks %d (owner ran %.0f %%, waited %.2f s); sweep chunks %d (owner ran %.0f %%, waited %.2f s)" % (
    ph[:, 17].sum(), 100 * ph[:, 16].sum() / max(ph[:, 17].sum(), 1), ph[:, 20].sum() / 1e8,
    ph[:, 19].sum(), 100 * ph[:, 18].sum() / max(ph[:, 19].sum(), 1), ph[:, 21].sum() / 1e8))
heavy = np.argsort(ph[:, 7])[-5:]
for u in heavy:
    c = cnt.reshape(-1, cnt.shape[-1])[u]
    fs_s = ph[u, 0] / 1e8
    print("  fit %d: %.3f s  M=%d inner=%d adds=%d fullstat=%d sumM2=%.3g fs_rate=%.1f GF/s act_GBs=%.1f | " % (u, ph[u, 7] / 1e8, c[9], c[1], c[2], c[5], c[8], 2.0 * a.p * c[8] / fs_s / 1e9, 8.0 * a.p * c[6] / (ph[u, 3] / 1e8) / 1e9),
          "fs_own=%.0f%% sq_own=%.0f%% fs_wait=%.2fs sq_wait=%.2fs " % (100 * ph[u, 16] / max(ph[u, 17], 1), 100 * ph[u, 18] / max(ph[u, 19], 1), ph[u, 20] / 1e8, ph[u, 21] / 1e8) +
          " ".join("%s=%.0f%%" % (names[k][:8], 100 * ph[u, k] / ph[u, 7]) for k in (0, 2, 3, 6, 8, 9, 10, 13, 4, 5, 14, 15, 11, 12)))
