"""Per-phase share of gm_cv_kernel on the config-4 workload (yeast n=200, k=300, Epis=yes: 45 150 columns) from the
-DPAREBEN_PHASE_TIMERS build (libpareben_hip_prof.so)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "pareben_amd", "lib", "libpareben_hip_prof.so")
import pareben_amd
from pareben_amd.grid import BuildGrid, AssignToFolds
d = np.load(os.path.join(ROOT, "tests", "golden", "yeast_timing_200x600.npz"))
n = int(d["n"]); B = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64) * 2 - 1
X = np.asfortranarray(B[:, :300]); y = d["y"].astype(np.float64)
alpha, lam = BuildGrid(X, y, 5, "yes", device=0); fid = AssignToFolds(X, 5)
path = os.path.join(tempfile.gettempdir(), "pareben_phase_c4.bin")
os.environ["PAREBEN_PHASE_DUMP"] = path
with pareben_amd.Context(X, y, fid, 5, epis=True) as ctx:
    E, st, cnt = ctx.run(alpha, lam)
    print("timing", ctx.last_timing(), ctx.launch_info())
ph = np.fromfile(path, dtype=np.int64).reshape(-1, 24).astype(np.float64)
tot = ph[:, 7].sum()
names = ["fullstat_features", "fullstat_rest", "delta_ml+collect", "actions", "noise", "spd_inverse", "action_ksweep", "total",
         "act_matvec", "act_rank1", "act_refresh", "h_build", "mu_after_inv", "batch_track", "inv_pivot", "inv_tn"]
print("sum of per-fit ticks %.1f s over %d fits; longest %.3f s" % (tot / 1e8, len(ph), ph[:, 7].max() / 1e8))
for k in (0, 2, 3, 6, 8, 9, 10, 13, 4, 5, 11, 12):
    print("  %-22s %6.2f %%" % (names[k], 100 * ph[:, k].sum() / tot))
print("  %-22s %6.2f %%" % ("other", 100 * (tot - ph[:, [0, 2, 3, 4, 5, 11, 12]].sum()) / tot))
c = cnt.reshape(-1, cnt.shape[-1]).astype(float)
print("per fit: outer %.1f inner %.1f adds %.1f dels %.1f reest %.1f fullstats %.1f m_max %.0f (max %d)" % (*[c[:, k].mean() for k in (0, 1, 2, 3, 4, 5, 10)], c[:, 10].max()))
