"""Per-phase shares of gm_cv_kernel on a stored real-R table's design (default: the Subset_Test table, 3803 x 5356, active
sets up to 1446 columns) from the -DPAREBEN_PHASE_TIMERS build; prints the launch totals and the longest fits."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "pareben_amd", "lib", "libpareben_hip_prof.so")
import pareben_amd
from pareben_amd.grid import BuildGrid, AssignToFolds
name = sys.argv[1] if len(sys.argv) > 1 else "subset5356"
d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")); n = int(d["n"]); k = int(d["drop_first_row"])
X = np.asfortranarray(np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64)[k:] * 2 - 1); y = d["pheno"].astype(np.float64)[k:]
if len(sys.argv) > 2:
    X = np.asfortranarray(X[:, :int(sys.argv[2])])
alpha, lam = BuildGrid(X, y, 3)
fid = AssignToFolds(X, 3, sample_kind="Rounding")
path = os.path.join(tempfile.gettempdir(), "pareben_phase.bin")
os.environ["PAREBEN_PHASE_DUMP"] = path
with pareben_amd.Context(X, y, fid, 3) as ctx:
    E, st, cnt = ctx.run(alpha, lam)
    print("timing", ctx.last_timing(), ctx.launch_info())
ph = np.fromfile(path, dtype=np.int64).reshape(-1, 24).astype(np.float64)
names = ["fullstat_features", "fullstat_rest", "delta_ml+collect", "actions", "noise", "spd_inverse", "action_ksweep", "total",
         "act_matvec", "act_rank1", "act_refresh", "h_build", "mu_after_inv", "batch_track", "inv_pivot", "inv_tn"]
tot = ph[:, 7].sum()
print("sum of per-fit wall: %.1f s over %d fits" % (tot / 1e8, len(ph)))
for i, nm in enumerate(names):
    if i != 7:
        print("  %-20s %6.2f %%" % (nm, 100 * ph[:, i].sum() / tot))
cc = cnt.reshape(-1, cnt.shape[-1])
for u in np.argsort(-ph[:, 7])[:8]:
    t = ph[u, 7]
    print("  fit %d (cell %d fold %d, alpha %.2f): %.2f s  m_max %d inner %d outer %d fullstat %d | " % (u, u // 3, u % 3 + 1, alpha[u // 3], t / 1e8, cc[u, 10], cc[u, 1], cc[u, 0], cc[u, 5])
          + " ".join("%s=%.0f%%" % (names[i][:10], 100 * ph[u, i] / t) for i in (0, 2, 3, 4, 5, 6, 9, 11, 14, 15)))
