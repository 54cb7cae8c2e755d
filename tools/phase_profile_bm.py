"""Diagnostic: per-phase share of the binomial fit kernel (config 3) from the -DPAREBEN_PHASE_TIMERS build."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "pareben_amd", "lib", "libpareben_hip_prof.so")
import pareben_amd
from pareben_amd.grid import BuildGrid, AssignToFolds
g = os.path.join(ROOT, "tests", "golden")
X = np.load(os.path.join(g, "BASISbinomial.npy")).astype(np.float64); y = np.load(os.path.join(g, "yBinomial.npy")).astype(np.float64)
alpha, lam = BuildGrid(X, y, 5); fid = AssignToFolds(X, 5)
path = os.path.join(tempfile.gettempdir(), "pareben_phase_bm.bin")
os.environ["PAREBEN_PHASE_DUMP"] = path
with pareben_amd.Context(X, y, fid, 5, prior="binomial") as ctx:
    E, st, cnt = ctx.run(alpha, lam)
    print("timing", ctx.last_timing(), ctx.launch_info())
ph = np.fromfile(path, dtype=np.int64).reshape(-1, 24).astype(np.float64)     # PH_N ticks per fit
tot = ph[:, 7].sum()
names = {0: "weighted rows (BP)", 1: "full-stat rest (bb, quad, out)", 2: "delta ML", 3: "actions", 5: "posterior mode (Newton)"}
print("sum of per-fit ticks %.1f s over %d fits; longest fit %.3f s" % (tot / 1e8, len(ph), ph[:, 7].max() / 1e8))
for k, n in names.items():
    print("  %-32s %6.2f %%" % (n, 100 * ph[:, k].sum() / tot))
print("  %-32s %6.2f %%" % ("other", 100 * (tot - ph[:, list(names)].sum()) / tot))
print("  inside the weighted-rows pass of the full-stat calls: staging w.*Phi in LDS %.2f %%, matrix-core loop %.2f %%" % (
    100 * ph[:, 11].sum() / tot, 100 * ph[:, 8].sum() / tot))
print("  inside the posterior mode: weights %.2f %%, gradient + Hessian %.2f %%, copy + inverse %.2f %%, step + line search %.2f %%; Newton iterations per full-stat %.2f" % (
    100 * ph[:, 16].sum() / tot, 100 * ph[:, 17].sum() / tot, 100 * ph[:, 18].sum() / tot, 100 * ph[:, 19].sum() / tot, ph[:, 20].sum() / max(cnt.reshape(-1, cnt.shape[-1])[:, 5].sum(), 1)))
c = cnt.reshape(-1, cnt.shape[-1])
print("per fit: inner %.1f adds %.1f dels %.1f reest %.1f fullstats %.1f" % tuple(c[:, k].mean() for k in (1, 2, 3, 4, 5)))
