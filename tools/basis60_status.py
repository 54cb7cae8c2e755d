"""Status words of the scaled-target sub-grid of tests/test_hip_parity.py::test_epistasis_vs_golden (BASIS[1:200,1:60],
Epis = "yes") on the HIP path -> gpurun_out/r03/basis60_scaled_status.npy (committed into
tests/golden/config4_grid_status.npz as basis60_scaled_status)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import pareben_amd  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "config4_gf.npz"))
B = np.load(os.path.join(ROOT, "tests", "golden", "BASIS.npy")).astype(np.float64)[:200, :60]
with pareben_amd.Context(B, g["y_scaled"], g["fold_id"], 5, epis=True) as ctx:
    E2, st, cnt = ctx.run(g["alpha_scaled"], g["lam_scaled"])
print("shape", st.shape, "hist", dict(zip(*np.unique(st, return_counts=True))))
os.makedirs(os.path.join(ROOT, "gpurun_out", "r03"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "r03", "basis60_scaled_status.npy"), st.astype(np.int8))
ref = g["fold_err_scaled"]
ok = (st & 9) == 0
print("ok", ok.sum(), "finite ref", np.isfinite(ref).sum(), "ok where ref finite", ok[np.isfinite(ref)].sum())
