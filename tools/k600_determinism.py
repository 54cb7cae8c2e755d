import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np, pareben_amd
from config4_table import load
d = np.load(os.path.join(ROOT, "tests", "golden", "config4_k600_cells.npz"))
X, y = load(600)
res = []
for rep in range(3):
    with pareben_amd.Context(X, y, d["fold_id"], 5, epis=True) as ctx:
        E, st, cnt = ctx.run(d["alpha"], d["lam"])
    res.append(E)
    print("run", rep, "rel vs oracle", (np.abs(E - d["fold_err"]) / d["fold_err"]).max(), "vs saved gpu table", (np.abs(E - d["gpu_fold_err"]) / d["gpu_fold_err"]).max(), flush=True)
print("bit-identical across runs:", np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2]))
i = list(d["cells"]).index(60)
print("cell 60 fold 4:", [r[i, 3] for r in res], "oracle", d["fold_err"][i, 3], "saved gpu", d["gpu_fold_err"][i, 3])
