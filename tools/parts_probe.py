"""Which inputs produced Full_Test/EBENoutput_part1..3 (2018-08-16; lambda = 2.195448, alpha = 0.5 -- the optimum of the
previous day's CrossValidate() run, itself on the first 13 248 columns of the 19 871-column design)?  The files name none.  On the GPU, fits over column
windows, column thirds (in every order) and row conventions of that design, printed next to the stored feature count /
Wald score / residual variance / intercept / largest locus.

Found: part1 = columns 1..13 248 (equally 1..13 247: column 13 248 is not selected), part2 = columns 6625..19 871, both
with the first sample dropped like every other run of that folder (tests/test_real_r_golden_gpu.py::
test_stored_refits_vs_real_r holds them to 1e-8).  part3 matches nothing tried here."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd
d = np.load(os.path.join(ROOT, "tests", "golden", "fulltest_looser19871.npz")); n = int(d["n"])
G = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64) * 2 - 1; y = d["pheno"].astype(np.float64)
P = G.shape[1]
lam, al = 2.195448, 0.5
for tag in ("part1", "part2", "part3"):
    print(tag, "stored: rows", d[tag + "_weight"].shape[0], "wald %.4f" % float(d[tag + "_WaldScore"]), "resid %.6f" % float(d[tag + "_residVar"]),
          "icpt %.6g" % float(d[tag + "_Intercept"]), "max locus", int(d[tag + "_weight"][:, 0].max()), flush=True)


def fit(name, cols, drop=1, rows=None):
    X, yy = G[drop:][:, cols], y[drop:]
    if rows is not None:
        X, yy = X[rows], yy[rows]
    try:
        r = pareben_amd.EBelasticNet.Gaussian(np.asfortranarray(X), yy, lam, al)
        print(name, "ncol", X.shape[1], "nrow", X.shape[0], "| rows", r["weight"].shape[0], "wald %.4f" % r["WaldScore"], "resid %.6f" % r["residVar"],
              "icpt %.6g" % r["Intercept"], "max locus", int(r["weight"][:, 0].max()), flush=True)
    except pareben_amd.ParebenError as e:
        print(name, "error", e, flush=True)


which = set(sys.argv[1:]) or {"windows"}
if "windows" in which:                       # 13 247-column windows at every 552nd offset, and the two ends
    for o in list(range(0, 6625, 552)) + [6623, 6624, 6625]:
        fit("window@%d" % o, np.arange(o, min(P, o + 13247)))
    fit("first 13248", np.arange(0, 13248))
if "thirds" in which:                        # two of three column thirds, in either order, for both ways of cutting 19 871 in three
    for nm, T in (("6624/6624/6623", [np.arange(0, 6624), np.arange(6624, 13248), np.arange(13248, P)]),
                  ("6623/6624/6624", [np.arange(0, 6623), np.arange(6623, 13247), np.arange(13247, P)])):
        for i in range(3):
            for j in range(3):
                if i != j:
                    fit("%s cbind(third%d, third%d)" % (nm, i + 1, j + 1), np.concatenate([T[i], T[j]]))
if "rows" in which:                          # row conventions and row thirds on the full and the main-effect (14 748-column) designs
    m = n - 1
    for ncol in (P, 14748, 13248):
        for drop in (0, 1, 2):
            fit("cols 1..%d drop %d" % (ncol, drop), np.arange(ncol), drop)
        for k in range(3):
            fit("cols 1..%d row third %d" % (ncol, k + 1), np.arange(ncol), 1, np.arange(m)[k * m // 3:(k + 1) * m // 3])
    fit("last 13247, all rows", np.arange(6624, P), 0)
