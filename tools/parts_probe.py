"""EBENoutput_part1..3 (Full_Test, 2018-08-16; lambda = 2.195448, alpha = 0.5): were they fitted on the 14 748 main-effect
columns of the 19 871-column design (= Subset_Test/filter_matrix_looser_0.02_main)?  Tries column prefixes and row
conventions on the GPU and prints feature counts / Wald scores next to the stored ones.  Report only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd
d = np.load(os.path.join(ROOT, "tests", "golden", "fulltest_looser19871.npz")); n = int(d["n"])
G = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64) * 2 - 1; y = d["pheno"].astype(np.float64)
for tag in ("part1", "part2", "part3"):
    print(tag, "stored: rows", d[tag + "_weight"].shape[0], "wald %.4f" % float(d[tag + "_WaldScore"]), "resid %.6f" % float(d[tag + "_residVar"]),
          "icpt %.6g" % float(d[tag + "_Intercept"]), "max locus", int(d[tag + "_weight"][:, 0].max()), flush=True)
lam, al = 2.195448, 0.5
P = G.shape[1]
Xd, yd = G[1:], y[1:]
A = [np.arange(0, 6624), np.arange(6624, 13248), np.arange(13248, P)]
B = [np.arange(0, 6623), np.arange(6623, 13247), np.arange(13247, P)]
for nm, T in (("6624/6624/6623", A), ("6623/6624/6624", B)):
    for i in range(3):
        for j in range(3):
            if i == j:
                continue
            cols = np.concatenate([T[i], T[j]])
            r = pareben_amd.EBelasticNet.Gaussian(np.asfortranarray(Xd[:, cols]), yd, lam, al)
            print(nm, "cbind(third%d, third%d)" % (i + 1, j + 1), "ncol", len(cols), "rows", r["weight"].shape[0], "wald %.4f" % r["WaldScore"], "resid %.6f" % r["residVar"],
                  "icpt %.6g" % r["Intercept"], "max locus", int(r["weight"][:, 0].max()), flush=True)
