"""Was the stored August 2018 CV table (Full_Test/parEBENoutput_2018-08-15*.RDS) computed on a column subset of the
19 871-column design?  Cell 0 (alpha = 1, lambda = lambda_max) and the table's optimum cell on several column sets, R 3.5 folds,
next to R's stored fold SSEs.  Report only.

Answer: yes -- the first 13 248 columns (the design of EBENoutput_part1): two of the three cells agree with R to all printed
digits, the third (alpha = 1) in one fold of three and to 1e-3 in the others, as chaotic alpha = 1 fits do
(tests/test_real_r_golden_gpu.py::test_second_table_vs_real_r holds the whole table)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd
from pareben_amd.grid import AssignToFolds
d = np.load(os.path.join(ROOT, "tests", "golden", "fulltest_looser19871.npz")); n = int(d["n"])
G = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64)[1:] * 2 - 1; y = d["pheno"].astype(np.float64)[1:]
P = G.shape[1]
fid = AssignToFolds(G, 3, sample_kind="Rounding")
key = {}
for f_, a_, l_, m_ in zip(d["detail_foldId"], d["detail_alpha"], d["detail_lambda"], d["detail_MSE"]):
    key[(round(float(a_), 6), "%.6e" % l_, int(f_))] = m_
lmax = float(d["detail_lambda"].max()); lopt = float(d["lambda_optimal"])
cells = [(1.0, lmax), (0.5, lopt), (0.05, float(np.unique(d["detail_lambda"])[10]))]
for a_, l_ in cells:
    print("R   alpha %.2f lambda %.6g:" % (a_, l_), [round(float(key[(round(a_, 6), "%.6e" % l_, f)]), 4) for f in (1, 2, 3)], flush=True)
for name, cols in (("all 19871", np.arange(P)), ("main 14748", np.arange(14748)), ("first 13248", np.arange(13248)), ("last 13247", np.arange(6624, P)),
                   ("epi 5123", np.arange(14748, P)), ("first 10000", np.arange(10000)), ("first 6624", np.arange(6624))):
    X = np.asfortranarray(G[:, cols])
    with pareben_amd.Context(X, y, fid, 3) as ctx:
        E, st, cnt = ctx.run(np.array([c[0] for c in cells]), np.array([c[1] for c in cells]))
    for (a_, l_), e, s in zip(cells, E, st):
        print("%-12s alpha %.2f lambda %.6g:" % (name, a_, l_), np.round(e, 4).tolist(), "status", s.tolist(), flush=True)
