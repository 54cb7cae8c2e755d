"""The Epis = "yes" jobs of the authors' own timing table (paper_materials/Timing Tests/
testoutput_time_4-12-2018_gaussian_EPIS_cf.csv: n = 200, k = 300 | 600, nFolds = 5, CrossValidate(search =
"local") on 8 doParallel workers: 12 158 s and 41 243 s elapsed) on one MI355X, on the same data
(tests/golden/yeast_timing_200x600.npz, rebuilt by tools/make_golden.py from the reference's files).
k(k+1)/2 = 45 150 / 180 300 implicit columns; the second one runs through the on-demand Gram-row pool.
Writes gpurun_out/config4_paper.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pareben_amd

d = np.load(os.path.join(ROOT, "tests", "golden", "yeast_timing_200x600.npz"))
n = int(d["n"])
B = np.asfortranarray(np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64) * 2.0 - 1.0)
y = d["y"].astype(np.float64)
paper = {300: {"serial_elapsed_s": 3619.95, "parallel_elapsed_s": 12158.2}, 600: {"serial_elapsed_s": 59577.8, "parallel_elapsed_s": 41243.39}}
ks = [int(v) for v in sys.argv[1:]] or [300, 600]
rep = {}
for k in ks:
    X = np.asfortranarray(B[:, :k])
    t0 = time.time()
    loc = pareben_amd.CrossValidate(X, y, nFolds=5, Epis="yes", prior="gaussian", search="local")
    t_local = time.time() - t0
    t0 = time.time()
    glo = pareben_amd.CrossValidate(X, y, nFolds=5, Epis="yes", prior="gaussian", search="global", return_stats=True)
    t_global = time.time() - t0
    st = glo["stats"]
    rep[str(k)] = {"n": n, "k": k, "implicit_columns": k * (k + 1) // 2, "nFolds": 5,
                   "local_search_wall_s": t_local, "local_alpha_opt": loc["alpha.optimal"], "local_lambda_opt": loc["lambda.optimal"],
                   "local_cells_visited": int((loc["fullCV"][:, 1] != 0).sum()),
                   "global_search_wall_s": t_global, "global_alpha_opt": glo["alpha.optimal"], "global_lambda_opt": glo["lambda.optimal"],
                   "kernel_ms": st["timing"], "launch": st["launch"], "aborted_fits": int(((st["status"] & 8) != 0).sum()),
                   "max_active": int(st["counters"][..., 10].max()), "paper_8_workers": paper.get(k)}
    print(json.dumps(rep[str(k)]), flush=True)
# two main-effect rows of the authors' table (testoutput_time_4-11-2018_gaussian_cf.csv:42, :61)
dm = np.load(os.path.join(ROOT, "tests", "golden", "yeast_timing_main.npz"))
for (nn, kk, nf, ser, par) in ((800, 600, 7, 6182.03, 350.65), (1000, 1200, 10, 2548.08, 2280.33)):
    if ks != [300, 600]:
        break
    Xm = np.asfortranarray(np.unpackbits(dm["bits_%dx%d" % (nn, kk)], axis=0)[:nn].astype(np.float64) * 2.0 - 1.0)
    ym = dm["y"][:nn]
    t0 = time.time()
    loc = pareben_amd.CrossValidate(Xm, ym, nFolds=nf, Epis="no", prior="gaussian", search="local")
    t_local = time.time() - t0
    rep["main_%dx%d" % (nn, kk)] = {"n": nn, "k": kk, "nFolds": nf, "Epis": "no", "local_search_wall_s": t_local,
                                    "local_alpha_opt": loc["alpha.optimal"], "local_lambda_opt": loc["lambda.optimal"],
                                    "paper_8_workers": {"serial_elapsed_s": ser, "parallel_elapsed_s": par}}
    print(json.dumps(rep["main_%dx%d" % (nn, kk)]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "config4_paper.json"), "w"), indent=1)
