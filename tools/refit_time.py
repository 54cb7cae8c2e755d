"""Wall time of one EBelasticNet.Gaussian refit on the stored Full_Test designs, with and without helper workgroups
(PAREBEN_SHARE=0 runs the fit on one workgroup alone); checks that the outputs are bit-identical."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd
G = os.path.join(ROOT, "tests", "golden")
for name in ("epi008", "zeo_main", "zeo_main_epi", "looser19871"):
    d = np.load(G + "/fulltest_%s.npz" % name); n = int(d["n"])
    X = np.asfortranarray(np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64)[1:] * 2 - 1); y = d["pheno"].astype(np.float64)[1:]
    lam, al = (float(d["c_lambda"]), float(d["c_alpha"])) if name == "epi008" else ((float(d["lambda"]), float(d["alpha"])) if "lambda" in d else (2.195448, 0.5))
    res = {}
    for mode in ("0", None, "0", None):
        if mode is None:
            os.environ.pop("PAREBEN_SHARE", None)
        else:
            os.environ["PAREBEN_SHARE"] = mode
        t = time.time(); r = pareben_amd.fit_gaussian(X, y, lam, al); dt = time.time() - t
        key = "alone" if mode == "0" else "helped"
        if key in res:
            assert np.array_equal(res[key][0]["Beta"], r["Beta"]) and res[key][0]["wald"] == r["wald"]
        res[key] = (r, dt)
    a, h = res["alone"], res["helped"]
    same = np.array_equal(a[0]["Beta"], h[0]["Beta"]) and a[0]["wald"] == h[0]["wald"] and a[0]["intercept"] == h[0]["intercept"]
    print(name, X.shape, "alone %.2f s  helped %.2f s  bit-identical %s  m_max %d inner %d" % (a[1], h[1], same, h[0]["counters"]["m_max"], h[0]["counters"]["n_inner"]), flush=True)
