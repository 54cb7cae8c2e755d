"""Regenerate tests/golden/ from the reference tree (run in the build container only; the
reference does not travel to the GPU box, these small fixtures do).

  data fixtures      BASIS / y / BASISbinomial / yBinomial from /root/reference/data/*.rda
                     (the data files the reference's own example and test use)
  rds_10000.npz      the stored real-R CrossValidate() output
                     paper_materials/Real Data Analysis/10000_Features/LooserSubset_10000_ParCV_5-3-2018.RDS
  config1_gm.npz     oracle output for the reference's own test case (tests/CrossValidate-test.R):
                     BASIS[1:50,1:100], nFolds=3, 20x20 grid -- 1200 fold SSEs + summary
  basis481_gm.npz    oracle output for a 15-cell sub-grid of the full bundled data, nFolds=5

Files are read with tools/rdata.py (a pure XDR parser; nothing in the files is executed).
"""
import json
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rdata import read_rda, read_rds, simplify          # noqa: E402
from pareben_amd.grid import BuildGrid, AssignToFolds, summarise_cv   # noqa: E402
import oracle_lib as O                                   # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    B = simplify(read_rda(REF + "/data/BASIS.rda")["BASIS"])
    y = simplify(read_rda(REF + "/data/y.rda")["y"]).reshape(-1)
    Bb = simplify(read_rda(REF + "/data/BASISbinomial.rda")["BASISbinomial"])
    yb = simplify(read_rda(REF + "/data/yBinomial.rda")["yBinomial"]).reshape(-1)
    np.save(OUT + "/BASIS.npy", B.astype(np.int8))
    np.save(OUT + "/y.npy", y)
    np.save(OUT + "/BASISbinomial.npy", Bb.astype(np.int8))
    np.save(OUT + "/yBinomial.npy", yb.astype(np.int8))

    g = simplify(read_rds(REF + "/paper_materials/Real Data Analysis/10000_Features/LooserSubset_10000_ParCV_5-3-2018.RDS"))
    D, S = g["Results.Detail"], g["Results.Summary"]
    np.savez_compressed(OUT + "/rds_10000.npz",
                        detail_foldId=D["foldId"], detail_alpha=D["alpha"], detail_lambda=D["lambda"], detail_MSE=D["MSE"],
                        summary_alpha=S["alpha"], summary_lambda=S["lambda"], summary_SE=S["SE"], summary_MSE=S["MSE"],
                        lambda_optimal=g["lambda.optimal"], alpha_optimal=g["alpha.optimal"])

    Bf = B.astype(np.float64)
    X, yy = Bf[:50, :100], y[:50]
    fid = AssignToFolds(X, 3)
    a, l = BuildGrid(X, yy, 3)
    E, cnt, rc = O.cv_grid(X, yy, fid, 3, a, l)
    assert rc == 0
    a_s, l_s, se, err, idx = summarise_cv(a, l, E, 3)
    np.savez_compressed(OUT + "/config1_gm.npz", alpha=a, lam=l, fold_id=fid, fold_err=E,
                        summary_alpha=a_s, summary_lambda=l_s, summary_SE=se, summary_MSE=err, idx=idx,
                        counters=np.array([cnt[k] for k in sorted(cnt)]), counter_names=np.array(sorted(cnt)))

    fid5 = AssignToFolds(Bf, 5)
    a5, l5 = BuildGrid(Bf, y, 5)
    A, L = np.unique(a5)[::-1], np.unique(l5)[::-1]
    sel = [(A[i], L[j]) for i in (0, 9, 19) for j in (0, 5, 10, 14, 19)]
    aa = np.array([s[0] for s in sel]); ll = np.array([s[1] for s in sel])
    E5, cnt5, rc = O.cv_grid(Bf, y, fid5, 5, aa, ll)
    assert rc == 0
    np.savez_compressed(OUT + "/basis481_gm.npz", alpha=aa, lam=ll, fold_id=fid5, fold_err=E5,
                        counters=np.array([cnt5[k] for k in sorted(cnt5)]), counter_names=np.array(sorted(cnt5)))

    # config 3: BASISbinomial / yBinomial, binomial prior, nFolds=5, 20x20 grid (oracle, ~3 min on 8 threads)
    Xb = Bb.astype(np.float64); ybf = yb.astype(np.float64)
    fidb = AssignToFolds(Xb, 5)
    ab, lb = BuildGrid(Xb, ybf, 5)
    Eb, cntb, rc = O.cv_grid(Xb, ybf, fidb, 5, ab, lb, prior="binomial", n_threads=8)
    assert rc == 0
    a_s, l_s, se, err, idx = summarise_cv(ab, lb, Eb, 5, prior="binomial")
    np.savez_compressed(OUT + "/config3_bm.npz", alpha=ab, lam=lb, fold_id=fidb, fold_err=Eb, summary_alpha=a_s,
                        summary_lambda=l_s, summary_SE=se, summary_Likelihood=err, idx=idx)

    # config-4 style epistasis case: BASIS[1:200,1:60], Epis="yes", nFolds=5 -- the reference's own
    # (near-degenerate) grid, plus a sub-grid on the normalised target where fits are non-trivial
    Xe, ye = Bf[:200, :60], y[:200]
    fide = AssignToFolds(Xe, 5)
    ae, le = BuildGrid(Xe, ye, 5, "yes")
    Ee, _, rc = O.cv_grid(Xe, ye, fide, 5, ae, le, epis=True, n_threads=8)
    assert rc == 0
    ys = (ye - ye.mean()) / np.linalg.norm(ye - ye.mean())
    a2, l2 = BuildGrid(Xe, ys, 5, "yes")
    sel = np.arange(3, 400, 19)
    E2, _, rc2 = O.cv_grid(Xe, ys, fide, 5, a2[sel], l2[sel], epis=True, n_threads=8)
    np.savez_compressed(OUT + "/config4_gf.npz", alpha=ae, lam=le, fold_id=fide, fold_err=Ee, alpha_scaled=a2[sel],
                        lam_scaled=l2[sel], fold_err_scaled=E2, y_scaled=ys)

    # yeast design of the stored real-R run (10000_Features): filter_matrix_looser[, 2:10001] (+-1) and
    # pheno1, bit-packed along samples (the tab-separated text is 512 MB; packed + deflated 60 kB)
    yeast_txt = os.environ.get("YEAST_MATRIX", "/tmp/work/filter_matrix_looser")   # unzip of Full_Test/filter_matrix_looser.zip
    if os.path.exists(yeast_txt):
        rows = []
        with open(yeast_txt) as f:
            for line in f:
                rows.append(np.array(line.split("\t", 10001)[1:10001], dtype=np.int8))
        Gy = np.stack(rows)
        ph = np.loadtxt(REF + "/paper_materials/Real Data Analysis/Full_Test/pheno1")
        np.savez_compressed(OUT + "/yeast_looser10000.npz", bits=np.packbits((Gy > 0).astype(np.uint8), axis=0),
                            n=np.int64(Gy.shape[0]), p=np.int64(Gy.shape[1]), pheno=ph)

    # The Epis data set of the authors' timing script (paper_materials/Timing Tests/test_time_Gaus.R:13-42,
    # n = 200, k = 600 rows of its template): genotype rows of the samples that have a phenotype, then
    # set.seed(1); sample(1:nrow, n); sample(1:ncol, k) with the R < 3.6 sampler (the runs are from April
    # 2018), phenotypes 1..n.  The k = 300 tests use the first 300 of these columns (the first picks of
    # sample() do not depend on its size).  PAPER_TIMING=1 enables it (46 s to parse the 310 MB table).
    if os.environ.get("PAPER_TIMING"):
        import zipfile
        import pandas as pd
        from pareben_amd.rlang import RRandom
        tt = REF + "/paper_materials/Timing Tests"
        with zipfile.ZipFile(tt + "/genotype_full.zip").open("genotype_full.txt") as f:
            df = pd.read_csv(f, sep="\t", header=0, dtype=str)
        ph = pd.read_csv(tt + "/pheno_left.txt", sep=r"\s+", header=None)
        keep = df.iloc[:, 0].isin(set(ph[0].values)).values
        geno3 = df.loc[keep].iloc[:, 1:].values.astype(np.int8)            # samples x markers, +-1
        rng = RRandom(1, sample_kind="Rounding")
        sn = np.array(rng.sample(range(1, geno3.shape[0] + 1), 200))
        sk = np.array(rng.sample(range(1, geno3.shape[1] + 1), 600))
        Bt = geno3[sn - 1][:, sk - 1]
        np.savez_compressed(OUT + "/yeast_timing_200x600.npz", bits=np.packbits((Bt > 0).astype(np.uint8), axis=0),
                            n=np.int64(200), k=np.int64(600), y=ph[1].values[:200].astype(np.float64),
                            sample_n=sn.astype(np.int32), sample_k=sk.astype(np.int32))
        # two main-effect rows of the same script's table (testoutput_time_4-11-2018_gaussian_cf.csv:42, :61);
        # the column sample depends on n (it is drawn after the n row picks), so each (n, k) is drawn afresh
        main = {"y": ph[1].values[:1000].astype(np.float64)}
        for (nn, kk) in ((1000, 1200), (800, 600)):
            rng = RRandom(1, sample_kind="Rounding")
            sn = np.array(rng.sample(range(1, geno3.shape[0] + 1), nn))
            sk = np.array(rng.sample(range(1, geno3.shape[1] + 1), kk))
            main["bits_%dx%d" % (nn, kk)] = np.packbits((geno3[sn - 1][:, sk - 1] > 0).astype(np.uint8), axis=0)
        np.savez_compressed(OUT + "/yeast_timing_main.npz", **main)

    # numbers recorded in SURVEY.md section 10 (compiled reference C, survey session)
    known = {
        "config1": {
            "folds": "13112123332113311233323312312123221231311121222223",
            "folds_rounding": "21111232333123331311231131112213222232221322313231",
            "lambda_first": 2.511584267297302, "lambda_last": 0.002511584267297304,
            "cell_alpha1_lambdamax": [2246.408042437753, 2004.1589385129005, 1558.6573039729979],
            "cell_alpha005_lambdamin": [2131.7403270851223, 3156.5655840558547, 3180.85087744276],
            "alpha_opt": 1.0, "lambda_opt": 0.022249290937151205,
            "cv_error": 1919.1816087603045, "SE": 180.6896031417665,
            "nonzero_total": 3682, "max_active": 13,
        },
        "rng": {"runif3": [0.2655087, 0.3721239, 0.5728534],
                "sample10": [9, 4, 7, 1, 2, 5, 3, 10, 6, 8],
                "sample10_rounding": [3, 4, 5, 7, 2, 8, 9, 6, 10, 1]},
        "config3": {"lambda_first": 4.901894677425395, "lambda_last": 0.004901894677425393,
                    "alpha_opt": 0.19999999999999996, "lambda_opt": 0.014589761289879953,
                    "likelihood": 0.3383112550401007, "SE": 0.023711752417084345, "max_active": 59},
        "gf_basis200x60": {"lambda_max": 28.94863306856644, "alpha_opt": 1.0, "lambda_opt": 289.4863306856643,
                           "cv_error": 4684.6559987079645},
        "yeast10000": {"lambda_max_x10": 3.156882755270842, "detail_mse_row1": 486.80072139},
    }
    with open(OUT + "/survey_known_answers.json", "w") as f:
        json.dump(known, f, indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
