"""tests/golden/fulltest_*.npz: the real-R outputs the reference keeps under
paper_materials/Real Data Analysis/Full_Test/ together with the inputs they were computed from (run in the build
container only; the reference does not travel to the GPU box, these fixtures do).

Stored outputs (R 3.5 + CRAN EBEN, read with tools/rdata.py -- a pure XDR parser, nothing in the files is executed):

  EBENoutput_epi0.08_residual*.RDS (3 files)   EBelasticNet.Gaussian(BASIS, y, lambda, alpha, Epis = "no") on
        filter_matrix_epi0.08.zip[, 2:202] (3844 x 201, +-1) and pheno_Zeo_residual
  EBENoutput_Zeo_2018-11-20*.RDS               same call on filter_matrix_main0.05_Zeo.zip[, 2:11397] + pheno_Zeo
  EBENoutput_epi0.08_2018-12-02*.RDS           same call on filter_matrix_main0.05_epi0.08.zip[, 2:11598] + pheno_Zeo
  parEBENoutput_2018-08-15*.RDS                CrossValidate(BASIS, y, nFolds = 3, Epis = "no", prior = "gaussian",
        search = "global") + pheno1 (1200 fits); inputs not named in the tree -- the first 13 248 columns of
        filter_matrix_looser_0.02_main_0.15_epi.zip[, 2:19872] (tools/cv19871_prefix_probe.py); the fixture keeps the whole design
  EBENoutput_part1/2/3_2018-08-16*.RDS         EBelasticNet.Gaussian at that run's optimum; inputs not named in the tree --
        part1 / part2 are the first 13 248 / last 13 247 columns of the same design (tools/parts_probe.py), part3 unidentified

  Subset_Test/SubsetParCV_5-2-2018.RDS (= Subset_4-15-2018_parCV.RDS) and Subset_4-15-2018_model.RDS
        CrossValidate(nFolds = 3, "gaussian", "global") and the refit at its optimum on `filter_matrix` (5356 features) + pheno1
        (Subset_Test/Subset_Test_Gaus_doMPI.R).  That file is one of the blobs missing from the tree (.MISSING_LARGE_BLOBS);
        it is the 233 main-effect columns + 5123 pair columns of the authors' single-locus filter (SL_filter.R), and both
        pieces are in the tree: filter_matrix_looser[, 2:234] and filter_matrix_looser_0.02_main_0.15_epi[, 14750:19872]
        (checked by what comes out: the stored table's first cell and the stored refit to 1e-14,
        tests/test_real_r_golden_gpu.py::test_third_table_vs_real_r).  Read with col_names, so all 3803 rows.
        -> tests/golden/subset5356.npz

Every run of Full_Test loaded its inputs the way Full_Test/dataprep.R:3-4 does -- read.delim() with its default
header = TRUE on files that have no header line -- so the first sample of the design and of the phenotype became
column names and the fits saw rows 2..n.  The fixtures keep all rows (``drop_first_row = 1`` records the
convention); that this is what happened is not an assumption: lambda_max of the stored grid (3.158042295...) is
reproduced only without the first row (3.156882755... with it), and the stored single fits are reproduced to 1e-12
without it and not at all with it.

Designs are +-1 genotype codes, bit-packed along samples and deflated (the 19871-column table is 189 MB of text,
378 kB here).
"""
import io
import os
import sys
import zipfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rdata import read_rds, simplify          # noqa: E402

FT = "/root/reference/paper_materials/Real Data Analysis/Full_Test/"
OUT = os.path.join(ROOT, "tests", "golden")


def genotypes(zipname, member):
    """samples x markers int8 (+-1); the first field of every line is the sample id."""
    rows = []
    with zipfile.ZipFile(FT + zipname).open(member) as f:
        for line in io.TextIOWrapper(f, newline=None):
            rows.append(np.array(line.rstrip("\n").split("\t")[1:], dtype=np.int8))
    G = np.stack(rows)
    assert set(np.unique(G)) <= {-1, 1}
    return G


def packed(G):
    return dict(bits=np.packbits((G > 0).astype(np.uint8), axis=0), n=np.int64(G.shape[0]), p=np.int64(G.shape[1]),
                drop_first_row=np.int64(1))


def fit_fields(path, prefix=""):
    o = simplify(read_rds(FT + path))
    return {prefix + "weight": np.asarray(o["weight"], dtype=np.float64),
            prefix + "WaldScore": np.float64(np.asarray(o["WaldScore"]).reshape(-1)[0]),
            prefix + "Intercept": np.float64(np.asarray(o["Intercept"]).reshape(-1)[0]),
            prefix + "residVar": np.float64(np.asarray(o["residVar"]).reshape(-1)[0]),
            prefix + "lambda": np.float64(np.asarray(o["lambda"]).reshape(-1)[0]),
            prefix + "alpha": np.float64(np.asarray(o["alpha"]).reshape(-1)[0])}


def main():
    # 1. three stored fits on the 201-column design
    d = packed(genotypes("filter_matrix_epi0.08.zip", "filter_matrix_epi0.08"))
    d["pheno"] = np.loadtxt(FT + "pheno_Zeo_residual")
    for tag, f in (("a", "EBENoutput_epi0.08_residual2018-12-05_20_38_39.RDS"),
                   ("b", "EBENoutput_epi0.08_residual_alpha0.05_lambda0.6625895_2018-12-07_11_35_35.RDS"),
                   ("c", "EBENoutput_epi0.08_residual_alpha0.5_lambda0.66258952018-12-07_10_17_43.RDS")):
        d.update(fit_fields(f, tag + "_"))
    # the local-search run whose optimum fit "a" was refitted at (folds there are drawn unseeded, R/LocalSearch.R:13-20,
    # so only its grid values are data one can check)
    cv = simplify(read_rds(FT + "parEBENoutput_epi0.08_residual_cv3local_2018-12-05_20_38_39.RDS"))
    d["local_CrossValidation"] = np.asarray(cv["CrossValidation"], dtype=np.float64)
    d["local_fullCV"] = np.asarray(cv["fullCV"], dtype=np.float64)
    d["local_lambda_optimal"] = np.float64(np.asarray(cv["lambda.optimal"]).reshape(-1)[0])
    d["local_alpha_optimal"] = np.float64(np.asarray(cv["alpha.optimal"]).reshape(-1)[0])
    np.savez_compressed(OUT + "/fulltest_epi008.npz", **d)

    # 2. one stored fit each on the 11396- and the 11597-column designs
    pz = np.loadtxt(FT + "pheno_Zeo")
    d = packed(genotypes("filter_matrix_main0.05_Zeo.zip", "filter_matrix_main0.05_Zeo.05"))
    d["pheno"] = pz
    d.update(fit_fields("EBENoutput_Zeo_2018-11-20_20_18_22.RDS"))
    np.savez_compressed(OUT + "/fulltest_zeo_main.npz", **d)
    d = packed(genotypes("filter_matrix_main0.05_epi0.08.zip", "filter_matrix_main0.05_epi0.08"))
    d["pheno"] = pz
    d.update(fit_fields("EBENoutput_epi0.08_2018-12-02_14_59_41.RDS"))
    np.savez_compressed(OUT + "/fulltest_zeo_main_epi.npz", **d)

    # 3. the stored 3-fold CrossValidate() table on the 19871-column design
    d = packed(genotypes("filter_matrix_looser_0.02_main_0.15_epi.zip", "filter_matrix_looser_0.02_main_0.15_epi"))
    d["pheno"] = np.loadtxt(FT + "pheno1")
    g = simplify(read_rds(FT + "parEBENoutput_2018-08-15_15_48_10.RDS"))
    D, S = g["Results.Detail"], g["Results.Summary"]
    d.update(detail_foldId=np.asarray(D["foldId"]), detail_alpha=np.asarray(D["alpha"]), detail_lambda=np.asarray(D["lambda"]),
             detail_MSE=np.asarray(D["MSE"]), summary_alpha=np.asarray(S["alpha"]), summary_lambda=np.asarray(S["lambda"]),
             summary_SE=np.asarray(S["SE"]), summary_MSE=np.asarray(S["MSE"]),
             lambda_optimal=np.float64(np.asarray(g["lambda.optimal"]).reshape(-1)[0]),
             alpha_optimal=np.float64(np.asarray(g["alpha.optimal"]).reshape(-1)[0]))
    for tag, f in (("part1", "EBENoutput_part1_2018-08-16_11_03_50.RDS"), ("part2", "EBENoutput_part2_2018-08-16_11_03_50.RDS"),
                   ("part3", "EBENoutput_part3_2018-08-16_11_03_50.RDS")):
        d.update(fit_fields(f, tag + "_"))
    np.savez_compressed(OUT + "/fulltest_looser19871.npz", **d)
    # 4. Subset_Test: the missing 5356-column `filter_matrix`, put together from its two pieces
    big = d["bits"]; nb = int(d["n"])
    epi = (np.unpackbits(big, axis=0)[:nb][:, 14748:] > 0)
    main = []
    with zipfile.ZipFile(FT + "filter_matrix_looser.zip").open("filter_matrix_looser") as f:
        for line in io.TextIOWrapper(f, newline=None):
            main.append(np.array(line.split("\t", 234)[1:234], dtype=np.int8))
    main = np.stack(main) > 0
    assert main.shape == (3803, 233) and epi.shape == (3803, 5123)
    ST = "/root/reference/paper_materials/Real Data Analysis/Subset_Test/"
    s = dict(bits=np.packbits(np.hstack([main, epi]).astype(np.uint8), axis=0), n=np.int64(3803), p=np.int64(5356), drop_first_row=np.int64(0),
             pheno=np.loadtxt(ST + "pheno1"))
    g = simplify(read_rds(ST + "SubsetParCV_5-2-2018.RDS"))
    g2 = simplify(read_rds(ST + "Subset_4-15-2018_parCV.RDS"))
    assert np.array_equal(np.asarray(g["Results.Detail"]["MSE"]), np.asarray(g2["Results.Detail"]["MSE"]))      # the same table twice
    D, S = g["Results.Detail"], g["Results.Summary"]
    s.update(detail_foldId=np.asarray(D["foldId"]), detail_alpha=np.asarray(D["alpha"]), detail_lambda=np.asarray(D["lambda"]),
             detail_MSE=np.asarray(D["MSE"]), summary_alpha=np.asarray(S["alpha"]), summary_lambda=np.asarray(S["lambda"]),
             summary_SE=np.asarray(S["SE"]), summary_MSE=np.asarray(S["MSE"]),
             lambda_optimal=np.float64(np.asarray(g["lambda.optimal"]).reshape(-1)[0]),
             alpha_optimal=np.float64(np.asarray(g["alpha.optimal"]).reshape(-1)[0]))
    o = simplify(read_rds(ST + "Subset_4-15-2018_model.RDS"))
    for k_ in ("weight", "WaldScore", "Intercept", "residVar", "lambda", "alpha"):
        s["model_" + k_] = np.asarray(o[k_], dtype=np.float64) if k_ == "weight" else np.float64(np.asarray(o[k_]).reshape(-1)[0])
    np.savez_compressed(OUT + "/subset5356.npz", **s)
    for f in sorted(os.listdir(OUT)):
        if f.startswith("fulltest_") or f.startswith("subset"):
            print(f, os.path.getsize(OUT + "/" + f))


if __name__ == "__main__":
    main()
