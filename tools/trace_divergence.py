"""First-divergence trace of one Gaussian fit of a stored real-R table: where, and by what margin, two builds part.

Every backend writes one 16-word record per inner iteration (types.h TR_*): the arg-max feature, its action and dML,
the runner-up, the block cut-off and the relative distance of the nearest dML to it, then the noise precision and
order-free XOR hashes of S_in, Q_in and (Sigma, mu) after the iteration.

    python tools/trace_divergence.py run <table> <cell> <fold> <backend> <out.npy>     backend: oracle | emul | gpu | gpu-strict
    python tools/trace_divergence.py cmp <a.npy> <b.npy>

`oracle` = oracle/liboracle.so (netlib order; reproduces real R to 1e-15 on non-chaotic fits and on about half of the chaotic ones), `emul` = the device source
compiled for the CPU (tests/emul), `gpu` = pareben_fit_gaussian on cuda:0 (default summation order), `gpu-strict` =
the same with PAREBEN_STRICT_ORDER=1.  Tables: subset5356 | yeast | looser13248 (tests/golden)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
NSLOT = 16
NAMES = ("iter", "i_iter", "M_before", "nu", "act", "n_todo", "sel", "M_after",
         "best", "second", "cutoff", "nearest", "beta", "hSin", "hQin", "hSig")
MAXREC = 40000


def load_table(name):
    g = os.path.join(ROOT, "tests", "golden")
    if name == "yeast":
        d = np.load(os.path.join(g, "yeast_looser10000.npz"))
        n = int(d["n"])
        X = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64) * 2.0 - 1.0
        y = d["pheno"].astype(np.float64)
    else:
        fn = "subset5356.npz" if name == "subset5356" else "fulltest_looser19871.npz"
        d = np.load(os.path.join(g, fn))
        n, k = int(d["n"]), int(d["drop_first_row"])
        X = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64)[k:] * 2.0 - 1.0
        y = d["pheno"].astype(np.float64)[k:]
        if name == "looser13248":
            X = X[:, :13248]
    return np.asfortranarray(X), y


def training_set(name, cell, fold):
    from pareben_amd.grid import AssignToFolds, BuildGrid
    if name.startswith("config4_"):                 # BASELINE config 4 (Epis = "yes") at k markers: grid and folds of the committed fixture
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from config4_table import load
        k = int(name.split("_")[1])
        X, y = load(k)
        d = np.load(os.path.join(ROOT, "tests", "golden", "config4_k%d_cells.npz" % k))
        i = int(np.nonzero(d["cells"] == cell)[0][0])
        tr = d["fold_id"] != fold
        return np.asfortranarray(X[tr]), np.ascontiguousarray(y[tr]), float(d["alpha"][i]), float(d["lam"][i])
    X, y = load_table(name)
    fid = AssignToFolds(X, 3, sample_kind="Rounding")
    alpha, lam = BuildGrid(X, y, 3)
    tr = fid != fold
    return np.asfortranarray(X[tr]), np.ascontiguousarray(y[tr]), float(alpha[cell]), float(lam[cell])


def run(name, cell, fold, backend, out):
    X, y, a, l = training_set(name, cell, fold)
    N, K = X.shape
    epis = name.startswith("config4_")
    buf = np.zeros((MAXREC + 1) * NSLOT, dtype=np.uint64)
    dp = C.POINTER(C.c_double)
    bp = buf.ctypes.data_as(C.POINTER(C.c_uint64))
    if backend == "oracle":
        import oracle_lib
        L = oracle_lib.lib()
        L.eben_set_trace.argtypes = [C.POINTER(C.c_uint64), C.c_int64]
        L.eben_set_trace(bp, MAXREC)
        if epis:
            oracle_lib.set_capacity_policy(True, 0)
        r = oracle_lib.fit_gaussian(X, y, l, a, epis=epis)
        L.eben_set_trace(None, 0)
        info = dict(intercept=r["intercept"], residual=r["residual"], counters=r["counters"])
    elif backend == "emul":
        import emul_lib
        L = emul_lib.lib()
        L.emul_set_trace.argtypes = [C.POINTER(C.c_uint64), C.c_int64]
        L.emul_set_trace(bp, MAXREC)
        cap = L.emul_default_cap(K)
        o = np.zeros(3); used = np.zeros(cap + 1, dtype=np.int32); mu = np.zeros(cap + 1); sd = np.zeros(cap + 1)
        cnt = np.zeros(14, dtype=np.int64)
        L.emul_gm_fit(X.ctypes.data_as(dp), y.ctypes.data_as(dp), N, K, C.c_double(l), C.c_double(a), o.ctypes.data_as(dp),
                      used.ctypes.data_as(C.POINTER(C.c_int32)), mu.ctypes.data_as(dp), sd.ctypes.data_as(dp),
                      cnt.ctypes.data_as(C.POINTER(C.c_int64)))
        L.emul_set_trace(None, 0)
        info = dict(intercept=o[0], residual=1 / (o[1] + 1e-10), counters=cnt.tolist())
    else:
        if backend == "gpu-strict":
            os.environ["PAREBEN_STRICT_ORDER"] = "1"
        from pareben_amd import _lib
        L = _lib.load()
        L.pareben_set_trace.argtypes = [C.POINTER(C.c_uint64), C.c_int64]
        L.pareben_set_trace(bp, MAXREC)
        Beta = np.zeros((K * (K + 1) // 2, 5) if epis else (K, 4), order="F"); w = C.c_double(); ic = C.c_double(); rs = C.c_double()
        cnt = np.zeros(14, dtype=np.int64)
        rc = (L.pareben_fit_gaussian_epis if epis else L.pareben_fit_gaussian)(X.ctypes.data_as(dp), y.ctypes.data_as(dp), l, a, Beta.ctypes.data_as(dp), C.byref(w), C.byref(ic),
                                    N, K, 0, C.byref(rs), 0, cnt.ctypes.data_as(C.POINTER(C.c_int64)))
        L.pareben_set_trace(None, 0)
        assert rc == 0, L.pareben_last_error()
        info = dict(intercept=ic.value, residual=rs.value, counters=cnt.tolist())
    n = int(buf[0])
    np.save(out, buf[: (n + 1) * NSLOT].reshape(n + 1, NSLOT))
    json.dump(dict(table=name, cell=cell, fold=fold, backend=backend, alpha=a, lam=l, N=N, K=K, records=n, **info),
              open(out + ".json", "w"), indent=1, default=int)
    print(backend, "records", n, info)


def view(path):
    """-> (ints [n, 8]: iter, i_iter, M_before, nu, act, n_todo, sel, M_after; doubles [n, 5]: best, second, cutoff, nearest, beta;
    hashes [n, 4]: XOR of S_in, Q_in, (Sigma, mu) bit patterns, and the order-free hash of the active set (0 in traces that predate it))"""
    t = np.load(path)[1:]
    ints = t[:, :8].astype(np.int64)
    used_hash = (t[:, 7] >> np.uint64(32)).astype(np.uint64)
    ints[:, 7] = (t[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
    dbl = t[:, 8:13].copy().view(np.float64)
    return ints, dbl, np.concatenate([t[:, 13:16], used_hash[:, None]], axis=1)


def compare(pa, pb, quiet=False):
    ia, da, ha = view(pa)
    ib, db, hb = view(pb)
    n = min(len(ia), len(ib))
    dec = [0, 1, 2, 3, 4, 5]                                  # iter, i_iter, M_before, nu, act, n_todo
    diff_dec = np.nonzero((ia[:n, dec] != ib[:n, dec]).any(axis=1))[0]
    diff_hash = np.nonzero((ha[:n, :3] != hb[:n, :3]).any(axis=1))[0]
    diff_best = np.nonzero(da[:n, 0] != db[:n, 0])[0]
    res = dict(records=(len(ia), len(ib)),
               first_hash_difference=int(diff_hash[0]) if len(diff_hash) else None,
               first_best_dml_bit_difference=int(diff_best[0]) if len(diff_best) else None,
               first_decision_difference=int(diff_dec[0]) if len(diff_dec) else None)
    if len(diff_dec):
        k = int(diff_dec[0])
        rec = {}
        for tag, ii, dd in (("a", ia, da), ("b", ib, db)):
            rec[tag] = {NAMES[j]: int(ii[k, j]) for j in range(8)}
            rec[tag].update({NAMES[8 + j]: float(dd[k, j]) for j in range(5)})
        res["at_divergence"] = rec
        # by how much the two builds' dML values differ just there (relative), against the margins a decision had:
        # runner-up gap (best - second) / best and the nearest dML to the block cut-off
        ba, bb = da[k, 0], db[k, 0]
        res["rel_diff_best_dml"] = abs(ba - bb) / max(abs(ba), 1e-300)
        res["runner_up_gap_a"] = (da[k, 0] - da[k, 1]) / max(abs(da[k, 0]), 1e-300)
        res["runner_up_gap_b"] = (db[k, 0] - db[k, 1]) / max(abs(db[k, 0]), 1e-300)
        res["nearest_to_cutoff_a"] = float(da[k, 3]); res["nearest_to_cutoff_b"] = float(db[k, 3])
        if k > 0:
            res["rel_diff_beta_before"] = abs(da[k - 1, 4] - db[k - 1, 4]) / abs(da[k - 1, 4])
    if not quiet:
        print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6])
    else:
        compare(sys.argv[2], sys.argv[3])
