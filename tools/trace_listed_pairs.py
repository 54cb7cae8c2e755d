"""GPU decision traces of every (cell, fold) pair the three real-R tables list as off R's value
(tests/golden/*_table_deviations.json): one pareben_fit_gaussian call with pareben_set_trace per pair.

    python tools/trace_listed_pairs.py <outdir> [table ...]

Writes <outdir>/<table>_<cell>_<fold>_gpu.npy (+ .json) and <outdir>/listed_pairs_gpu.json (records and event counters
per pair: what tools/trace_divergence.py's oracle runs are chosen from)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import trace_divergence as td  # noqa: E402

FIX = {"subset5356": "subset5356_table_deviations.json", "yeast": "yeast_table_deviations.json",
       "looser13248": "looser13248_table_deviations.json"}

if __name__ == "__main__":
    from pareben_amd import _lib
    from pareben_amd.grid import AssignToFolds, BuildGrid
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    L = _lib.load()
    L.pareben_set_trace.argtypes = [C.POINTER(C.c_uint64), C.c_int64]
    summary = {}
    for name in (sys.argv[2:] or list(FIX)):
        fx = json.load(open(os.path.join(ROOT, "tests", "golden", FIX[name])))
        X, y = td.load_table(name)
        fid = AssignToFolds(X, 3, sample_kind="Rounding")
        alpha, lam = BuildGrid(X, y, 3)
        for p in fx["pairs"]:
            c, f = p["cell"], p["fold"]
            tr = fid != f
            Xt, yt = np.asfortranarray(X[tr]), np.ascontiguousarray(y[tr])
            buf = np.zeros((td.MAXREC + 1) * td.NSLOT, dtype=np.uint64)
            L.pareben_set_trace(buf.ctypes.data_as(C.POINTER(C.c_uint64)), td.MAXREC)
            r = _lib.fit_gaussian(Xt, yt, lam[c], alpha[c])
            L.pareben_set_trace(None, 0)
            n = int(buf[0])
            path = os.path.join(out, "%s_%d_%d_gpu.npy" % (name, c, f))
            np.save(path, buf[: (n + 1) * td.NSLOT].reshape(n + 1, td.NSLOT))
            summary["%s_%d_%d" % (name, c, f)] = dict(records=n, counters=r["counters"], real_r=p["real_r"], recorded_gpu=p["gpu"])
            print(name, c, f, "records", n, "m_max", r["counters"]["m_max"], "sum_m2_full %.3g" % r["counters"]["sum_m2_full"], flush=True)
    json.dump(summary, open(os.path.join(out, "listed_pairs_gpu.json"), "w"), indent=1)
