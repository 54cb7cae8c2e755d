"""Where, and by what margin, the HIP path leaves real R's trajectory on the fits the three real-R tables list as deviating.

For every listed (cell, fold) pair with both traces present (tools/trace_listed_pairs.py for the GPU, tools/trace_divergence.py
run ... oracle for the netlib-order oracle, which reproduces real R on these fits to 1e-15) the two decision traces are
walked record by record.  Records whose arg-max feature differs but whose action type, to-do count and resulting model size
agree are exact ties between duplicated columns (both builds report best == runner-up bit for bit) and change nothing: the
action goes to the lowest-index member of the to-do list either way.  The first REAL divergence is the first inner iteration
whose (action type, to-do count, model size after) differ; reported there: the relative gap between the best dML and the
runner-up and the relative distance of the nearest dML to the block cut-off on both sides -- the margin the decision had --
next to the relative difference between the two builds' dML values over the preceding iterations -- the rounding noise of
two summation orders.

    python tools/first_divergence.py <trace dir> <out.json>"""
import glob
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_divergence import view  # noqa: E402


def analyse(po, pg):
    io, do, ho = view(po)
    ig, dg, hg = view(pg)
    n = min(len(io), len(ig))
    key = [0, 1, 2, 4, 5, 6, 7]                 # iter, i_iter, M_before, act, n_todo, sel, M_after
    kd = (io[:n][:, key] != ig[:n][:, key]).any(axis=1)
    both = (ho[:n, 3] != 0) & (hg[:n, 3] != 0)  # the active-set hash, where both traces carry it: the same features, not only as many
    kd |= both & (ho[:n, 3] != hg[:n, 3])
    ties = (io[:n, 3] != ig[:n, 3]) & ~kd
    rel = np.abs(do[:n, 0] - dg[:n, 0]) / np.maximum(np.abs(do[:n, 0]), 1e-300)
    out = dict(records_oracle=len(io), records_gpu=len(ig), benign_argmax_ties_before=None, first_divergence=None)
    if not kd.any():
        out["benign_argmax_ties_before"] = int(ties.sum())
        return out
    k = int(np.nonzero(kd)[0][0])
    out["benign_argmax_ties_before"] = int(ties[:k].sum())
    exact = [bool(do[j, 0] == do[j, 1] and dg[j, 0] == dg[j, 1]) for j in np.nonzero(ties[:k])[0]]
    out["benign_ties_all_exact"] = bool(all(exact))
    gap = lambda d: float((d[k, 0] - d[k, 1]) / max(abs(d[k, 0]), 1e-300))
    lo = max(0, k - 50)
    noise = float(np.median(rel[lo:k])) if k > lo else float("nan")
    kind = "action type of the arg-max" if io[k, 4] != ig[k, 4] else ("to-do list" if io[k, 5] != ig[k, 5] else "model after the block")
    margin = min(gap(do), gap(dg)) if kind.startswith("action") else min(float(do[k, 3]), float(dg[k, 3]), gap(do), gap(dg))
    out["first_divergence"] = dict(
        record=k, outer_iteration=int(io[k, 0]), inner_iteration=int(io[k, 1]), active_set=int(io[k, 2]), kind=kind,
        oracle=dict(feature=int(io[k, 3]), action=int(io[k, 4]), n_todo=int(io[k, 5]), best_dml=float(do[k, 0]), runner_up=float(do[k, 1]),
                    runner_up_gap=gap(do), nearest_to_cutoff=float(do[k, 3])),
        gpu=dict(feature=int(ig[k, 3]), action=int(ig[k, 4]), n_todo=int(ig[k, 5]), best_dml=float(dg[k, 0]), runner_up=float(dg[k, 1]),
                 runner_up_gap=gap(dg), nearest_to_cutoff=float(dg[k, 3])),
        decision_margin=margin, dml_noise_between_builds=noise,
        beta_rel_diff_before=float(abs(do[k - 1, 4] - dg[k - 1, 4]) / abs(do[k - 1, 4])) if k > 0 else None)
    return out


if __name__ == "__main__":
    d, out = sys.argv[1], sys.argv[2]
    res = {}
    for po in sorted(glob.glob(os.path.join(d, "*_oracle.npy"))):
        tag = os.path.basename(po)[:-len("_oracle.npy")]
        pg = os.path.join(d, tag + "_gpu.npy")
        if os.path.exists(pg):
            res[tag] = analyse(po, pg)
    margins = [v["first_divergence"]["decision_margin"] for v in res.values() if v["first_divergence"]]
    summary = dict(pairs=len(res), diverging=len(margins), max_decision_margin=max(margins) if margins else None,
                   median_decision_margin=float(np.median(margins)) if margins else None,
                   pairs_without_divergence=[k for k, v in res.items() if not v["first_divergence"]])
    json.dump(dict(summary=summary, pairs=res), open(out, "w"), indent=1)
    print(json.dumps(summary, indent=1))
    for k, v in res.items():
        f = v["first_divergence"]
        if f:
            print("%-22s rec %5d M %4d %-28s margin %.2e noise %.2e ties %d" % (k, f["record"], f["active_set"], f["kind"], f["decision_margin"], f["dml_noise_between_builds"], v["benign_argmax_ties_before"]))
