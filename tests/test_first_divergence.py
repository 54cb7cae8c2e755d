"""The committed first-divergence analysis (tools/first_divergence.py over the decision traces of the listed deviating fits of
the three real-R tables, HIP path vs the netlib-order oracle): at least two pairs per table, and every first real divergence a
decision whose margin -- relative gap between the best dML and its runner-up, or distance of the nearest dML to the block
cut-off -- is below 1e-12, i.e. far inside the rounding noise between two summation orders."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_first_divergences_are_coin_flips():
    j = json.load(open(os.path.join(ROOT, "profiles", "r03", "first_divergence.json")))
    pairs = j["pairs"]
    per_table = {}
    for k, v in pairs.items():
        f = v["first_divergence"]
        if f is None:
            continue
        per_table.setdefault(k.rsplit("_", 2)[0], []).append(f["decision_margin"])
        assert f["decision_margin"] < 1e-12, (k, f)
        assert v["benign_ties_all_exact"] in (True, False)
    assert set(per_table) == {"subset5356", "yeast", "looser13248"} and all(len(m) >= 2 for m in per_table.values()), per_table
    assert sum(len(m) for m in per_table.values()) >= 6
    assert j["summary"]["max_decision_margin"] < 1e-12
