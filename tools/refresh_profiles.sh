#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + the PMC passes of one bench.py workload, each in
# its own run (the guide's HBM/rocprofv3 section: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc never
# together with the trace domains gpurun refuses).  Writes gpurun_out/prof_<tag>_<workload>/
#   kernel_stats.csv, pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv, pmc_SQ_a.csv, pmc_SQ_b.csv, bench_line.json and
#   pmc_<workload>.json (what bench.py attaches as roofline.traffic / roofline.pmc: copy it to profiles/<round>/).
# usage: bash tools/refresh_profiles.sh <tag> [workload=config2] [passes="stats fetch write sqa sqb"]
set -e
TAG=${1:-vX}
WL=${2:-config2}
PASSES=${3:-"stats fetch write sqa sqb"}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --workload $WL --steps 1 --warmup 0 --cpu-baseline 0"
for P in $PASSES; do
  case $P in
    stats) FL="--kernel-trace --stats" ;;
    fetch) FL="--kernel-trace --pmc FETCH_SIZE" ;;
    write) FL="--kernel-trace --pmc WRITE_SIZE" ;;
    sqa)   FL="--kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" ;;
    sqb)   FL="--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" ;;
  esac
  rocprofv3 $FL -d $OUT/$P -o out --output-format csv -- python3 $ARGS > $OUT/${P}_bench.json.log 2> $OUT/$P.err
  echo "$P pass done"
done
python3 - "$OUT" "$TAG" "$WL" <<'PY'
import csv, glob, json, os, shutil, sys
out, tag, wl = sys.argv[1:4]
def find(d, pat):
    f = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return f[0] if f else None
# the dominant kernel = largest total duration in the stats pass
kern, kms, calls, share = None, None, None, None
f = find("stats", "*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    kern = rows[0]["Name"].split("(")[0]
    calls = int(rows[0]["Calls"]); kms = float(rows[0]["AverageNs"]) / 1e6; share = float(rows[0]["Percentage"])
    shutil.copy(f, os.path.join(out, "kernel_stats.csv"))
line = None
for p in ("stats", "fetch", "write", "sqa", "sqb"):
    lf = os.path.join(out, p + "_bench.json.log")
    if os.path.exists(lf):
        for l in open(lf):
            if l.startswith("{"):
                line = json.loads(l)
                if p == "stats":
                    json.dump(line, open(os.path.join(out, "bench_line.json"), "w"))
        if line and p == "stats":
            break
if kern is None and line:
    kern = line["roofline"]["kernel"].split(" ")[0]
tot, ms = {}, {}
for d, name in (("fetch", "pmc_FETCH_SIZE.csv"), ("write", "pmc_WRITE_SIZE.csv"), ("sqa", "pmc_SQ_a.csv"), ("sqb", "pmc_SQ_b.csv")):
    f = find(d, "*counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, os.path.join(out, name))
    n_disp = set()
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            n_disp.add(row.get("Dispatch_Id"))
    ms[d + "_dispatches"] = len(n_disp)
    kt = find(d, "*kernel_trace.csv")
    if kt:
        t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(kt)) if kern in r["Kernel_Name"]]
        ms[d + "_kernel_ms_total"] = sum(t)
launches = calls or ms.get("fetch_dispatches") or 1
j = {"workload": line["config"]["workload"] if line else wl, "kernel": kern, "launches_per_bench_run": launches,
     "kernel_ms_avg_stats_pass": kms, "kernel_share_of_gpu_time_pct": share, "counters": tot, "passes": ms,
     "source": "tools/refresh_profiles.sh %s %s: rocprofv3 --kernel-trace [--stats | --pmc ...] in separate passes -- python3 bench.py --workload %s --steps 1 --warmup 0 --cpu-baseline 0" % (tag, wl, wl)}
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    fetch, write = tot["FETCH_SIZE"], tot["WRITE_SIZE"]
    j.update({"FETCH_SIZE_kb": fetch, "WRITE_SIZE_kb": write, "raw_bytes_per_launch": (fetch + write) * 1024.0 / launches,
              "hbm_bytes_per_launch": (2 * fetch + write) * 1024.0 / launches,
              "correction": "MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies the 128-B "
                            "requests of wide coalesced streaming reads at 64 B -> read side doubled; WRITE_SIZE exact.  The fit kernels mix 8-B and "
                            "16-B per-lane loads, widths the guide marks uncalibrated, so the true value lies between raw and corrected."})
if "SQ_VALU_MFMA_BUSY_CYCLES" in tot and ms.get("sqb_kernel_ms_total"):
    cycles = ms["sqb_kernel_ms_total"] * 1e-3 * 2.4e9
    j["matrix_pipe_busy_fraction"] = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4 * cycles)
    j["mfma_f64_ops"] = tot.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / 4.0
    j["note_mfma"] = "SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles per v_mfma_f64_16x16x4_f64 (MOPS/4 x 64); busy fraction = that / (1024 SIMDs x kernel cycles at 2.4 GHz)"
if tot.get("SQ_WAVE_CYCLES"):
    j["wave_cycles_waiting_fraction"] = tot.get("SQ_WAIT_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"]
json.dump(j, open(os.path.join(out, "pmc_%s.json" % wl), "w"), indent=1)
print(json.dumps(j))
PY
for P in $PASSES; do rm -rf $OUT/$P; done
