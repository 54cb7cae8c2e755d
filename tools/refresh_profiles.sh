#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + the two PMC passes of bench.py's workload,
# then writes gpurun_out/prof/{kernel_stats.csv, pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv, traffic.json}.
# usage: bash tools/refresh_profiles.sh <tag>
set -e
TAG=${1:-vX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline 0"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o out --output-format csv -- $CMD > $OUT/stats_bench.json.log 2> $OUT/stats.err
echo "stats pass done" 
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o out --output-format csv -- $CMD > $OUT/fetch_bench.json.log 2> $OUT/fetch.err
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o out --output-format csv -- $CMD > $OUT/write_bench.json.log 2> $OUT/write.err
echo "write pass done"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
def find(d, pat):
    f = glob.glob(os.path.join(out, d, "**", pat), recursive=True)
    return f[0] if f else None
def counter(d, name):
    tot, ms = 0.0, None
    for row in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        if "gm_cv_kernel" in row["Kernel_Name"] and row["Counter_Name"] == name:
            tot += float(row["Counter_Value"])
    for row in csv.DictReader(open(find(d, "*kernel_trace.csv"))):
        if "gm_cv_kernel" in row["Kernel_Name"]:
            ms = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    return tot, ms
fetch, ms_f = counter("fetch", "FETCH_SIZE")
write, ms_w = counter("write", "WRITE_SIZE")
raw = (fetch + write) * 1024.0
cor = (2 * fetch + write) * 1024.0
j = {"workload": "synthetic gaussian n=1000 p=10000 nFolds=5 grid=20alpha x 100lambda Epis=no", "kernel": "gm_cv_kernel", "launches": 1,
     "FETCH_SIZE_kb": fetch, "WRITE_SIZE_kb": write, "raw_bytes": raw, "corrected_bytes": cor,
     "correction": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads -> read side doubled; WRITE_SIZE exact. The kernel mixes 8-B and 16-B per-lane loads, widths the guide marks uncalibrated, so the true value lies between raw and corrected.",
     "source": "tools/refresh_profiles.sh %s: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0" % tag,
     "kernel_ms_fetch_pass": ms_f, "kernel_ms_write_pass": ms_w}
json.dump(j, open(os.path.join(out, "traffic.json"), "w"), indent=1)
import shutil
for d, pat, name in (("stats", "*kernel_stats.csv", "kernel_stats.csv"), ("fetch", "*counter_collection.csv", "pmc_FETCH_SIZE.csv"), ("write", "*counter_collection.csv", "pmc_WRITE_SIZE.csv")):
    f = find(d, pat)
    if f: shutil.copy(f, os.path.join(out, name))
print(json.dumps(j))
PY
rm -rf $OUT/stats $OUT/fetch $OUT/write
