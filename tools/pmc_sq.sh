#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of bench.py's fit kernel (two rocprofv3 --pmc passes), summed over the
# launch -> gpurun_out/pmc_sq.json with the matrix-pipe utilisation derived from them.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 0 --cpu-baseline 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/a -o out --output-format csv -- $CMD > $OUT/a.log 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/b -o out --output-format csv -- $CMD > $OUT/b.log 2> $OUT/b.err
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
tot, ms = {}, None
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "gm_cv_kernel" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "gm_cv_kernel" in row["Kernel_Name"]:
            ms = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
n_simd = 256 * 4
cycles = ms * 1e-3 * 2.4e9
j = {"kernel": "gm_cv_kernel", "kernel_ms": ms, "counters": tot,
     "mfma_ops": tot.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / 4.0,
     "matrix_pipe_busy_fraction": tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (n_simd * cycles),
     "wave_cycles_waiting_fraction": (tot.get("SQ_WAIT_INST_ANY", 0) / tot["SQ_WAVE_CYCLES"]) if tot.get("SQ_WAVE_CYCLES") else None,
     "note": "SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles per v_mfma_f64_16x16x4_f64 (checked: MOPS/4 x 64); busy fraction = that / (1024 SIMDs x kernel cycles at 2.4 GHz); SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY are in units of 4 cycles"}
json.dump(j, open(os.path.join(os.path.dirname(out), "pmc_sq.json"), "w"), indent=1)
print(json.dumps(j))
PY
rm -rf $OUT/a $OUT/b
