#!/bin/bash
# PMC counters of the full-stat micro-benchmark (diagnostic): one pass per counter group
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_fs/$tag -o out --output-format csv -- python3 $R/tools/ubench/fullstat_rate.py $R/pareben_amd/lib/libpareben_hip_diag.so 256 > /dev/null 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
tot=collections.defaultdict(float)
for f in glob.glob(R+"/gpurun_out/pmc_fs/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "diag_fullstat" in row.get("Kernel_Name",""):
            tot[row["Counter_Name"]]+=float(row["Counter_Value"])
for k,v in sorted(tot.items()): print(k, "%.4g"%v)
PY
