#!/bin/bash
# extra SQ counters for one bench workload (diagnostic; names that the device does not have make a pass fail on its own)
WL=${1:-config3}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_extra_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $SET -d $OUT/$tag -o out --output-format csv -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --cpu-baseline 0 > $OUT/$tag.log 2> $OUT/$tag.err || echo "pass $tag failed"
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
tot = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "cv_kernel" in r["Kernel_Name"]:
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(tot)
PY
  fi
  rm -rf $OUT/$tag
done
