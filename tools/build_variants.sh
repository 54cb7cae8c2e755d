#!/bin/bash
# Builds the diagnostic variants of the HIP library next to the shipped one (pareben_amd/lib):
#   libpareben_hip_prof.so   -DPAREBEN_PHASE_TIMERS  (tools/phase_profile*.py)
#   libpareben_hip_diag.so   -DPAREBEN_DIAG  (tools/ubench/*.py: single phases on their own)
#   libpareben_hip_diagprof.so   the same with -DPAREBEN_PHASE_TIMERS (DIAG_LIB=libpareben_hip_diagprof.so: per-phase ticks; the timers cost ~0.2 us each)
# usage: bash tools/build_variants.sh [prof] [diag] [diagprof]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -pthread"
for v in ${@:-prof}; do
  case $v in
    prof) D="-DPAREBEN_PHASE_TIMERS" ;;
    diag) D="-DPAREBEN_DIAG" ;;
    diagprof) D="-DPAREBEN_DIAG -DPAREBEN_PHASE_TIMERS" ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
  $HIPCC $FLAGS $D -o $R/pareben_amd/lib/libpareben_hip_$v.so $R/pareben_amd/csrc/pareben_hip.hip -ldl
  echo "built libpareben_hip_$v.so"
done
