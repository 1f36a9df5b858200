#!/usr/bin/env bash
# Developer aid (GPU box): one SQ counter pass of bench.py's pass-2 prefilter kernel for several .so variants.
#   bash tools/pmc_one.sh out_dir lib1.so lib2.so ...
set -uo pipefail
export TMPDIR=/tmp
out="$1"; shift
mkdir -p "$out"
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-recall --no-hard-leg"
K="pass2_kernel<false"
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  export LMI_LIB="$PWD/$lib"
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d "$out/$tag/sq" -- $B > /dev/null 2> "$out/$tag.sq.err" || { echo "sq pass failed for $tag"; exit 1; }
  echo "done $tag"
done
