#!/usr/bin/env bash
# Developer aid (GPU box): PMC passes of bench.py's pass-2 prefilter kernel for several .so variants.
#   bash tools/pmc_variants.sh out_dir lib1.so lib2.so ...
set -uo pipefail
export TMPDIR=/tmp
out="$1"; shift
mkdir -p "$out"
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-recall"
K="pass2_kernel<false"
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  export LMI_LIB="$PWD/$lib"
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$K" --output-format csv -d "$out/$tag/fetch" -- $B > /dev/null 2> "$out/$tag.fetch.err" || { echo "fetch pass failed for $tag"; exit 1; }
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d "$out/$tag/sq" -- $B > /dev/null 2> "$out/$tag.sq.err" || { echo "sq pass failed for $tag"; exit 1; }
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ --kernel-include-regex "$K" --output-format csv -d "$out/$tag/tcc" -- $B > /dev/null 2> "$out/$tag.tcc.err" || { echo "tcc pass failed for $tag"; exit 1; }
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM --kernel-include-regex "$K" --output-format csv -d "$out/$tag/lds" -- $B > /dev/null 2> "$out/$tag.lds.err" || { echo "lds pass failed for $tag"; exit 1; }
  echo "done $tag"
done
