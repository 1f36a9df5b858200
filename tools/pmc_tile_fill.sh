#!/usr/bin/env bash
# Developer aid (GPU box): one SQ counter pass of pass 2 under tools/tile_fill.py (every bucket exactly M queries) for several .so variants:
# held clock (GRBM_GUI_ACTIVE / 8 / duration), matrix-pipe busy share, wave time parked.
#   bash tools/pmc_tile_fill.sh out_dir M lib1.so lib2.so ...   then   python3 tools/pmc_table.py out_dir
set -uo pipefail
export TMPDIR=/tmp
out="$1"; M="$2"; shift 2
mkdir -p "$out"
K="pass2_kernel<false"
for lib in "$@"; do
  tag=$(basename "$lib" .so)_m$M
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d "$out/$tag/sq" -- python3 tools/tile_fill.py "$lib" --m "$M" > "$out/$tag.out" 2> "$out/$tag.sq.err" || { echo "sq pass failed for $tag"; tail -n 5 "$out/$tag.sq.err"; exit 1; }
  echo "done $tag"
done
