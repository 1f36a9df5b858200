#!/usr/bin/env bash
# Developer aid (GPU box): FETCH_SIZE, SQ and TCC passes of one kernel of bench.py (regex in $2).
#   bash tools/pmc_kernel.sh out_dir 'select_rescore_kernel' [extra bench flags]
set -uo pipefail
export TMPDIR=/tmp
out="$1"; K="$2"; shift 2
mkdir -p "$out"
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-recall --no-hard-leg $*"
timeout -k 10 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$K" --output-format csv -d "$out/k/fetch" -- $B > /dev/null 2> "$out/fetch.err" || { echo "fetch pass failed"; exit 1; }
timeout -k 10 90 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d "$out/k/sq" -- $B > /dev/null 2> "$out/sq.err" || { echo "sq pass failed"; exit 1; }
timeout -k 10 90 rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-include-regex "$K" --output-format csv -d "$out/k/tcc" -- $B > /dev/null 2> "$out/tcc.err" || { echo "tcc pass failed"; exit 1; }
echo done
