#!/usr/bin/env bash
# Developer aid (GPU box): SQ counter passes of the pass-2 kernel of bench.py.

set -uo pipefail
export TMPDIR=/tmp
out="$PWD/$1"; K="${2:-pass2_kernel<false}"
mkdir -p "$out"
B="python3 $PWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-recall --no-hard-leg"
cd /tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d "$out/v/sq" -- $B > /dev/null 2> "$out/sq.err" || { echo "sq pass failed"; tail -3 "$out/sq.err"; exit 1; }
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex "$K" --output-format csv -d "$out/v/lds" -- $B > /dev/null 2> "$out/lds.err" || { echo "lds pass failed"; tail -3 "$out/lds.err"; exit 1; }
echo done
