#!/usr/bin/env bash
# Developer aid (GPU box): SQ / LDS counter passes of the query-resident pass-2 kernel of bench.py.
#   bash tools/pmc_qr.sh out_dir [extra bench flags]
set -uo pipefail
export TMPDIR=/tmp
out="$1"; shift
K="pass2_qr_kernel"
mkdir -p "$out"
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-recall --no-hard-leg $*"
timeout -k 10 90 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d "$out/k/sq" -- $B > /dev/null 2> "$out/sq.err" || { echo "sq pass failed"; tail -3 "$out/sq.err"; exit 1; }
timeout -k 10 90 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --kernel-include-regex "$K" --output-format csv -d "$out/k/lds" -- $B > /dev/null 2> "$out/lds.err" || { echo "lds pass failed"; tail -3 "$out/lds.err"; exit 1; }
timeout -k 10 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT TCC_MISS TCC_REQ --kernel-include-regex "$K" --output-format csv -d "$out/k/tcc" -- $B > /dev/null 2> "$out/tcc.err" || { echo "tcc pass failed"; tail -3 "$out/tcc.err"; exit 1; }
python3 tools/pmc_table.py "$out" "$K" | tee "$out/table.txt"
