#!/usr/bin/env bash
# Developer aid (GPU box): SQ counter passes of the low-dimensional pass-2 kernel (bench.py --config c5).
#   bash tools/pmc_c5.sh out_dir [lib.so]     then   python3 tools/pmc_table.py out_dir "pass2_small_kernel<3, false"
set -uo pipefail
export TMPDIR=/tmp
out="$1"; lib="${2:-learnedmetricindex_amd/liblmi_hip.so}"
mkdir -p "$out"
tag=$(basename "$lib" .so)
export LMI_LIB="$PWD/$lib"
B="python3 bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-recall"
K="pass2_small_kernel<3, false"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d "$out/$tag/sq" -- $B > /dev/null 2> "$out/$tag.sq.err" || { echo "sq pass failed"; tail -n 5 "$out/$tag.sq.err"; exit 1; }
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES --kernel-include-regex "$K" --output-format csv -d "$out/$tag/insts" -- $B > /dev/null 2> "$out/$tag.insts.err" || { echo "insts pass failed"; tail -n 5 "$out/$tag.insts.err"; exit 1; }
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN --kernel-include-regex "$K" --output-format csv -d "$out/$tag/misc" -- $B > /dev/null 2> "$out/$tag.misc.err" || { echo "misc pass failed"; tail -n 5 "$out/$tag.misc.err"; }
echo "done $tag"
