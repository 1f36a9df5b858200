#!/usr/bin/env bash
# Developer aid (GPU box): FETCH_SIZE pass of one kernel of bench.py.   bash tools/pmc_fetch.sh out_dir kernel_regex [bench flags]
set -uo pipefail
export TMPDIR=/tmp
out="$1"; K="$2"; shift 2
mkdir -p "$out"
B="python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-recall --no-hard-leg $*"
timeout -k 10 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$K" --output-format csv -d "$out/k/fetch" -- $B > /dev/null 2> "$out/fetch.err" || { echo "fetch pass failed"; grep -v "^[WE]2026" "$out/fetch.err" | tail -3; exit 1; }
python3 tools/pmc_table.py "$out" "$K"
