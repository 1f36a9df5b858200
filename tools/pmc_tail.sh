#!/usr/bin/env bash
# GPU box: instruction-cache and wait counters of tail_kernel (is its ~10 000-instruction straight-line body an instruction-fetch problem?)
#   bash tools/pmc_tail.sh "--config c5"
set -uo pipefail
export TMPDIR=/tmp
EXTRA="${1:-}"
root="$PWD"; out="$root/gpurun_out/pmc_tail"
mkdir -p "$out"
B="python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-hard-leg --no-other-configs --no-exact-leg $EXTRA"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --kernel-include-regex "tail_kernel" --output-format csv -d "$out/ic" -- $B > /dev/null 2> "$out/ic.err" || { echo "icache pass failed"; tail -3 "$out/ic.err"; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH --kernel-include-regex "tail_kernel" --output-format csv -d "$out/sq" -- $B > /dev/null 2> "$out/sq.err" || { echo "sq pass failed"; tail -3 "$out/sq.err"; exit 1; }
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("ic", "sq"):
    f = glob.glob(f"{out}/{sub}/**/*_counter_collection.csv", recursive=True)
    if not f:
        print(sub, "no counter file"); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(max(f))):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(sub, {k: round(v / max(1, n[k])) for k, v in acc.items()}, "launches", max(n.values()) if n else 0)
PY
