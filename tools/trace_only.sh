#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace + stats of a short bench; prints the lmi kernels' average durations.   bash tools/trace_only.sh out_tag [bench flags]
set -uo pipefail
export TMPDIR=/tmp
tag="$1"; shift
root="$PWD"; out="$root/gpurun_out/trace_$tag"
mkdir -p "$out"
B="python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-hard-leg $*"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B > "$out/bench.json" 2> "$out/trace.err" || { echo "trace pass failed"; tail -5 "$out/trace.err"; exit 1; }
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "lmi::" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print(f'{r["Name"][:84]:84s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
