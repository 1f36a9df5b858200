#!/usr/bin/env bash
# GPU box: rocprofv3 kernel traces of short bench runs and the per-kernel step timeline (tools/trace_gaps.py) of each.
#   bash tools/trace_steps.sh <tag> "<config> <timing-level>" ["<config> <timing-level>" ...]
set -uo pipefail
export TMPDIR=/tmp
tag="$1"; shift
root="$PWD"; out="$root/gpurun_out/steps_$tag"
mkdir -p "$out"
for spec in "$@"; do
  set -- $spec; cfg="$1"; lvl="$2"
  B="python3 $root/bench.py --config $cfg --steps 12 --warmup 3 --timing-level $lvl --no-cpu-baseline --no-recall --no-hard-leg --no-other-configs --no-exact-leg"
  d="$out/${cfg}_t${lvl}"
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$d" -- $B > "$d.json" 2> "$d.err") || { echo "trace $spec failed"; tail -5 "$d.err"; exit 1; }
  python3 tools/trace_gaps.py "$d" > "$out/${cfg}_t${lvl}_timeline.txt" 2>&1 || true
  python3 - "$d.json" <<'PY' >> "$out/summary.txt"
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], j["value"], j["ms_per_step"], j.get("resident", {}).get("ms_per_step"), j["phases_ms"])
except Exception as e:
    print(sys.argv[1], "unparsed", e)
PY
  rm -rf "$d"
  echo "done $spec"
done
