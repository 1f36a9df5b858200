#!/usr/bin/env bash
# Developer aid (GPU box): GPU test suite + a short headline bench, outputs under gpurun_out/$1   (bash tools/gpu_check.sh tag [tests|bench])
set -uo pipefail
out="gpurun_out/${1:-qr}"; mkdir -p "$out"
if [ "${2:-tests}" = "tests" ]; then
  timeout -k 10 400 python -m pytest tests -m gpu -x -q > "$out/tests.log" 2>&1; echo "rc=$?" >> "$out/tests.log"; tail -4 "$out/tests.log"
fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hard-leg > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
python - "$out/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["phases_ms"], "frac", j["roofline"]["frac"], "recall", j["recall_at_10"], j["prefilter"], "resident", j["resident"]["ms_per_step"])
PY
