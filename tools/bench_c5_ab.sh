#!/usr/bin/env bash
# Developer aid (GPU box): bench.py --config c5 for several .so variants, R rounds alternating.   bash tools/bench_c5_ab.sh out.txt R lib1.so ...
set -uo pipefail
out="$1"; R="$2"; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
for r in $(seq 1 "$R"); do
  for lib in "$@"; do
    LMI_LIB="$PWD/$lib" timeout -k 10 200 python3 bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_AB_FLAGS:-} 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln)
        print('%-12s %.2f M q/s  %.4f ms/step | recall %s | %s | frac %.4f | %s' % ('$(basename $lib .so)', j['value'] / 1e6, j['ms_per_step'], j['recall_at_10'], ' '.join('%s %.3f' % kv for kv in j['phases_ms'].items()), j['roofline']['frac'], j['prefilter']))
" >> "$out"
  done
done
cat "$out"
