#!/usr/bin/env bash
# Developer aid (GPU box): bench.py on one config's shape at several dimensionalities (looking for shapes a kernel handles badly).
#   bash tools/bench_d_sweep.sh out.txt CONFIG N "D1 D2 ..." [extra bench flags]
set -uo pipefail
out="$1"; cfg="$2"; n="$3"; ds="$4"; shift 4
mkdir -p "$(dirname "$out")"; : > "$out"
for D in $ds; do
  timeout -k 10 300 python3 bench.py --config "$cfg" --n "$n" --d "$D" --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-hard-leg --no-exact-leg --no-other-configs "$@" 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln)
        r = j['roofline']
        print('d %5d  %.3f M q/s  %.4f ms/step | %s | %s %.4f ms frac %.4f' % ($D, j['value'] / 1e6, j['ms_per_step'], ' '.join('%s %.3f' % kv for kv in j['phases_ms'].items()), r.get('kernel'), r.get('avg_launch_ms', 0), r['frac']))
" >> "$out" || { echo "d=$D failed" >> "$out"; }
done
cat "$out"
