#!/usr/bin/env bash
# Developer aid (GPU box): bench.py on a list of shapes "n:d:leaves:nb:nq" (looking for shapes a kernel handles badly).
#   bash tools/bench_shape_sweep.sh out.txt "4000000:768:120:4:10000 ..." [extra bench flags]
set -uo pipefail
out="$1"; shapes="$2"; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
for sh in $shapes; do
  IFS=: read -r n d leaves nb nq <<< "$sh"
  timeout -k 10 300 python3 bench.py --config c2 --n "$n" --d "$d" --leaves "$leaves" --nb "$nb" --nq "$nq" --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-hard-leg --no-exact-leg --no-other-configs "$@" 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln)
        r = j['roofline']
        print('%-28s %8.3f M q/s  %8.4f ms/step | %s | %s %.4f ms frac %.4f' % ('$sh', j['value'] / 1e6, j['ms_per_step'], ' '.join('%s %.3f' % kv for kv in j['phases_ms'].items()), r.get('kernel'), r.get('avg_launch_ms', 0), r['frac']))
" >> "$out" || { echo "$sh failed" >> "$out"; }
done
cat "$out"
