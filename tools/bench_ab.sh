#!/usr/bin/env bash
# Developer aid (GPU box): the C2 bench (no CPU baseline / recall / other legs) for several .so variants, alternating, R rounds.
#   bash tools/bench_ab.sh out.txt R lib1.so lib2.so ...
set -uo pipefail
out="$1"; R="$2"; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
for r in $(seq 1 "$R"); do
  for lib in "$@"; do
    LMI_LIB="$PWD/$lib" timeout -k 10 240 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-recall --no-other-configs ${BENCH_AB_FLAGS:-} 2> /tmp/bench_ab.err | python3 -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if not ln.startswith('{'): continue
    j = json.loads(ln)
    ph = j.get('phases_ms', {})
    hl = j.get('hard_leg') or {}
    print('%-14s %.3f M q/s  %.3f ms/step | %s | roofline frac %.4f | hard %s' % ('$(basename $lib .so)', j['value'] / 1e6, j['ms_per_step'], ' '.join('%s %.3f' % (k, v) for k, v in ph.items()), j['roofline']['frac'], ('%.3f M q/s' % (hl['value'] / 1e6)) if hl else '-'))
" >> "$out" || { echo "bench failed for $lib"; tail -n 5 /tmp/bench_ab.err; }
  done
done
cat "$out"
