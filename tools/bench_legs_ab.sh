#!/usr/bin/env bash
# Developer aid (GPU box): the default bench legs (C2, hard, C1, C5) for several .so variants, R rounds alternating.
#   bash tools/bench_legs_ab.sh out.txt R lib1.so lib2.so ...
set -uo pipefail
out="$1"; R="$2"; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
for r in $(seq 1 "$R"); do
  for lib in "$@"; do
    LMI_LIB="$PWD/$lib" timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-exact-leg --no-recall 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln)
        oc = j.get('other_configs') or {}
        h = j.get('hard_leg') or {}
        def ph(d): return ' '.join('%s %.3f' % (k, d[k]) for k in ('pf_sample', 'pf_emit', 'rescore') if k in d)
        print('%-12s C2 %.3f M (%s) | hard %.3f M (%s) | %s' % ('$(basename $lib .so)', j['value'] / 1e6, ph(j['phases_ms']), h.get('value', 0) / 1e6, ph(h.get('phases_ms', {})),
              ' | '.join('%s %.3f M (%s)' % (k, v['value'] / 1e6, ph(v.get('phases_ms', {}))) for k, v in oc.items())))
" >> "$out"
  done
done
cat "$out"
