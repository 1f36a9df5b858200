#!/usr/bin/env bash
# Developer aid (GPU box): the host pipeline's depth (slots in flight) on the default legs.  LMI_PIPE_DEPTH = 2 (default) / 3 / 4
for dp in 2 3 4 2 3; do
  LMI_PIPE_DEPTH=$dp timeout -k 10 300 python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-recall --no-exact-leg --no-hard-leg 2>/dev/null | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('depth $dp', {k:v['ms'] for k,v in j['legs'].items()}, 'scan', j['phases_ms']['scan'])" || exit 1
done
