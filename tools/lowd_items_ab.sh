#!/usr/bin/env bash
# Developer aid (GPU box): graded pass-2 items on / off at d = 45 .. 128 (10M rows, 256 leaves: bench.py --config c5 --d D)
for dd in 45 64 96 128; do
for g in 0 1; do
LMI_P2_GRADED=$g timeout -k 10 200 python3 bench.py --config c5 --d $dd --steps 40 --warmup 8 --no-cpu-baseline --no-recall --no-hard-leg --no-other-configs --no-exact-leg 2>/dev/null | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);p=j['phases_ms'];print('d $dd graded $g','step',j['ms_per_step'],'p1',p['pf_sample'],'p2',p['pf_emit'],'tail',p['rescore'])" || exit 1
done; done
