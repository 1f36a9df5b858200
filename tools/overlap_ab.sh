#!/usr/bin/env bash
# Developer aid (GPU box): the next batch's MLP on a stream of its own (LMI_PIPE_OVERLAP=1, default) or on the scan's stream (0), default legs
for ov in 1 0 1 0; do
  LMI_PIPE_OVERLAP=$ov timeout -k 10 300 python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-recall --no-exact-leg --no-hard-leg 2>/dev/null | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);o=j['other_configs'];print('overlap $ov', 'c2',j['ms_per_step'],'resident',j['resident']['ms_per_step'],'p2',j['phases_ms']['pf_emit'],'mhz',j['roofline']['shader_clock_mhz_under_kernel'],'| c1',o['c1']['ms_per_step'],'| c5',o['c5']['ms_per_step'])" || exit 1
done
