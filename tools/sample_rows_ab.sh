#!/usr/bin/env bash
# Developer aid (GPU box): pass 1's small-bucket sampling threshold (LMI_PF_SAMPLE_ROWS builds under vb/) on C1 and two many-leaf shapes
for lib in learnedmetricindex_amd/liblmi_hip.so vb/sr256.so learnedmetricindex_amd/liblmi_hip.so vb/sr256.so; do
  for cfg in "--config c1" "--config c2 --n 4000000 --leaves 2000" "--config c2 --n 4000000 --leaves 500"; do
  LMI_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py $cfg --steps 40 --warmup 8 --no-cpu-baseline --no-recall --no-exact-leg --no-hard-leg --no-other-configs 2>/dev/null | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);p=j['phases_ms'];print('$lib', '$cfg', 'step',j['ms_per_step'],'p1',p['pf_sample'],'p2',p['pf_emit'],'tail',p['rescore'], 'surv', j['prefilter']['survivors_per_slot'])" || exit 1
  done
done
