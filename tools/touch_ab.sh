#!/usr/bin/env bash
# Developer aid (GPU box): the narrow tiles' touch prefetch (lmi_pass2.h TOUCH) against the build before it (vb/head.so) on C1, a 1 000-query batch on the C2
# index, and C2 itself (whose pass 1 runs narrow tiles on the primary columns)
for lib in vb/head.so learnedmetricindex_amd/liblmi_hip.so vb/head.so learnedmetricindex_amd/liblmi_hip.so; do
  for cfg in "--config c1" "--config c2 --nq 1000" "--config c2"; do
  LMI_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py $cfg --steps 30 --warmup 6 --no-cpu-baseline --no-recall --no-exact-leg --no-hard-leg --no-other-configs 2>/dev/null | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);p=j['phases_ms'];print('$lib', '$cfg', 'step',j['ms_per_step'],'p1',p['pf_sample'],'p2',p['pf_emit'],'tail',p['rescore'])" || exit 1
  done
done
