#!/usr/bin/env bash
# Developer aid (GPU box): one-piece chunks for few-row re-rank batches (lmi_rescore.h rc_stream_batch) against the build before (vb/head.so)
for lib in vb/head.so learnedmetricindex_amd/liblmi_hip.so vb/head.so learnedmetricindex_amd/liblmi_hip.so; do
  for cfg in "--emulate-shard 0/8" "--emulate-shard 5/8" "--config c1" ""; do
  LMI_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py $cfg --steps 30 --warmup 6 --no-cpu-baseline --no-recall --no-exact-leg --no-hard-leg --no-other-configs 2>/dev/null | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);p=j['phases_ms'];print('$lib', '$cfg', 'step',j['ms_per_step'],'resident',j['resident']['ms_per_step'],'p2',p['pf_emit'],'tail',p['rescore'])" || exit 1
  done
done
