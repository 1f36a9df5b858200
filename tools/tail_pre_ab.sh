#!/usr/bin/env bash
# Developer aid (GPU box): the tail's rank-list inputs requested ahead (lmi_tail.h / rescore_core PRE) against the build before (vb/head.so)
for lib in vb/head.so learnedmetricindex_amd/liblmi_hip.so vb/head.so learnedmetricindex_amd/liblmi_hip.so; do
  LMI_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-recall --no-exact-leg 2>/dev/null | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);o=j['other_configs'];print('$lib', 'c2',j['ms_per_step'],'tail',j['phases_ms']['rescore'],'| hard',j['hard_leg']['ms_per_step'],j['hard_leg']['phases_ms']['rescore'],'| c1',o['c1']['ms_per_step'],o['c1']['phases_ms']['rescore'],'| c5',o['c5']['ms_per_step'],o['c5']['phases_ms']['rescore'])" || exit 1
done
