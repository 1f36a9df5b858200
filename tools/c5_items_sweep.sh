#!/usr/bin/env bash
# Developer aid (GPU box): pass 2's item lengths at 10M x 45 (bench.py --config c5).  run NAME [VAR=value ...]
run() {
  name=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --config c5 --steps 60 --warmup 10 --no-cpu-baseline --no-recall --no-hard-leg --no-other-configs --no-exact-leg 2>/dev/null | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);p=j['phases_ms'];print('$name','step',j['ms_per_step'],'p1',p['pf_sample'],'p2',p['pf_emit'],'tail',p['rescore'])"
}
run off LMI_P2_GRADED=0 &&
run c LMI_P2_CHUNKS=8192,4096,1024 LMI_P2_CHUNK_FRAC=0.25,0.08 &&
run e LMI_P2_CHUNKS=8192,4096,2048 LMI_P2_CHUNK_FRAC=0.3,0.1 &&
run f LMI_P2_CHUNKS=8192,2048,1024 LMI_P2_CHUNK_FRAC=0.2,0.06 &&
run g LMI_P2_CHUNKS=8192,4096,1024 LMI_P2_CHUNK_FRAC=0.35,0.12 &&
run h LMI_P2_CHUNKS=6144,3072,1024 LMI_P2_CHUNK_FRAC=0.25,0.08 &&
run c2 LMI_P2_CHUNKS=8192,4096,1024 LMI_P2_CHUNK_FRAC=0.25,0.08 &&
run off2 LMI_P2_GRADED=0
