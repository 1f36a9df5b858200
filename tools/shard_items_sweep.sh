#!/usr/bin/env bash
# Developer aid (GPU box): pass 2's item lengths on an emulated 1/8 shard (bench.py --emulate-shard).  run NAME SHARD [VAR=value ...]
run() {
  name=$1; sh=$2; shift 2
  env "$@" timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --emulate-shard $sh/8 --no-cpu-baseline --no-recall --no-hard-leg 2> gpurun_out/shs_$name.err | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);p=j['phases_ms'];print('$name shard $sh','step',j['ms_per_step'],'resident',j['resident']['ms_per_step'],'p1',p['pf_sample'],'p2',p['pf_emit'],'tail',p['rescore'])"
}
for sh in 0 7; do
run off $sh LMI_P2_GRADED=0 &&
run default $sh LMI_P2_GRADED=1 &&
run a $sh LMI_P2_CHUNKS=1024,512,256 &&
run b $sh LMI_P2_CHUNKS=1024,256,256 &&
run c $sh LMI_P2_CHUNKS=2048,512,256 &&
run d $sh LMI_P2_CHUNKS=1024,512,256 LMI_P2_CHUNK_FRAC=0.3,0.1 || exit 1
done
