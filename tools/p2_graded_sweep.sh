#!/usr/bin/env bash
# Developer aid (GPU box): A/B of pass 2's item boundary and graded chunks on the C2 + hard legs.  run NAME [VAR=value ...]
run() {
  name=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-exact-leg --no-other-configs 2> gpurun_out/p2s_$name.err | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);p=j['phases_ms'];print('$name','step',j['ms_per_step'],'p1',p['pf_sample'],'p2',p['pf_emit'],'hard',j['legs']['hard']['ms'],j['legs']['hard']['kernel_ms'])"
  grep "p2 ends" gpurun_out/p2s_$name.err || true
}
run head LMI_LIB=$PWD/vb/head.so &&
run new_off LMI_P2_GRADED=0 &&
run new_A LMI_P2_CHUNKS=2048,1024,512 &&
run new_E LMI_P2_CHUNKS=2048,1024,512 LMI_P2_CHUNK_FRAC=0.2,0.08 &&
run new_F LMI_P2_CHUNKS=2048,512,256 LMI_P2_CHUNK_FRAC=0.12,0.03 &&
run head2 LMI_LIB=$PWD/vb/head.so &&
run new_off2 LMI_P2_GRADED=0
