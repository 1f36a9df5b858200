#!/usr/bin/env bash
# Developer aid (GPU box): A/B of pass 2's graded work items on the C2 + hard legs.  run NAME [VAR=value ...]
run() {
  name=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-exact-leg --no-other-configs 2> gpurun_out/p2s_$name.err | python3 -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);p=j['phases_ms'];print('$name','step',j['ms_per_step'],'resident',j['resident']['ms_per_step'],'p1',p['pf_sample'],'p2',p['pf_emit'],'tail',p['rescore'],'p2+tail',round(p['pf_emit']+p['rescore'],4),'hard',j['legs']['hard']['ms'],j['legs']['hard']['kernel_ms'])"
  grep "p2 ends" gpurun_out/p2s_$name.err || true
}
run off LMI_P2_GRADED=0 &&
run on LMI_P2_GRADED=1 &&
run off2 LMI_P2_GRADED=0 &&
run on2 LMI_P2_GRADED=1 &&
run off3 LMI_P2_GRADED=0 &&
run on3 LMI_P2_GRADED=1
