#!/bin/bash
out=gpurun_out; mkdir -p $out
for cfg in "0,0,0,0,0,0,1" "16,0,0,0,0,0,1" "16,16,0,16,0,0,1" "16,16,0,16,8,8,1" "16,32,0,32,16,16,1" "16,16,8,16,8,8,1" "16,16,0,16,8,8,4" "16,16,0,16,8,8,8" "24,24,0,24,12,12,2"; do
  T2AMD_PB_DELAYS=$cfg timeout 120 python tools/bench_decode_b1.py --steps 1000 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
ph=d['persistent_phases']['first workgroup (attention team)']
w=[round(ph[k],2) for k in ph if k.startswith('wait')]
print('delays $cfg', 'us/step', round(d['persistent_bf16']['us_per_step_loop_only'],2), 'waits', w, 'chain', round(d['launch_chain_bf16']['us_per_step_loop_only'],2))
" | tee -a $out/r02_d_decode_b1_delays.txt
done
