#!/bin/bash
out=gpurun_out; mkdir -p $out
A="--steps 16 --warmup 3 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab"
run() { python bench.py $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), 'ms/step', round(d['value']))" | tee -a $out/r02_h_fused_bwd_delay.txt; }
run "two launches          "
for dl in 0 4 8 12 16 24; do T2AMD_ATTN_FUSED_BWD=1 T2AMD_ATTN_FUSED_DELAY=$dl run "fused bwd, delay $dl   "; done
run "two launches (again)  "
T2AMD_ATTN_FUSED_BWD=1 T2AMD_ATTN_FUSED_DELAY=4 run "fused bwd, delay 4 (again)"
T2AMD_ATTN_FUSED_BWD=1 T2AMD_ATTN_FUSED_DELAY=16 run "fused bwd, delay 16 (again)"
