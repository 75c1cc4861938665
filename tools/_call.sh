#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q -k "bf16 and (train or To870)" 2>&1 | tail -3
for w in 0 1 1; do echo WGRAD16=$w; T2AMD_WGRAD16=$w timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-fp32-leg --no-inference --no-optimizer-ab 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('final_loss'))"; done
