#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q -k "conv16 or gemm16 or (bf16 and (train or To870)) or config5" 2>&1 | tail -4
for w in 0 1 1; do echo CONV16=$w; T2AMD_CONV16=$w timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-fp32-leg --no-inference --no-optimizer-ab 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('final_loss'))"; done
for w in 0 1; do echo CONV16=$w; T2AMD_CONV16=$w timeout 200 python tools/bench_infer.py --precision bf16 --only config5_B256 2>/dev/null | grep "^config"; done
