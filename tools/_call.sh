#!/bin/bash
out=gpurun_out; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q -k "transpose_cast or (bf16 and To870)" 2>&1 | tail -3
timeout 300 python tools/microbench_gemm16.py 2>&1 | grep -o "transpose-cast.*" | head -3
for w in 1 1; do T2AMD_CONV16=$w timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-fp32-leg --no-inference --no-optimizer-ab 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('final_loss'))"; done
