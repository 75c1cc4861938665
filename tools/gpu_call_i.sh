#!/bin/bash
tag=${1:-r02_i}
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -x -q -k "fused or full_size or two_stream" > $out/${tag}_pytest.log 2>&1; echo "rc=$?" >> $out/${tag}_pytest.log; tail -5 $out/${tag}_pytest.log
for i in 1 2 3; do timeout 300 python -m pytest tests/test_parity_gpu.py -x -q -k "full_size" 2>&1 | tail -1; done
A="--steps 16 --warmup 3 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab"
run() { python bench.py $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), 'ms/step', round(d['value']))" | tee -a $out/r02_i_fused_bwd_after_fix.txt; }
run "fused attention bwd (drained), delay 16"
T2AMD_ATTN_FUSED_BWD=0 run "two-launch attention bwd             "
T2AMD_ATTN_FUSED_DELAY=0 run "fused attention bwd (drained), delay 0 "
T2AMD_ATTN_FUSED_BWD=0 run "two-launch attention bwd (again)     "
run "fused attention bwd (drained), delay 16 (again)"
