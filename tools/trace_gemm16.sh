#!/bin/bash
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_t && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab > /dev/null 2>&1
f=$(find /tmp/prof_t -name '*kernel_trace.csv' | head -1)
head -1 $f > $GRAFT_REPO_ROOT/gpurun_out/r03_v_gemm16_trace.csv
grep -E "gemm16|splitk_reduce|cast_halo|gemm_bf16x3" $f >> $GRAFT_REPO_ROOT/gpurun_out/r03_v_gemm16_trace.csv
wc -l $GRAFT_REPO_ROOT/gpurun_out/r03_v_gemm16_trace.csv
