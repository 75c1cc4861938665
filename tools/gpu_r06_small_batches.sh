#!/bin/bash
# B = 2 .. 8 free-running decode: matrix-vector kernels (T2AMD_SMALL_BATCH_MAX=8) against the 64-row tiles (=3), three modes.
# usage (GPU box): bash tools/gpu_r06_small_batches.sh <tag>
tag=${1:-r06_s}
mkdir -p gpurun_out
o=gpurun_out/${tag}_bench_infer_small_batches.txt
: > $o
for prec in bf16 fp32 bf16x3; do
  for m in 8 3; do
    echo "# --precision $prec  T2AMD_SMALL_BATCH_MAX=$m" >> $o
    T2AMD_SMALL_BATCH_MAX=$m timeout 600 python tools/bench_infer.py --precision $prec --small 2>/dev/null | grep '^B' >> $o
  done
done
cat $o
