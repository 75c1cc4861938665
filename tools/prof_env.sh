#!/bin/bash
# kernel stats of the training step under an environment switch: tools/prof_env.sh VAR tag
var=$1; tag=$2
export TMPDIR=/tmp
for val in 1 0; do
  ( cd /tmp && rm -rf /tmp/prof_$val && env $var=$val timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$val -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab > /dev/null 2>&1 )
  find /tmp/prof_$val -name '*kernel_stats.csv' -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${tag}_${var}_${val}_kernel_stats.csv \;
done
