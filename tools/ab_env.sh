#!/bin/bash
# A/B of ONE build under two environments (a run-time switch of the library): alternating short bench.py runs, ms per step
# and the final loss of each.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/ab_env.sh T2AMD_ELEMENTWISE_SCALAR=1 3'
setting=${1:?VAR=value}; rounds=${2:-3}
args="--steps 10 --warmup 3 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab"
for r in $(seq 1 $rounds); do
  for which in product "$setting"; do
    if [ "$which" = product ]; then out=$(python bench.py $args 2>/dev/null); else out=$(env "$setting" python bench.py $args 2>/dev/null); fi
    echo "$which $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.3f final_loss %.9f" % (d["ms_per_step"], d["final_loss"]))')"
  done
done
