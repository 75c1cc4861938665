#!/bin/bash
# alternating A/B of an environment switch on the training step
var=$1; rounds=${2:-2}
args="--steps 10 --warmup 3 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab"
for r in $(seq 1 $rounds); do
  for val in 1 0; do
    out=$(env $var=$val python bench.py $args 2>/dev/null)
    echo "$var=$val $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.3f final_loss %.9f" % (d["ms_per_step"], d["final_loss"]))')"
  done
done
