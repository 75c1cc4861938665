#!/bin/bash
# A/B of two BUILDS of the library on the BASELINE configs[1] training step (python -m tacotron2_amd.build --variant <tag>
# <defines...> makes lib/libtacotron2_amd_<tag>.so): alternating short bench.py runs, ms per step and the final loss of
# each (same batches, same dropout seeds: equal losses = the variants compute the same thing).
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/ab_lib.sh epifirst 3'
tag=${1:?variant tag}; rounds=${2:-3}
lib=$GRAFT_REPO_ROOT/tacotron2_amd/lib/libtacotron2_amd_${tag}.so
[ -f "$lib" ] || { echo "missing $lib"; exit 1; }
args="--steps 10 --warmup 3 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab"
for r in $(seq 1 $rounds); do
  for which in product $tag; do
    if [ $which = product ]; then out=$(python bench.py $args 2>/dev/null); else out=$(T2AMD_LIB=$lib python bench.py $args 2>/dev/null); fi
    echo "$which $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms_per_step %.3f final_loss %.9f" % (d["ms_per_step"], d["final_loss"]))')"
  done
done
