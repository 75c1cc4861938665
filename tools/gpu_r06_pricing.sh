#!/bin/bash
# Round 6: the two ceilings the verdict asked to be priced before building, one gpurun call.
#   item 6 (two phase-shifted half-batches in the persistent forward): the same launch at B = 32 -- is a time step still ~18 us?
#   item 5 (decoder-LSTM chain off the backward's critical path): the chain without cell_d and the K = 4 Hd dgrad half (NOT legal
#           arithmetic: T2AMD_BWD_TIMING_NO_D=1), at split-K 2 and 4
out=gpurun_out; mkdir -p $out
args="--steps 6 --warmup 2 --cpu-sample 0 --no-fp32-leg --no-inference --no-optimizer-ab"
pick='import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(json.dumps({"ms_per_step": d["ms_per_step"], "value": d["value"], "chain": {k: {"avg_launch_us": v["avg_launch_us"], "us_per_time_step": v["us_per_time_step"], "launches": v["launches"]} for k, v in r["chain"].items()}}))'
echo "{" > $out/r06_e_ceilings.json
echo "\"B64_product\": $(python bench.py $args 2>/dev/null | python -c "$pick")," >> $out/r06_e_ceilings.json
echo "\"B32_forward_pricing\": $(python bench.py --batch-size 32 $args 2>/dev/null | python -c "$pick")," >> $out/r06_e_ceilings.json
echo "\"B64_bwd_no_d_split2\": $(T2AMD_BWD_TIMING_NO_D=1 python bench.py $args 2>/dev/null | python -c "$pick")," >> $out/r06_e_ceilings.json
echo "\"B64_bwd_no_d_split4\": $(T2AMD_BWD_TIMING_NO_D=1 T2AMD_DGRAD_SPLIT=4 python bench.py $args 2>/dev/null | python -c "$pick")," >> $out/r06_e_ceilings.json
echo "\"B64_product_split4\": $(T2AMD_DGRAD_SPLIT=4 python bench.py $args 2>/dev/null | python -c "$pick")" >> $out/r06_e_ceilings.json
echo "}" >> $out/r06_e_ceilings.json
cat $out/r06_e_ceilings.json
