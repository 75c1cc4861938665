#!/bin/bash
# Round 6 final evidence set (one gpurun call):  /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/gpu_r06_final.sh r06_z'
tag=${1:-r06_z}
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
bash tools/round_gpu_check.sh $tag
bash tools/pmc_counters.sh $tag > /dev/null 2>&1; head -12 $out/${tag}_pmc_counters_bf16.csv | cut -c1-200
# the driver's own command
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_cmd.json 2> $out/${tag}_bench_driver_cmd.err; echo "driver-cmd bench rc=$?"; cut -c1-300 $out/${tag}_bench_driver_cmd.json
# the accurate-fast mode as the timed mode, and its kernel summary
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x3 -o x3 -- python $GRAFT_REPO_ROOT/bench.py --precision bf16x3 --steps 4 --warmup 1 --cpu-sample 0 --no-fp32-leg --no-inference --no-optimizer-ab > $GRAFT_REPO_ROOT/$out/${tag}_bench_bf16x3_under_rocprof.json 2>/dev/null )
find /tmp/prof_x3 -name '*kernel_stats.csv' -exec cp {} $out/${tag}_kernel_stats_bf16x3.csv \;
head -8 $out/${tag}_kernel_stats_bf16x3.csv | cut -c1-150
timeout 300 python tools/bench_infer.py --precision bf16x3 > $out/${tag}_bench_infer_bf16x3.txt 2>&1; grep "^config\|^B" $out/${tag}_bench_infer_bf16x3.txt
python tools/scan_pk_overlap.py > $out/${tag}_pk_sites.txt 2>&1; tail -2 $out/${tag}_pk_sites.txt
