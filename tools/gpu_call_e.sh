#!/bin/bash
# round-2 call E: full GPU suite, the new bench line, 2-rank gloo bench, hipGraph decode probe, PMC traffic, kernel stats
tag=${1:-r02_e}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=8 > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_pytest_gpu.log
tail -22 $out/${tag}_pytest_gpu.log
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"; tail -c 2500 $out/${tag}_bench.json; tail -3 $out/${tag}_bench.err
timeout 300 python tools/microbench_decode_graph.py > $out/${tag}_decode_graph.json 2> $out/${tag}_decode_graph.err; cat $out/${tag}_decode_graph.json; tail -2 $out/${tag}_decode_graph.err
T2AMD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 --no-inference --cpu-sample 0 --no-fp32-leg > $out/${tag}_bench_2ranks_gloo.json 2> $out/${tag}_bench_2ranks_gloo.err; echo "2-rank rc=$?"; tail -c 900 $out/${tag}_bench_2ranks_gloo.json; tail -3 $out/${tag}_bench_2ranks_gloo.err
BARGS="--steps 1 --warmup 0 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py $BARGS > /dev/null 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py $BARGS > /dev/null 2>&1 )
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $out/${tag}_pmc_traffic_bf16.csv bf16; cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json
head -8 $out/${tag}_pmc_traffic_bf16.csv | cut -c1-160
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-fp32-leg --no-inference --no-optimizer-ab > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_rocprof.json 2>/dev/null )
find /tmp/prof_bench -name '*kernel_stats.csv' -exec cp {} $out/${tag}_kernel_stats_bf16.csv \;
head -14 $out/${tag}_kernel_stats_bf16.csv | cut -c1-150
