#!/bin/bash
# One gpurun call that (re)establishes the GPU evidence of a round:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/round_gpu_check.sh r02_a'
# Writes everything under gpurun_out/<tag>_*; copy what should be judged into profiles/.
# rocprofv3: kernel-trace/stats and --pmc passes are separate runs (never combined with other trace domains).
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_pytest_gpu.log
tail -5 $out/${tag}_pytest_gpu.log
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 600 $out/${tag}_bench.json
python bench.py --fused-optimizer --no-inference --no-fp32-leg --cpu-sample 0 --no-roofline > $out/${tag}_bench_fused_optimizer.json 2>> $out/${tag}_bench.err; tail -c 400 $out/${tag}_bench_fused_optimizer.json
# beyond the BASELINE config: the serial chain is per TIME STEP, so a larger per-GPU batch amortises it (288 GB of HBM holds B=256 easily)
timeout 600 python bench.py --batch-size 256 --steps 3 --warmup 1 --no-inference --no-fp32-leg --cpu-sample 0 --no-roofline > $out/${tag}_bench_B256.json 2>> $out/${tag}_bench.err; tail -c 400 $out/${tag}_bench_B256.json
timeout 120 python tools/microbench_barrier.py > $out/${tag}_barrier.txt 2>&1; cat $out/${tag}_barrier.txt
timeout 300 python tools/microbench_audio.py > $out/${tag}_audio.json 2> $out/${tag}_audio.err; cat $out/${tag}_audio.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-fp32-leg --no-inference > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_rocprof.json 2>/dev/null )
find /tmp/prof_bench -name '*kernel_stats.csv' -exec cp {} $out/${tag}_kernel_stats_bf16.csv \;
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_audio -o audio -- python $GRAFT_REPO_ROOT/tools/microbench_audio.py > /dev/null 2>&1 )
find /tmp/prof_audio -name '*kernel_stats.csv' -exec cp {} $out/${tag}_kernel_stats_audio.csv \;
ls -la $out | tail -12
