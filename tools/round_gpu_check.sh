#!/bin/bash
# One gpurun call that (re)establishes the GPU evidence of a round:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/round_gpu_check.sh r02_k'
# Writes everything under gpurun_out/<tag>_*; copy what should be judged into profiles/ (and <tag>_pmc_traffic.json over
# profiles/pmc_traffic.json: bench.py reports its numbers as roofline.traffic while the kernel sources' SHA-1 matches).
# rocprofv3: kernel-trace/stats and --pmc passes are separate runs (never combined with other trace domains).
# SKIP_INFER_PROF=1 leaves out the two per-configuration inference profiles.
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --durations=6 > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_pytest_gpu.log
tail -12 $out/${tag}_pytest_gpu.log
bash tools/pmc_pass.sh $tag | tail -4                     # first: the bench line below then carries this build's traffic
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"; tail -c 1200 $out/${tag}_bench.json; tail -2 $out/${tag}_bench.err
# two ranks over RCCL whenever the box shows two GPUs (every box of rounds 1-4 showed one)
ngpu=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null)
if [ "${ngpu:-1}" -ge 2 ]; then
  timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 --cpu-sample 0 --no-inference --no-fp32-leg > $out/${tag}_bench_2gpu_rccl.json 2> $out/${tag}_bench_2gpu_rccl.err; echo "2-GPU RCCL bench rc=$?"; cut -c1-400 $out/${tag}_bench_2gpu_rccl.json
else
  echo "one GPU visible: no RCCL world-size-2 run" | tee $out/${tag}_bench_2gpu_rccl.txt
fi
timeout 300 python tools/bench_decode_b1.py > $out/${tag}_decode_b1.json 2> /dev/null; cut -c1-700 $out/${tag}_decode_b1.json
timeout 300 python tools/bench_infer.py --precision bf16 > $out/${tag}_bench_infer_bf16.txt 2>&1; grep "^config\|^B" $out/${tag}_bench_infer_bf16.txt
timeout 300 python tools/bench_infer.py --precision fp32 > $out/${tag}_bench_infer_fp32.txt 2>&1; grep "^config\|^B" $out/${tag}_bench_infer_fp32.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-fp32-leg --no-inference --no-optimizer-ab > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_rocprof.json 2>/dev/null )
find /tmp/prof_bench -name '*kernel_stats.csv' -exec cp {} $out/${tag}_kernel_stats_bf16.csv \;
head -12 $out/${tag}_kernel_stats_bf16.csv | cut -c1-150
if [ -z "$SKIP_INFER_PROF" ]; then
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -o b1 -- python $GRAFT_REPO_ROOT/tools/bench_infer.py --precision bf16 --only config4_B1 > /dev/null 2>&1 )
find /tmp/prof_b1 -name '*kernel_stats.csv' -exec cp {} $out/${tag}_infer_config4_B1_kernel_stats_bf16.csv \;
head -6 $out/${tag}_infer_config4_B1_kernel_stats_bf16.csv | cut -c1-150
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/tools/bench_infer.py --precision bf16 --only config5_B256 > /dev/null 2>&1 )
find /tmp/prof_c5 -name '*kernel_stats.csv' -exec cp {} $out/${tag}_infer_config5_B256_kernel_stats_bf16.csv \;
head -10 $out/${tag}_infer_config5_B256_kernel_stats_bf16.csv | cut -c1-150
fi
