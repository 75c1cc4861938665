#!/bin/bash
# round-2 call A: every GPU test (new full-size parity + DP tests included), the bench line, inference kernel stats
tag=${1:-r02_a}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=15 > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_pytest_gpu.log
tail -30 $out/${tag}_pytest_gpu.log
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 1500 $out/${tag}_bench.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o infer -- python $GRAFT_REPO_ROOT/tools/bench_infer.py --precision bf16 > $GRAFT_REPO_ROOT/$out/${tag}_infer_under_rocprof.txt 2>&1 )
find /tmp/prof_inf -name '*kernel_stats.csv' -exec cp {} $out/${tag}_infer_kernel_stats_bf16.csv \;
head -12 $out/${tag}_infer_kernel_stats_bf16.csv | cut -c1-160
ls -la $out | tail -20
