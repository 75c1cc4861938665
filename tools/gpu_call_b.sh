#!/bin/bash
# round-2 call B: persistent decode kernel (tests + timing), DP diagnostics
tag=${1:-r02_b}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz7_persistent_decode_gpu.py -x -q > $out/${tag}_pytest_persist.log 2>&1; echo "rc=$?" >> $out/${tag}_pytest_persist.log
tail -25 $out/${tag}_pytest_persist.log
timeout 300 python tools/bench_decode_b1.py > $out/${tag}_decode_b1.json 2> $out/${tag}_decode_b1.err; cat $out/${tag}_decode_b1.json; tail -3 $out/${tag}_decode_b1.err
timeout 300 python -m pytest tests/test_zz9_dp_gpu.py -x -q > $out/${tag}_pytest_dp.log 2>&1; echo "rc=$?" >> $out/${tag}_pytest_dp.log
grep -n "AssertionError\|passed\|failed" $out/${tag}_pytest_dp.log | head -20
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o infer -- python $GRAFT_REPO_ROOT/tools/bench_infer.py --precision bf16 > $GRAFT_REPO_ROOT/$out/${tag}_infer_under_rocprof.txt 2>&1 )
find /tmp/prof_inf -name '*kernel_stats.csv' -exec cp {} $out/${tag}_infer_kernel_stats_bf16.csv \;
head -12 $out/${tag}_infer_kernel_stats_bf16.csv | cut -c1-200
