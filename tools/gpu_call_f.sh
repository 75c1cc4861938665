#!/bin/bash
# round-2 call F: new tests, B=256 / B=64 inference profiles (per config), mel front end under the profiler
tag=${1:-r02_f}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz1_hotpath_extra_gpu.py tests/test_zz5_fullsize_parity_gpu.py::test_batched_inference_config5_lengths tests/test_parity_gpu.py::test_bf16_compute_mode_batched_inference tests/test_zz9_dp_gpu.py -x -q > $out/${tag}_pytest.log 2>&1; echo "rc=$?" >> $out/${tag}_pytest.log; tail -6 $out/${tag}_pytest.log
for cfg in config5_B256 B64; do
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o i -- python $GRAFT_REPO_ROOT/tools/bench_infer.py --precision bf16 --only $cfg > $GRAFT_REPO_ROOT/$out/${tag}_infer_${cfg}.txt 2>&1 )
find /tmp/prof_$cfg -name '*kernel_stats.csv' -exec cp {} $out/${tag}_infer_${cfg}_kernel_stats_bf16.csv \;
grep "^$cfg" $out/${tag}_infer_${cfg}.txt; head -11 $out/${tag}_infer_${cfg}_kernel_stats_bf16.csv | cut -c1-140
done
timeout 300 python tools/microbench_audio.py > $out/${tag}_audio.json 2> $out/${tag}_audio.err; cat $out/${tag}_audio.json; tail -2 $out/${tag}_audio.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_audio -o audio -- python $GRAFT_REPO_ROOT/tools/microbench_audio.py > /dev/null 2>&1 )
find /tmp/prof_audio -name '*kernel_stats.csv' -exec cp {} $out/${tag}_kernel_stats_audio.csv \;
head -8 $out/${tag}_kernel_stats_audio.csv | cut -c1-140
