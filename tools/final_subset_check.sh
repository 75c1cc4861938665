#!/bin/bash
tag=r03_zd; out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz1_hotpath_extra_gpu.py tests/test_zz7_persistent_decode_gpu.py -m gpu -q -k "give_up_poisons or batched_encoder or two_or_three or route" > $out/${tag}_pytest_subset.log 2>&1; tail -2 $out/${tag}_pytest_subset.log
bash tools/pmc_pass.sh $tag | tail -2
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"; tail -c 300 $out/${tag}_bench.json
timeout 200 python tools/bench_infer.py --precision bf16 > $out/${tag}_bench_infer_bf16.txt 2>&1; grep "^config\|^B" $out/${tag}_bench_infer_bf16.txt | cut -c1-200
timeout 100 python tools/ab_encoder_batch_persistent.py 2>/dev/null | tail -1 > $out/${tag}_ab_encoder_batch_persistent.json; timeout 100 python tools/ab_encoder_batch_persistent.py --B 16 2>/dev/null | tail -1 >> $out/${tag}_ab_encoder_batch_persistent.json; timeout 100 python tools/ab_encoder_batch_persistent.py --B 256 --T 187 2>/dev/null | tail -1 >> $out/${tag}_ab_encoder_batch_persistent.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-fp32-leg --no-inference --no-optimizer-ab > $GRAFT_REPO_ROOT/$out/${tag}_bench_under_rocprof.json 2>/dev/null )
find /tmp/prof_bench -name '*kernel_stats.csv' -exec cp {} $out/${tag}_kernel_stats_bf16.csv \;
head -6 $out/${tag}_kernel_stats_bf16.csv | cut -c1-120
