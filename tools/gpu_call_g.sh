#!/bin/bash
tag=${1:-r02_g}
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz5_fullsize_parity_gpu.py::test_batched_inference_config5_lengths tests/test_parity_gpu.py -x -q -k "inference" > $out/${tag}_pytest.log 2>&1; echo "rc=$?" >> $out/${tag}_pytest.log; tail -12 $out/${tag}_pytest.log
timeout 300 python tools/bench_infer.py --precision bf16 --only config5_B256 2>&1 | grep config5
timeout 300 python tools/bench_infer.py --precision fp32 --only config5_B256 2>&1 | grep config5
