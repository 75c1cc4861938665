#!/bin/bash
# round-2 call C: persistent decode kernel after tuning (tests + phase clock)
tag=${1:-r02_c}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_zz7_persistent_decode_gpu.py -x -q > $out/${tag}_pytest_persist.log 2>&1; echo "rc=$?" >> $out/${tag}_pytest_persist.log
tail -5 $out/${tag}_pytest_persist.log
timeout 300 python tools/bench_decode_b1.py > $out/${tag}_decode_b1.json 2> $out/${tag}_decode_b1.err; cat $out/${tag}_decode_b1.json; tail -3 $out/${tag}_decode_b1.err
