#!/bin/bash
# One short gpurun call for the BPTT cell fold: its bitwise tests, the kernels it touched, and the A/B on the training step.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_call_fold.sh r02_w'
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -q \
    -k "folded_cells or cell_fold or fused_matches or lstm_backward or train_step_matches or two_stream or attention_forward or forms_agree" \
    > $out/${tag}_pytest_fold.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_pytest_fold.log
tail -25 $out/${tag}_pytest_fold.log
timeout 400 python tools/ab_cell_fold.py --configs "${AB_CONFIGS:-1,1,0;1,1,1}" --steps 6 --rounds 3 > $out/${tag}_ab_cell_fold.json 2> $out/${tag}_ab_cell_fold.err; echo "ab rc=$?"
cat $out/${tag}_ab_cell_fold.json; tail -5 $out/${tag}_ab_cell_fold.err
