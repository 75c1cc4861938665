run() { tag=$1; shift; for i in 1 2 3; do rm -f gpurun_out/dp_unequal_repeats_rank0_*.json; env "$@" T2AMD_DP_REPEATS=300 timeout 280 python -m pytest tests/test_zz9_dp_gpu.py -q -x -k "real_engine and $PREC" > /tmp/o.txt 2>&1; tail -1 /tmp/o.txt; for f in gpurun_out/dp_unequal_repeats_rank0_*.json; do [ -f $f ] && cp $f gpurun_out/dpf_${tag}_$i.json; done; done; }
PREC=fp32 run fp32 X=1
PREC=bf16 run bf16 X=1
PREC=fp32 run fp32nowide T2AMD_FP32_WIDE=0
