#!/bin/bash
python tools/microbench_gemm.py --only wgrad --prec 2 --splitk 1,2,3,4 --src16 --iters 6 2>&1 | tee gpurun_out/r02_j_gemm_src16.txt
