#!/bin/bash
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only, separate runs) of one training step plus a FETCH_SIZE
# pass of the access-shape calibration probe, post-processed into profiles/pmc_traffic.json (stamped with the SHA-1 of all
# kernel sources) and a per-kernel CSV:
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_pass.sh r03_a'
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cmd="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab"
[ -x $GRAFT_REPO_ROOT/tools/probe/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/probe/fetch_calib.hip -o $GRAFT_REPO_ROOT/tools/probe/fetch_calib
( cd /tmp && rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_c /tmp/pmc_cw
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- $cmd > /dev/null 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- $cmd > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_c -o c -- $GRAFT_REPO_ROOT/tools/probe/fetch_calib > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_cw -o c -- $GRAFT_REPO_ROOT/tools/probe/fetch_calib > /dev/null 2>&1 )
cd $GRAFT_REPO_ROOT && python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $out/${tag}_pmc_traffic_bf16.csv bf16 /tmp/pmc_c /tmp/pmc_cw && cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json
head -12 $out/${tag}_pmc_traffic_bf16.csv | cut -c1-160
