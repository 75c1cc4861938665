python tools/scratch/hammer.py 40 mm &
sleep 8
L=$GRAFT_REPO_ROOT/tacotron2_amd/lib
timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 100 1 2>&1 | tail -1 | cut -c1-200
T2AMD_LIB=$L/libtacotron2_amd_kb1wait0.so timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 100 1 2>&1 | tail -1 | cut -c1-200
T2AMD_LIB=$L/libtacotron2_amd_kb1bar.so timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 100 1 2>&1 | tail -1 | cut -c1-200
wait
