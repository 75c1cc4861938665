L=$GRAFT_REPO_ROOT/tacotron2_amd/lib
DBG=1 INPROC=30 T2AMD_LIB=$L/libtacotron2_amd_kb1dbg.so timeout 100 python tools/scratch/stress_attn_bwd.py 0 3 23 60 1 2>&1 | tail -16 | cut -c1-420
