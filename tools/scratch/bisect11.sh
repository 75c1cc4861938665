L=$GRAFT_REPO_ROOT/tacotron2_amd/lib
INPROC=30 T2AMD_LIB=$L/libtacotron2_amd_kb1np.so timeout 100 python tools/scratch/stress_attn_bwd.py 0 3 23 600 1 2>&1 | tail -1 | cut -c1-150
INPROC=30 timeout 100 python tools/scratch/stress_attn_bwd.py 1 3 23 600 1 2>&1 | tail -1 | cut -c1-150
INPROC=30 timeout 100 python tools/scratch/stress_attn_bwd.py 0 3 23 600 1 m16 2>&1 | tail -1 | cut -c1-150
INPROC=10 timeout 100 python tools/scratch/stress_attn_bwd.py 0 64 177 100 1 2>&1 | tail -1 | cut -c1-150
INPROC=10 timeout 100 python tools/scratch/stress_attn_bwd.py 1 64 177 100 1 2>&1 | tail -1 | cut -c1-150
