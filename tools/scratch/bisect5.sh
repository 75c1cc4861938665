python tools/scratch/hammer.py 35 mm &
sleep 8
timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 50 1 2>&1 | tail -4 | cut -c1-700
FULL_LENS=1 timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 50 1 2>&1 | tail -4 | cut -c1-700
wait
