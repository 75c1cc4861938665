python tools/scratch/hammer.py 45 mm &
sleep 8
timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 200 1 2>&1 | tail -1 | cut -c1-1200
timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 200 1 m16 2>&1 | tail -1 | cut -c1-1200
timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 200 2 m16 2>&1 | tail -1 | cut -c1-1200
wait
