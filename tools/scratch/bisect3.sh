python tools/scratch/hammer.py 60 mm &
sleep 8
timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 300 2>&1 | tail -1 | cut -c1-1500
timeout 60 python tools/scratch/stress_attn_bwd.py 0 5 150 300 2>&1 | tail -1 | cut -c1-1500
timeout 60 python tools/scratch/stress_attn_bwd.py 1 3 23 300 2>&1 | tail -1 | cut -c1-600
wait
