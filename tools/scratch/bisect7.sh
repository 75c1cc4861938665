INPROC=30 timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 100 1 2>&1 | tail -1 | cut -c1-160
python tools/scratch/hammer.py 25 ew &
sleep 8
timeout 60 python tools/scratch/stress_attn_bwd.py 0 3 23 100 1 2>&1 | tail -1 | cut -c1-160
wait
