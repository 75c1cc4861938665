REGPOISON=40 timeout 80 python tools/scratch/stress_attn_bwd.py 0 3 23 100 1 2>&1 | tail -5 | cut -c1-300
REGPOISON=40 timeout 80 python tools/scratch/stress_attn_bwd.py 1 3 23 100 1 2>&1 | tail -2 | cut -c1-300
