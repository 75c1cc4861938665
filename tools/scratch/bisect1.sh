python tools/scratch/hammer.py 170 mm &
sleep 8
export POISON_LAUNCHES=0
STRESS_TAG=a timeout 60 python tools/stress_lds_poison.py fp32 100 separate 2>&1 | tail -1
STRESS_TAG=b timeout 60 python tools/stress_lds_poison.py bf16 100 separate 2>&1 | tail -1
STRESS_TAG=c T2AMD_FP32_WIDE=0 timeout 60 python tools/stress_lds_poison.py fp32 100 separate 2>&1 | tail -1
STRESS_TAG=d timeout 100 python tools/stress_lds_poison.py fp32 40 fused 2>&1 | tail -1
STRESS_TAG=e timeout 100 python tools/stress_lds_poison.py bf16 40 fused 2>&1 | tail -1
wait
