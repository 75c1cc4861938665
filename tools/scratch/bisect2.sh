python tools/scratch/hammer.py 100 mm &
sleep 8
export POISON_LAUNCHES=0
for f in abwd0 fold0 abwd0,fold0 fwdp0,afwd0 encp0; do STRESS_TAG=$f timeout 60 python tools/stress_lds_poison.py fp32 20 $f 2>&1 | tail -1 | cut -c1-330; done
wait
