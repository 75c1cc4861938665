#!/bin/bash
# SQ / TCC counters of the chain kernels (rocprofv3 --pmc, kernel trace only, one pass per counter group):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_counters.sh r03_s'
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cmd="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  ( cd /tmp && rm -rf /tmp/pmc_g$i && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_g$i -o g -- $cmd > /dev/null 2>&1 )
done
cd $GRAFT_REPO_ROOT && python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for i in range(1, 5):
    for f in glob.glob("/tmp/pmc_g%d/**/*counter_collection.csv" % i, recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            k = row["Kernel_Name"]; acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])].add(row.get("Dispatch_Id"))
names = sorted({c for k in acc for c in acc[k]})
want = ("attn_bwd_main_kernel", "skinny_wide_kernel<true, 3,", "skinny_wide_kernel<false, 3,", "attn_fwd_fused_kernel", "gemm16_tn_kernel", "gemm16_kk_kernel",
        "dec_train_fwd_persistent_kernel", "encoder_bilstm_batch_persistent")
with open("gpurun_out/${tag}_pmc_counters_bf16.csv", "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --pmc <group> (four separate passes) of one bf16 training step; per-launch averages\n")
    fh.write("kernel,launches," + ",".join(names) + ",wait_frac,issue_stall_frac,active_frac,l2_hit_rate\n")
    for k in acc:
        if not any(w in k for w in want): continue
        n = max(len(cnt[(k, c)]) for c in acc[k]) or 1
        v = {c: acc[k][c] / max(len(cnt[(k, c)]), 1) for c in acc[k]}
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        hit, miss = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
        fh.write('"%s",%d,' % (k[:60], n) + ",".join("%.0f" % v.get(c, 0) for c in names) +
                 ",%.3f,%.3f,%.3f,%.3f\n" % (v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc, hit / (hit + miss) if hit + miss else 0))
print(open("gpurun_out/${tag}_pmc_counters_bf16.csv").read()[:3000])
PY
