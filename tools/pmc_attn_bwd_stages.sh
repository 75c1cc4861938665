#!/bin/bash
# Where the attention backward's LDS bank conflicts are: the instrumented build (stage returns) truncated after
# stage n = 1..5 and whole (0), SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS / SQ_WAVE_CYCLES per launch of attn_bwd_main_kernel.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_attn_bwd_stages.sh r06_l'
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
python -m tacotron2_amd.build --stamps > /dev/null 2>&1
lib=$GRAFT_REPO_ROOT/tacotron2_amd/lib/libtacotron2_amd_stamps.so
for st in 1 2 3 4 5 0; do
  ( cd /tmp && rm -rf /tmp/pmc_st$st && T2AMD_LIB=$lib T2AMD_ATTN_STAGE=$st T2AMD_MB_BF16=1 timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_st$st -o s -- python $GRAFT_REPO_ROOT/tools/microbench_attn.py > /tmp/mb_st$st.txt 2>&1 )
done
cd $GRAFT_REPO_ROOT && python - <<PY
import csv, glob, collections
rows = []
for st in (1, 2, 3, 4, 5, 0):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob("/tmp/pmc_st%d/**/*counter_collection.csv" % st, recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            k = r["Kernel_Name"]
            if "attn_bwd" not in k and "attn_fwd" not in k and "attn_energy" not in k and "attn_context" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id"))
    for k in acc:
        m = max(len(n[k]), 1)
        rows.append((st, k[:70], m) + tuple(acc[k].get(c, 0) / m for c in ("SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES")))
    try: rows.append((st, open("/tmp/mb_st%d.txt" % st).read().strip().splitlines()[0][:120],))
    except Exception: pass
with open("gpurun_out/${tag}_attn_bwd_stage_conflicts.txt", "w") as fh:
    fh.write("# stage (0 = whole kernel), kernel, launches, per launch: LDS bank-conflict cycles, LDS instructions, LDS array cycles, wave cycles (quad-cycles)\n")
    for r in rows:
        fh.write("  ".join(("%.0f" % x) if isinstance(x, float) else str(x) for x in r) + "\n")
print(open("gpurun_out/${tag}_attn_bwd_stage_conflicts.txt").read())
PY
