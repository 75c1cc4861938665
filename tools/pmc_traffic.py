"""Post-process two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 0` into
profiles/pmc_traffic.json (what bench.py reports as roofline.traffic) and a per-kernel CSV.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out_csv> [precision]

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts a wide (16 B/lane) streaming read at half its bytes
(MI355X_MICROARCH.md, HBM): reads are doubled before they are compared with a byte count.  The JSON is stamped with the
SHA-1 of the dominant kernel's sources, so that bench.py reports null instead of a stale number after a kernel change."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"]
                a = acc.setdefault(k, [0.0, set()])
                a[0] += float(row["Counter_Value"])
                a[1].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
    return {k: (v[0], max(len(v[1]), 1)) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, out_csv = sys.argv[1:4]
    prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = []
    for k in sorted(fe, key=lambda k: -fe[k][0]):
        f_kb = fe[k][0] / fe[k][1]
        w_kb = wr.get(k, (0.0, 1))[0] / wr.get(k, (0.0, 1))[1]
        rows.append((k, fe[k][1], f_kb, w_kb, 2 * f_kb / 1e3, (2 * f_kb + w_kb) / 1e3))
    with open(out_csv, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python bench.py --steps 1 "
                 "--warmup 0 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab (precision %s)\n" % prec)
        fh.write("# units: KB per launch; hbm_read_MB applies the gfx950 x2 correction for wide streaming reads\n")
        fh.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_read_MB_corrected,hbm_total_MB\n")
        for r in rows[:40]:
            fh.write('"%s",%d,%.1f,%.1f,%.2f,%.2f\n' % r)
    import bench
    want = "skinny_wide_kernel<true, 3>" if prec == "bf16" else "skinny_gemm_kernel<true, 3"
    rec_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    rec = json.load(open(rec_path)) if os.path.exists(rec_path) else {}
    for r in rows:
        if want in r[0]:
            rec["fused_" + prec] = r[5] * 1e6
            rec["fused_%s_fetch_kb" % prec] = r[2]
            rec["fused_%s_write_kb" % prec] = r[3]
            break
    rec["source_sha1"] = bench.kernel_source_sha1()
    rec["source"] = "%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes" % os.path.relpath(out_csv, ROOT)
    rec["correction"] = "FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads (MI355X_MICROARCH.md); WRITE_SIZE as reported"
    json.dump(rec, open(rec_path, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in rec if k.startswith("fused_") or k == "source_sha1"}))


if __name__ == "__main__":
    main()
