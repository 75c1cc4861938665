"""Post-process rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) of
`bench.py --steps 1 --warmup 0` into profiles/pmc_traffic.json (what bench.py reports as roofline.traffic for each of
the four kernels of a decoder time step) and a per-kernel CSV.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out_csv> [precision] [calib_fetch_dir] [calib_write_dir]

FETCH_SIZE / WRITE_SIZE are in KiB (1024 bytes: the calibration below reads 2.048 with 1000-byte units, 2.000 with 1024).  On gfx950 FETCH_SIZE counts a wide (16 B/lane) coalesced streaming read at HALF its
bytes (MI355X_MICROARCH.md, HBM) and is uncalibrated for other access shapes: `calib_fetch_dir` holds a FETCH_SIZE pass of
tools/probe/fetch_calib (1 GiB read once per access shape), from which the bytes-per-counted-byte factor of every shape
is taken; without it the guide's x2 is applied to the LDS-DMA weight streams and the attention kernels are reported with
BOTH bounds (x1, x2).  The JSON is stamped with the SHA-1 of ALL kernel sources (bench.kernel_source_sha1), so that
bench.py reports null instead of a stale number after any kernel change."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CALIB_BYTES = float(1 << 30)
# kernel-name substring per chain kernel, and the access shapes its HBM streams are made of (weights by bytes, SURVEY 8d:
# per encoder position 512 bf16 memory channels in 256-byte column groups + 128 f32 processed-memory dims in 128-byte slices)
CHAIN = {
    "bf16": {"decoder_forward_persistent": ("dec_train_fwd_persistent_kernel", {"calib_lds16": 39.2, "calib_seg256": 5.8, "calib_seg128": 2.0}),
             "lstm_pair": ("skinny_wide_kernel<true, 3,", {"calib_lds16": 1.0}),
             "dgrad_pair": ("skinny_wide_kernel<false, 3,", {"calib_lds16": 1.0}),
             "attention_forward": ("attn_fwd_fused_kernel", {"calib_seg256": 1024.0, "calib_seg128": 512.0}),
             "attention_backward": ("attn_bwd_main_kernel", {"calib_seg256": 1024.0, "calib_seg128": 512.0})},
    "fp32": {"lstm_pair": ("skinny_gemm_kernel<true, 3", {"calib_lds16": 1.0}),
             "dgrad_pair": ("skinny_gemm_kernel<false, 3", {"calib_lds16": 1.0}),
             "attention_forward": ("attn_fwd_fused_kernel", {"calib_b16": 2048.0, "calib_seg128": 512.0}),
             "attention_backward": ("attn_bwd_main_kernel", {"calib_b16": 2048.0, "calib_seg128": 512.0})},
}


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


def calibration(calib_dir, write_dir=None):
    """shape -> true bytes per byte FETCH_SIZE counted (1.0 = the counter is exact for that shape); "write" -> the same
    for WRITE_SIZE on a 16 B/lane coalesced store stream."""
    out = {}
    if write_dir:
        for k, (kb, n) in per_kernel(write_dir, "WRITE_SIZE").items():
            if "calib_w16" in k and kb > 0:
                out["write"] = CALIB_BYTES / (kb / n * 1024.0)
    if not calib_dir:
        return out
    for k, (kb, n) in per_kernel(calib_dir, "FETCH_SIZE").items():
        for shape in ("calib_b4", "calib_b8", "calib_b16", "calib_seg128", "calib_seg256", "calib_lds16"):
            if shape + "(" in k or k.startswith(shape) or (" " + shape) in k:
                out[shape] = CALIB_BYTES / (kb / n * 1024.0)
    return out


def main():
    fetch_dir, write_dir, out_csv = sys.argv[1:4]
    prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
    cal = calibration(sys.argv[5] if len(sys.argv) > 5 else None, sys.argv[6] if len(sys.argv) > 6 else None)
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = []
    for k in sorted(fe, key=lambda k: -fe[k][0]):
        f_kb = fe[k][0] / fe[k][1]
        w_kb = wr.get(k, (0.0, 1))[0] / wr.get(k, (0.0, 1))[1]
        rows.append((k, fe[k][1], f_kb, w_kb))
    with open(out_csv, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python bench.py --steps 1 "
                 "--warmup 0 --cpu-sample 0 --no-roofline --no-fp32-leg --no-inference --no-optimizer-ab (precision %s)\n" % prec)
        fh.write("# units: KB per launch, RAW counter values (no correction); FETCH_SIZE calibration of this box, true bytes per "
                 "counted byte: %s\n" % (json.dumps({k: round(v, 3) for k, v in sorted(cal.items())}) if cal else "not run"))
        fh.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch\n")
        for r in rows[:40]:
            fh.write('"%s",%d,%.1f,%.1f\n' % r)
    import bench
    rec_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    rec = json.load(open(rec_path)) if os.path.exists(rec_path) else {}
    kernels = {}
    wfac = cal.get("write", 1.0)                       # true bytes per counted WRITE_SIZE byte (calib_w16), 1.0 if not run
    for key, (sub, shapes) in CHAIN[prec].items():
        for k, n, f_kb, w_kb in rows:
            if sub not in k:
                continue
            e = {"kernel": k, "launches": n, "fetch_kb_raw": f_kb, "write_kb": w_kb}
            if all(s in cal for s in shapes):
                tot = sum(shapes.values())
                factor = sum(w * cal[s] for s, w in shapes.items()) / tot
                e["fetch_factor"] = factor
                e["fetch_factor_source"] = "tools/probe/fetch_calib on this box, shapes %s" % json.dumps(shapes)
            elif set(shapes) == {"calib_lds16"}:
                factor = 2.0
                e["fetch_factor"] = 2.0
                e["fetch_factor_source"] = "MI355X_MICROARCH.md: x2 for 16 B/lane streaming reads (not re-calibrated)"
            else:
                factor = None
                e["hbm_bytes_bounds"] = [(f_kb + w_kb) * 1024.0, (2 * f_kb + w_kb) * 1024.0]
            e["hbm_bytes"] = ((factor * f_kb + w_kb * wfac) * 1024.0) if factor else None
            kernels[key] = e
            break
    rec["kernels_" + prec] = kernels
    if cal:
        rec["fetch_calibration"] = cal
    rec.pop("fused_" + prec, None); rec.pop("fused_%s_fetch_kb" % prec, None); rec.pop("fused_%s_write_kb" % prec, None)
    rec["source_sha1"] = bench.kernel_source_sha1()
    rec["source"] = "%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes" % os.path.relpath(out_csv, ROOT)
    rec["correction"] = ("hbm_bytes = fetch_factor x FETCH_SIZE + WRITE_SIZE; fetch_factor per access shape from "
                         "tools/probe/fetch_calib (1 GiB read once per shape) when that pass was run")
    json.dump(rec, open(rec_path, "w"), indent=1)
    print(json.dumps({k: (v.get("hbm_bytes"), v.get("fetch_factor")) for k, v in kernels.items()}))
    print(json.dumps(cal))


if __name__ == "__main__":
    main()
