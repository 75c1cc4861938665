"""Scan the gfx950 ISA of every csrc/*.hip for packed-f32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose
DESTINATION register pair is also a SOURCE pair that is read with a cross-half selection (op_sel / op_sel_hi): the low result
reading the pair's high dword, or the high result reading its low dword.  Round 5 found that such an instruction can give wrong
results in the last lanes of a wave when the SIMD is shared with another kernel's waves; the engine's
kernels must not contain any.  Exit code 1 when one is found.

    python tools/scan_pk_overlap.py            # compiles every csrc/*.hip to assembly (hipcc -S) and scans it
    python tools/scan_pk_overlap.py --lines    # ... and groups the instances by the source line they come from
    python tools/scan_pk_overlap.py --packed   # ... with the packed instructions turned back ON (what round 5 shipped: 117 sites)

Since round 6 the product library is compiled WITHOUT packed-f32 instructions (tacotron2_amd/build.py: NO_PACKED_F32 is part of
CFLAGS) and `build()` itself disassembles the LINKED library and fails on a single one (`build.scan_packed_f32`,
`python -m tacotron2_amd.build --scan`); this source-level scanner compiles with the same flags, says where a site comes from, and
ends with the library scan.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tacotron2_amd import build as _build  # noqa: E402
PAT = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+(.*)$")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return int(m.group(1)), int(m.group(1))
    return None


def scan(path, with_lines=False):
    """(packed-f32 instruction count, [(function, instruction[, "file:line"])]); `with_lines` reads the .loc directives of an
    assembly made with -gline-tables-only."""
    hits, n, fn = [], 0, "?"
    files, loc = {}, ""
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn = m.group(1)
        if with_lines:
            m = re.match(r'^\s*\.file\s+(\d+)\s+(?:"([^"]*)"\s+)?"([^"]*)"', line)
            if m:
                files[m.group(1)] = m.group(3)
            m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", line)
            if m:
                loc = "%s:%s" % (os.path.basename(files.get(m.group(1), "?")), m.group(2))
        m = PAT.match(line)
        if not m:
            continue
        n += 1
        body = m.group(2).split(";")[0]
        mods = dict((k, [int(x) for x in v.split(",")]) for k, v in re.findall(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[([\d,]+)\]", body))
        ops = [t.strip() for t in re.sub(r"\s+(op_sel|op_sel_hi|neg_lo|neg_hi):\[[\d,]+\]", "", body).split(",")]
        dst, srcs = regs(ops[0]), ops[1:]
        sel = mods.get("op_sel", [0, 0, 0]) + [0, 0, 0]
        selhi = mods.get("op_sel_hi", [1, 1, 1]) + [1, 1, 1]
        for k, stok in enumerate(srcs):
            r = regs(stok)
            if r is None or dst is None or r[1] < dst[0] or r[0] > dst[1]:
                continue
            if sel[k] == 1 or selhi[k] == 0:            # low result reads the high dword, or high result reads the low dword
                hits.append((fn, line.strip(), loc) if with_lines else (fn, line.strip()))
    return n, hits


def main():
    out = tempfile.mkdtemp(prefix="pkscan_")
    total, bad = 0, []
    lines = "--lines" in sys.argv            # also say which source line every instance comes from (-gline-tables-only)
    packed = "--packed" in sys.argv
    extra = [a for a in sys.argv[1:] if a not in ("--lines", "--packed")] + (["-gline-tables-only"] if lines else [])
    cflags = [f for f in _build.CFLAGS if f != "-fPIC"] + (_build.PACKED_F32 if packed else [])
    for src in sorted(glob.glob(os.path.join(ROOT, "tacotron2_amd", "csrc", "*.hip"))):
        asm = os.path.join(out, os.path.basename(src)[:-4] + ".s")
        subprocess.run([_build.HIPCC] + cflags + ["-S", "--cuda-device-only", "-o", asm, src] + extra,
                       check=True, stderr=subprocess.DEVNULL, cwd=out)
        n, hits = scan(asm, lines)
        total += n
        bad += [(os.path.basename(src),) + h for h in hits]
        print("%-22s %5d packed-f32 instructions, %3d with a swizzled source on the destination pair" % (os.path.basename(src), n, len(hits)))
    if lines:
        by_site = {}
        for f, fn, line, loc in bad:
            by_site.setdefault(loc, []).append(fn)
        for loc, fns in sorted(by_site.items(), key=lambda kv: -len(kv[1])):
            print("  %-28s %3d instances in %d kernels" % (loc, len(fns), len(set(fns))))
    else:
        for f, fn, line in bad[:60]:
            print("  %s  %s\n      %s" % (f, subprocess.run(["c++filt", fn], capture_output=True, text=True).stdout.strip()[:110], line))
    print("total: %d packed-f32 instructions, %d hazardous" % (total, len(bad)))
    rc = 1 if bad else 0
    if os.path.exists(_build.OUT) and not packed:
        n, hits = _build.scan_packed_f32(_build.OUT)
        print("linked library %s (sha1 %s): %d packed-f32 instructions, %d hazardous"
              % (os.path.relpath(_build.OUT, ROOT), (_build.built_sha1() or "?")[:12], n, len(hits)))
        rc |= int(n > 0)
    return rc


if __name__ == "__main__":
    sys.exit(main())
