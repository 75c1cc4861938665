"""Per-kernel register / spill / LDS summary of one csrc/*.hip (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

    python tools/kernel_resources.py attention [name-filter] [extra hipcc flags...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "tacotron2_amd", "csrc", sys.argv[1] + ".hip")
filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
sys.path.insert(0, ROOT)
from tacotron2_amd import build as _build  # noqa: E402  (the product's own code-generation flags: packed-f32-free since round 6)
cmd = [_build.HIPCC] + _build.CFLAGS + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[\w/]+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
for k, v in rows.items():
    if filt in k:
        print("%-110s VGPR %3d AGPR %3d SGPR %3d spillS %3d spillV %3d scratch %4d LDS %6d occ %s" % (
            k[:110], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("TotalSGPRs", -1), v.get("SGPRs Spill", -1), v.get("VGPRs Spill", -1),
            v.get("ScratchSize", -1), v.get("LDS Size", -1), v.get("Occupancy", "?")))
