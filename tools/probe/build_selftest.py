"""Builds tools/probe/gpu_selftest (torch-free GPU self-test of the round's last kernels, see gpu_selftest.cpp).
No GPU needed to build; the binary links against tacotron2_amd/lib/libtacotron2_amd.so through an $ORIGIN rpath."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tacotron2_amd import build  # noqa: E402

build.build(verbose=False)
out = os.path.join(HERE, "gpu_selftest")
cmd = [build.HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off",
       os.path.join(HERE, "gpu_selftest.cpp"), "-o", out,
       "-L", os.path.join(ROOT, "tacotron2_amd", "lib"), "-ltacotron2_amd",
       "-Wl,-rpath,$ORIGIN/../../tacotron2_amd/lib"]
print(" ".join(cmd))
subprocess.check_call(cmd)
print(out)
