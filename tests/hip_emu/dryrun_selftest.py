"""Dry run of tools/probe/gpu_selftest.cpp on the CPU: the self-test's own host reference code against the emulated
kernels (audio.hip / optim.hip compiled for the host) — so that the one GPU call it is meant for is not spent finding
a mistake in the checker.  The GEMM is a plain loop here and the barrier section is a stub."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "tacotron2_amd", "csrc")
exe = os.path.join(HERE, "selftest_dryrun")
cmd = ["g++", "-std=c++17", "-O1", "-w", "-ffp-contract=off", "-pthread", "-I", HERE, "-x", "c++",
       os.path.join(ROOT, "tools", "probe", "gpu_selftest.cpp"), os.path.join(CSRC, "audio.hip"),
       os.path.join(CSRC, "optim.hip"), os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "selftest_stubs.cpp"),
       "-o", exe]
subprocess.check_call(cmd)
sys.exit(subprocess.call([exe]))
