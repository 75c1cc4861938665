"""Builds tests/hip_emu/libt2amd_emu.so: csrc/audio.hip, csrc/optim.hip and csrc/cell_bwd.h (through cell_bwd_emu.cpp)
compiled FOR THE HOST against the stand-in HIP header of this directory (test infrastructure only; see hip/hip_runtime.h)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "tacotron2_amd", "csrc")
OUT = os.path.join(HERE, "libt2amd_emu.so")
SOURCES = [os.path.join(CSRC, "audio.hip"), os.path.join(CSRC, "optim.hip"), os.path.join(HERE, "emu_runtime.cpp"),
           os.path.join(HERE, "cell_bwd_emu.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "cell_bwd.h"), os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(CSRC, "common.h"),
                  os.path.join(ROOT, "include", "tacotron2_amd.h")]


def build(verbose=False):
    if os.path.exists(OUT) and all(os.path.getmtime(f) <= os.path.getmtime(OUT) for f in DEPS):
        return OUT
    cmd = ["g++", "-std=c++17", "-O1", "-g0", "-w", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
           "-I", HERE, "-x", "c++"] + SOURCES + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(verbose=True))
