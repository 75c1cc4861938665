"""Build ``lib/libtacotron2_amd.so`` for gfx950 with hipcc (cross-compiles without a GPU).

    python -m tacotron2_amd.build [--force]

The library is built in-tree so that it travels with the repository snapshot to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
DEPS = SRC + glob.glob(os.path.join(HERE, "csrc", "*.h")) + \
    [os.path.join(HERE, "..", "include", "tacotron2_amd.h")]
OUT = os.path.join(HERE, "lib", "libtacotron2_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"]


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(f) <= t for f in DEPS if os.path.exists(f))


STAMPS_OUT = os.path.join(HERE, "lib", "libtacotron2_amd_stamps.so")


def build(force=False, verbose=True, stamps=False):
    """stamps=True builds the instrumented variant (in-kernel phase stamps, tools only) next to the product library;
    select it with T2AMD_LIB=<path> T2AMD_ATTN_TS=1."""
    if stamps:
        os.makedirs(os.path.dirname(STAMPS_OUT), exist_ok=True)
        cmd = [HIPCC] + FLAGS + ["-DT2AMD_PHASE_STAMPS", "-o", STAMPS_OUT] + SRC
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return STAMPS_OUT
    if not force and up_to_date():
        return OUT
    if not force and os.path.exists(OUT) and not os.path.exists(HIPCC):
        # a box without the toolchain: the library shipped with the snapshot is the only one there can be
        print("tacotron2_amd.build: %s not found, using the shipped %s" % (HIPCC, OUT), file=sys.stderr)
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [HIPCC] + FLAGS + ["-o", OUT] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build(stamps=True))
    else:
        build(force="--force" in sys.argv)
        print(OUT)
