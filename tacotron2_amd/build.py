"""Build ``lib/libtacotron2_amd.so`` for gfx950 with hipcc (cross-compiles without a GPU).

    python -m tacotron2_amd.build [--force]

The library is built in-tree so that it travels with the repository snapshot to the GPU box.  Every ``csrc/*.hip`` is
compiled to its own object (in parallel, rebuilt only when it or a header is newer) and the objects are linked into
one shared library.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
HDRS = glob.glob(os.path.join(HERE, "csrc", "*.h")) + [os.path.join(HERE, "..", "include", "tacotron2_amd.h")]
DEPS = SRC + HDRS
OUT = os.path.join(HERE, "lib", "libtacotron2_amd.so")
OBJ_DIR = os.path.join(HERE, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# Code generation WITHOUT packed-f32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), for every translation unit
# of the product library and of every variant (round 6).  Round 5 proved that one whose destination pair is
# also a source pair read with a cross-half op_sel gives wrong lanes when another kernel's MFMA waves share the SIMD (DESIGN.md
# section 5.3); hipcc's SLP vectoriser makes that pattern wherever two adjacent f32 FMAs share an operand (expf/tanhf inlined from
# __clang_hip_math.h alone made 90 of the 117 instances of round 5).  The A/B of profiles/r05_j_* shows no cost, the guide lists
# packed f32 beside MFMAs as an anti-lever anyway.  `scan_packed_f32()` below disassembles the LINKED library and build() fails
# when a single such instruction is left.
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + NO_PACKED_F32
FLAGS = CFLAGS + ["-shared"]
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


STAMP = OUT + ".sha1"


def source_sha1():
    """SHA-1 over EVERY kernel source (csrc/*) and the C-ABI header, names included.  It is compiled into the library
    (``t2amd_source_sha1()``, csrc/api.hip), written beside it (``libtacotron2_amd.so.sha1``), recorded with the PMC passes
    (profiles/pmc_traffic.json) and printed in the bench line: binary, counters and sources are provably the same tree."""
    import hashlib
    h = hashlib.sha1()
    h.update(" ".join(CFLAGS).encode())        # a library built with other code-generation flags is another library (round 6)
    # exactly what is compiled: the .hip sources, their headers and torch_ops.cpp -- not whatever else lies in
    # csrc/ (an editor backup or a sub-directory used to make import fail with a misleading "built from other sources")
    files = sorted(f for f in glob.glob(os.path.join(HERE, "csrc", "*"))
                   if os.path.isfile(f) and f.endswith((".hip", ".h", ".cpp")))
    files.append(os.path.join(HERE, "..", "include", "tacotron2_amd.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def built_sha1():
    """The source hash the shipped library was built from (None: no library / no stamp)."""
    try:
        with open(STAMP) as fh:
            return fh.read().strip() or None
    except OSError:
        return None


def up_to_date():
    """The library exists and was built from exactly these sources -- by content, not by mtime (a
    checkout or a snapshot copy resets mtimes, a hash cannot be fooled either way)."""
    return os.path.exists(OUT) and built_sha1() == source_sha1()


STAMPS_OUT = os.path.join(HERE, "lib", "libtacotron2_amd_stamps.so")
# TORCH_LIBRARY registration of the loop-level entry points over the same C ABI (csrc/torch_ops.cpp): host-only C++,
# compiled with g++ against torch's headers, linked to the library above (found beside it through $ORIGIN)
TORCH_OPS_SRC = os.path.join(HERE, "csrc", "torch_ops.cpp")
TORCH_OPS_OUT = os.path.join(HERE, "lib", "libtacotron2_amd_torch.so")
CXX = os.environ.get("CXX", "g++")


def build_torch_ops(force=False, verbose=True):
    deps = [TORCH_OPS_SRC, os.path.join(HERE, "..", "include", "tacotron2_amd.h"), OUT]
    if not force and os.path.exists(TORCH_OPS_OUT) and all(os.path.getmtime(f) <= os.path.getmtime(TORCH_OPS_OUT)
                                                            for f in deps[:2]):
        return TORCH_OPS_OUT
    import shutil
    if shutil.which(CXX) is None:
        if os.path.exists(TORCH_OPS_OUT):
            return TORCH_OPS_OUT
        raise RuntimeError("tacotron2_amd.build: no C++ compiler (%s) for csrc/torch_ops.cpp" % CXX)
    import torch
    ti = os.path.dirname(torch.__file__)
    abi = int(bool(torch._C._GLIBCXX_USE_CXX11_ABI))
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi, "-I" + os.path.join(ti, "include"),
           "-I" + os.path.join(ti, "include", "torch", "csrc", "api", "include"), "-I" + os.path.join(rocm, "include"),
           TORCH_OPS_SRC, "-o", TORCH_OPS_OUT, "-L" + os.path.join(ti, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
           "-L" + os.path.dirname(OUT), "-ltacotron2_amd", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(ti, "lib")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TORCH_OPS_OUT


def _compile_objects(obj_dir, extra, verbose):
    os.makedirs(obj_dir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in HDRS if os.path.exists(h))
    sha = source_sha1()
    sha_file = os.path.join(obj_dir, "api.sha1")           # the hash api.o was compiled with
    flags_file = os.path.join(obj_dir, "flags.txt")        # the code-generation flags EVERY object here was compiled with
    flags = " ".join(CFLAGS + extra)
    try:
        with open(flags_file) as fh:
            same_flags = fh.read().strip() == flags
    except OSError:
        same_flags = False
    if not same_flags:                                     # objects of another code generation are not "up to date" (round 6)
        for o in glob.glob(os.path.join(obj_dir, "*.o")):
            os.remove(o)
    try:
        with open(sha_file) as fh:
            api_sha = fh.read().strip()
    except OSError:
        api_sha = None
    jobs = []
    for src in SRC:
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        is_api = os.path.basename(src) == "api.hip"
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t) and not (is_api and api_sha != sha):
            continue
        # the hash of ALL sources is compiled into api.o (t2amd_source_sha1): it is rebuilt whenever any source changed
        jobs.append([HIPCC] + CFLAGS + extra + (['-DT2AMD_SOURCE_SHA1="%s"' % sha] if is_api else []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        # (the host pass of hipcc does not know the AMDGPU feature of NO_PACKED_F32 and says so once per file: not a diagnostic)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        err = "\n".join(l for l in r.stderr.splitlines() if "is not a recognized feature for this target" not in l)
        if err.strip():
            print(err, file=sys.stderr, flush=True)
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    with open(sha_file, "w") as fh:
        fh.write(sha)
    with open(flags_file, "w") as fh:
        fh.write(flags)
    return [os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o") for s in SRC]


_PK_PAT = None


def _pk_hazard(body):
    """True when a packed-f32 instruction's destination pair is also a source pair read with a cross-half selection (the low
    result reading the pair's high dword or the high result its low dword): the operand pattern of DESIGN.md section 5.3."""
    import re

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return int(m.group(1)), int(m.group(2))
        m = re.match(r"v(\d+)$", tok)
        return (int(m.group(1)), int(m.group(1))) if m else None

    mods = dict((k, [int(x) for x in v.split(",")]) for k, v in re.findall(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[([\d,]+)\]", body))
    ops = [t.strip() for t in re.sub(r"\s+(op_sel|op_sel_hi|neg_lo|neg_hi):\[[\d,]+\]", "", body).split(",")]
    dst = regs(ops[0])
    sel = mods.get("op_sel", [0, 0, 0]) + [0, 0, 0]
    selhi = mods.get("op_sel_hi", [1, 1, 1]) + [1, 1, 1]
    for k, stok in enumerate(ops[1:]):
        r = regs(stok)
        if r is None or dst is None or r[1] < dst[0] or r[0] > dst[1]:
            continue
        if sel[k] == 1 or selhi[k] == 0:
            return True
    return False


def scan_packed_f32(lib=None):
    """Disassemble every gfx950 code object of the LINKED library (what ships, not what a re-compile would give) and return
    ``(number of v_pk_{fma,mul,add}_f32 instructions, [(kernel, instruction)] of those with the section-5.3 operand pattern)``.
    The product build requires the first number to be 0 (the stricter condition: no packed f32 at all)."""
    import re
    import shutil
    import tempfile
    lib = lib or OUT
    tmp = tempfile.mkdtemp(prefix="t2amd_pkscan_")
    try:
        cp = os.path.join(tmp, "lib.so")
        shutil.copy(lib, cp)
        subprocess.run([OBJDUMP, "--offloading", cp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        objs = sorted(f for f in glob.glob(cp + ".*") if "amdgcn" in os.path.basename(f))
        if not objs:
            raise RuntimeError("tacotron2_amd.build: no gfx950 code object found in %s" % lib)
        pat = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+(.*)$")
        n, hits = 0, []
        for o in objs:
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
            fn = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
                if m:
                    fn = m.group(1)
                    continue
                m = pat.match(line)
                if not m:
                    continue
                n += 1
                body = re.split(r"//|;", m.group(2))[0].strip()
                if _pk_hazard(body):
                    hits.append((fn, m.group(1) + " " + body))
        return n, hits
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def enforce_no_packed_f32(lib, verbose=True):
    """The hard gate of build(): the linked library carries NO packed-f32 VALU instruction (so none of the hazardous form
    either).  A box without llvm-objdump cannot have compiled the library in the first place (it came with the snapshot)."""
    if not os.path.exists(OBJDUMP):
        print("tacotron2_amd.build: %s not found, ISA scan of %s skipped" % (OBJDUMP, lib), file=sys.stderr)
        return None
    n, hits = scan_packed_f32(lib)
    if verbose:
        print("tacotron2_amd.build: ISA scan of %s: %d packed-f32 instructions, %d hazardous" % (os.path.basename(lib), n, len(hits)),
              flush=True)
    if n or hits:
        try:
            os.remove(lib)                                   # never leave a library that failed the gate where load() finds it
        except OSError:
            pass
        raise RuntimeError("tacotron2_amd.build: %s contains %d packed-f32 instructions (%d with a swizzled source on the "
                           "destination pair, DESIGN.md 5.3), e.g. %s" % (lib, n, len(hits), hits[:2]))
    return n


def _link(objs, out, verbose):
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=True, stamps=False):
    """stamps=True builds the instrumented variant (in-kernel phase stamps, tools only) next to the product library;
    select it with T2AMD_LIB=<path> T2AMD_ATTN_TS=1."""
    if stamps:
        os.makedirs(os.path.dirname(STAMPS_OUT), exist_ok=True)
        _link(_compile_objects(OBJ_DIR + "_stamps", ["-DT2AMD_PHASE_STAMPS"], verbose), STAMPS_OUT, verbose)
        enforce_no_packed_f32(STAMPS_OUT, verbose)
        return STAMPS_OUT
    if not force and up_to_date():
        _torch_ops_best_effort(False, verbose)
        return OUT
    if not force and os.path.exists(OUT) and not os.path.exists(HIPCC):
        # a box without the toolchain: the library shipped with the snapshot is the only one there can be
        print("tacotron2_amd.build: %s not found, using the shipped %s" % (HIPCC, OUT), file=sys.stderr)
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if force:
        for o in glob.glob(os.path.join(OBJ_DIR, "*.o")):
            os.remove(o)
    sha = source_sha1()
    _link(_compile_objects(OBJ_DIR, [], verbose), OUT, verbose)
    enforce_no_packed_f32(OUT, verbose)
    with open(STAMP, "w") as fh:
        fh.write(sha + "\n")
    _torch_ops_best_effort(force, verbose)
    return OUT


def _torch_ops_best_effort(force, verbose):
    """The dispatcher registration is an ADDITIONAL route to the same C ABI: a box that cannot compile it (no g++, other
    torch headers) still has the whole engine through ctypes -- say so, do not fail the build."""
    try:
        build_torch_ops(force=force, verbose=verbose)
    except Exception as e:                                   # noqa: BLE001
        print("tacotron2_amd.build: TORCH_LIBRARY registration not built (%s: %s); the engine uses the ctypes route"
              % (type(e).__name__, e), file=sys.stderr)
        if os.path.exists(TORCH_OPS_OUT) and os.path.getmtime(TORCH_OPS_OUT) < os.path.getmtime(OUT):
            os.remove(TORCH_OPS_OUT)                         # never pair a stale registration with a newer library


def build_variant(tag, defines, verbose=True, enforce_scan=True):
    """An A/B build next to the product library: every ``-D<define>`` applied to all translation units, output
    ``lib/libtacotron2_amd_<tag>.so`` (select it with T2AMD_LIB=<path>; tools only)."""
    out = os.path.join(HERE, "lib", "libtacotron2_amd_%s.so" % tag)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # (an entry that starts with '-' is passed to hipcc as it is: a code-generation option instead of a define)
    _link(_compile_objects(OBJ_DIR + "_" + tag, [d if d.startswith("-") else "-D" + d for d in defines], verbose), out, verbose)
    if enforce_scan:
        enforce_no_packed_f32(out, verbose)
    elif os.path.exists(OBJDUMP):
        n, hits = scan_packed_f32(out)
        print("tacotron2_amd.build: ISA scan of %s: %d packed-f32 instructions, %d hazardous (not enforced for this variant)"
              % (os.path.basename(out), n, len(hits)), flush=True)
    return out


# The A/B counterpart of round 5's `--no-packed-f32`: since round 6 the product library is the packed-free one, `--packed-f32`
# builds lib/libtacotron2_amd_pk.so WITH the packed instructions (tools only: select it with T2AMD_LIB; build() never ships it,
# and the ISA scan is reported, not enforced, for it).
PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "+packed-fp32-ops"]


if __name__ == "__main__":
    if "--packed-f32" in sys.argv:
        print(build_variant("pk", PACKED_F32, enforce_scan=False))
    elif "--scan" in sys.argv:                          # python -m tacotron2_amd.build --scan [library]
        libs = [a for a in sys.argv[1:] if not a.startswith("--")] or [OUT]
        rc = 0
        for lib in libs:
            n, hits = scan_packed_f32(lib)
            print("%s: %d packed-f32 instructions, %d with a swizzled source on the destination pair" % (lib, n, len(hits)))
            for fn, ins in hits[:40]:
                print("   %s\n      %s" % (fn[:120], ins))
            rc |= int(n > 0)
        sys.exit(rc)
    elif "--variant" in sys.argv:                       # python -m tacotron2_amd.build --variant epifirst T2AMD_SW_EPI_FIRST
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    elif "--stamps" in sys.argv:
        print(build(stamps=True))
    else:
        build(force="--force" in sys.argv)
        print(OUT)
