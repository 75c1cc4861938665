"""tools/scan_pk_overlap.py: the ISA pattern behind the fault of DESIGN.md section 5.3 -- a packed-f32 VALU instruction whose
destination pair is also a source pair read with a cross-half op_sel / op_sel_hi.  The parser is checked on the two instructions that
were proven wrong (and on their harmless neighbours); when hipcc is here, the kernel they were found in is compiled and must be clean."""
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scan_pk_overlap as scan                                   # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"


def _scan_text(text):
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as fh:
        fh.write(text)
        path = fh.name
    try:
        return scan.scan(path)
    finally:
        os.unlink(path)


def test_the_two_proven_instructions_are_recognised_and_their_neighbours_are_not():
    n, hits = _scan_text("""_Z1kv:
	v_pk_fma_f32 v[34:35], v[10:11], v[34:35], v[66:67] op_sel:[0,1,0]
	v_pk_fma_f32 v[36:37], v[12:13], v[36:37], v[38:39] op_sel_hi:[1,0,1]
	v_pk_fma_f32 v[36:37], v[50:51], v[76:77], v[36:37] op_sel_hi:[1,0,1]
	v_pk_fma_f32 v[34:35], v[6:7], v[76:77], v[34:35] op_sel:[0,1,0]
	v_pk_fma_f32 v[38:39], v[58:59], v[34:35], 0 op_sel_hi:[1,0,0]
	v_pk_mul_f32 v[2:3], v[2:3], v[4:5]
	v_pk_add_f32 v[8:9], v[6:7], v[8:9] op_sel_hi:[1,0]
	v_fma_f32 v1, v2, v3, v1
""")
    assert n == 7
    lines = [h[1] for h in hits]
    assert len(lines) == 3, lines                                 # the two proven ones + the packed add that broadcasts its own low half
    assert "v[10:11], v[34:35], v[66:67] op_sel:[0,1,0]" in lines[0]
    assert "v[12:13], v[36:37], v[38:39] op_sel_hi:[1,0,1]" in lines[1]
    assert lines[2].startswith("v_pk_add_f32")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles without a GPU)")
def test_the_kernel_the_fault_was_found_in_is_clean():
    out = tempfile.mkdtemp(prefix="pkscan_test_")
    try:
        asm = os.path.join(out, "attention.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm,
                        os.path.join(ROOT, "tacotron2_amd", "csrc", "attention.hip")], check=True, stderr=subprocess.DEVNULL, cwd=out)
        n, hits = scan.scan(asm)
        assert n > 1000                                           # the translation unit does use packed f32 (the scan is looking at real code)
        culprit = [h for h in hits if "attn_bwd_dw_kernelILb0E" in h[0]]
        assert not culprit, culprit
        # the f32 accumulation of K_b1 is unpacked by construction: its four FMAs per row and column group are plain v_fma_f32
        body = open(asm).read()
        i = body.find("_Z18attn_bwd_dw_kernelILb0EEv13AttnBwdParams:")
        k = body[i:body.find("s_endpgm", i)]
        assert k.count("v_fma_f32") >= 64 and k.count("v_pk_fma_f32") == 0, (k.count("v_fma_f32"), k.count("v_pk_fma_f32"))
    finally:
        shutil.rmtree(out, ignore_errors=True)
