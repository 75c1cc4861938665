"""hipGraph replay vs eager enqueue of the decode launch chain (B <= 8 path, loops.hip), BASELINE configs[3] shape.
    python tools/microbench_decode_graph.py
The engine's own descriptor of a 64-step chunk is intercepted (nothing is rebuilt by hand) and handed to the tools-only
C function t2amd_debug_graph_decode_, which times the identical kernel sequence enqueued eagerly and replayed from one
captured graph.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import engine, native as nv                # noqa: E402
from tacotron2_amd.hparams import create_hparams               # noqa: E402
from tacotron2_amd.model import Tacotron2                      # noqa: E402

lib = nv.load()
# the one-launch attention step carries its launch token as a kernel argument: a replayed graph would present the same
# token again and read the previous replay's granules, so the captured chain uses the two-launch form
nv.set_attn_fwd_fused(0)
fn = lib.t2amd_debug_graph_decode_
fn.argtypes = [C.POINTER(nv.DecInfer), C.c_int, C.POINTER(C.c_float), C.c_void_p]
fn.restype = C.c_int
dev = torch.device("cuda")
side = torch.cuda.Stream()
out = {}
real = nv.decoder_infer_steps
for name, prec, B in (("B1_bf16", "bf16", 1), ("B1_fp32", "fp32", 1), ("B4_bf16", "bf16", 4), ("B8_fp32", "fp32", 8)):
    hp = create_hparams()
    hp.max_decoder_steps = 64
    hp.gate_threshold = 2.0
    torch.manual_seed(1234)
    m = Tacotron2(hp).to(dev).eval()
    m.precision = prec
    engine.PERSISTENT_DECODE = False
    text = torch.randint(1, 148, (B, 100), device=dev)
    res = {}

    def probe(d, res=res):
        real(d)                                               # the engine's own call (state then holds step 64)
        torch.cuda.synchronize()
        d.t0, d.n_steps = 0, 62                               # stays clear of the max_decoder_steps stop at step 63
        ms = (C.c_float * 2)()
        with torch.cuda.stream(side):
            rc = fn(C.byref(d), 20, ms, C.c_void_p(side.cuda_stream))
        torch.cuda.synchronize()
        res.update(rc=rc, eager_us_per_step=ms[0] * 1e3 / 62, graph_us_per_step=ms[1] * 1e3 / 62)

    nv.decoder_infer_steps = probe
    try:
        with torch.no_grad():
            m.inference(text, torch.full((B,), 100, device=dev)) if B > 1 else m.inference(text)
    finally:
        nv.decoder_infer_steps = real
    out[name] = res
engine.PERSISTENT_DECODE = True
print(json.dumps(out))
