"""Inference timing (BASELINE configs 4 and 5): decode steps/s.  python tools/bench_infer.py
Config 4: B=1, Ti=100, forced 1000 steps (gate threshold above 1: the stop never fires).
Config 5: 256 random-length texts, max_decoder_steps=2000 would run to the cap with random weights, so the
          batched run is also forced: 400 steps, all utterances active (worst case for the ragged path)."""
import json
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native
from tacotron2_amd.hparams import create_hparams
from tacotron2_amd.model import Tacotron2
from tacotron2_amd.synth import synth_lengths

dev = torch.device("cuda")
native.load()
PREC = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "fp32"
out = {"precision": PREC}
ONLY = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None      # e.g. --only config5_B256
CASES = (("config4_B1", 1, 1000), ("B16", 16, 400), ("B64", 64, 400), ("config5_B256", 256, 400))
if "--small" in sys.argv:        # B = 2 .. 8: the launch chain against consecutive single-utterance persistent decodes
    CASES = tuple(("B%d" % b, b, 400) for b in (2, 3, 4, 5, 6, 8))
for name, B, steps in CASES:
    if ONLY and name not in ONLY:
        continue
    hp = create_hparams()
    hp.max_decoder_steps = steps
    hp.gate_threshold = 2.0
    torch.manual_seed(1234)
    m = Tacotron2(hp).to(dev).eval()
    m.precision = PREC
    if B == 1:
        text = torch.randint(1, 148, (1, 100), device=dev)
        lens = None
    else:
        ti, _ = synth_lengths(B, 1234)
        Ti = int(ti.max())
        text = torch.zeros(B, Ti, dtype=torch.long, device=dev)
        for b in range(B):
            text[b, :ti[b]] = torch.randint(1, 148, (int(ti[b]),), device=dev)
        lens = torch.from_numpy(ti.copy()).to(dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            o = m.inference(text, lens) if lens is not None else m.inference(text)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    T = o[0].shape[2]
    out[name] = {"B": B, "steps": T, "seconds": dt, "decode_steps_per_s": T / dt, "utterance_steps_per_s": B * T / dt,
                 "decode_path": m.last_decode_path}
    print(name, json.dumps(out[name]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/bench_infer_%s%s.json" % (PREC, "_" + "_".join(ONLY) if ONLY else ""), "w"), indent=1)
