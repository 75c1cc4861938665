"""A/B of the single-utterance encoder bi-LSTM: one persistent launch vs the Ti-launch chain (python tools/ab_encoder_persistent.py).
Times Tacotron2.inference with max_decoder_steps = 2, so that the call is encoder + two decode steps + postnet."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import engine, native
from tacotron2_amd.hparams import create_hparams
from tacotron2_amd.model import Tacotron2

native.load()
dev = torch.device("cuda")
out = {}
for prec in ("fp32", "bf16"):
    for Ti in (100, 187):
        hp = create_hparams("max_decoder_steps=2")
        hp.gate_threshold = 2.0
        torch.manual_seed(1234)
        m = Tacotron2(hp).to(dev).eval()
        m.precision = prec
        text = torch.randint(1, 148, (1, Ti), device=dev)
        res = {}
        for rnd in range(3):
            for on in (True, False):
                engine.PERSISTENT_ENCODER = on
                with torch.no_grad():
                    for _ in range(3):
                        m.inference(text)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        m.inference(text)
                    torch.cuda.synchronize()
                res.setdefault("persistent" if on else "chain", []).append((time.perf_counter() - t0) / 20 * 1e3)
        out["%s_Ti%d" % (prec, Ti)] = {k: min(v) for k, v in res.items()}
        print(prec, Ti, out["%s_Ti%d" % (prec, Ti)], m.last_encoder_path, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/ab_encoder_persistent.json", "w"), indent=1)
