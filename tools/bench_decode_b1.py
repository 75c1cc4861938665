"""BASELINE configs[3] (Tacotron2.inference, batch 1, Ti = 100): decode steps/s of every variant of the decode loop.
    python tools/bench_decode_b1.py [--steps 1000]
Variants: persistent weight-stationary kernel (bf16 and fp32 weights), launch chain bf16, launch chain fp32.  Forced length (gate
threshold above 1) so that the timing does not depend on the random weights; whole Tacotron2.inference call inside the
timed region (encoder + loop + postnet); the loop alone is reported from the difference to a 1-step call."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import engine, native                     # noqa: E402
from tacotron2_amd.hparams import create_hparams             # noqa: E402
from tacotron2_amd.model import Tacotron2                    # noqa: E402

STEPS = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1000
dev = torch.device("cuda")
native.load()


def timed(prec, persistent, steps, phases=False):
    hp = create_hparams()
    hp.max_decoder_steps = steps
    hp.gate_threshold = 2.0
    torch.manual_seed(1234)
    m = Tacotron2(hp).to(dev).eval()
    m.precision = prec
    text = torch.randint(1, 148, (1, 100), device=dev)
    engine.PERSISTENT_DECODE = persistent
    m.persist_timing = phases
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            o = m.inference(text)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert o[0].shape[2] == steps
    global PHASES
    if m.persist_timing and getattr(m, 'last_persist_timing', None) is not None:
        tk = m.last_persist_timing.cpu().tolist()
        names = ["wait p2", "lstm_a tail + publish h_a", "wait h_a", "energies (team) / h_a parts (others)", "wait energies",
                 "softmax + ctx publish", "h_a parts (team, deferred)", "wait ctx", "lstm_d tail + publish h_d", "ctx part of next lstm_a",
                 "wait h_d", "projection rows + publish p1", "h_d part of next lstm_d", "wait p1", "prenet layer 2 + publish p2"]
        PHASES = {"unit": "us per step (100 MHz wall clock, thread 0)",
                  "first workgroup (attention team)": {n: tk[i] / steps / 100.0 for i, n in enumerate(names)},
                  "last workgroup": {n: tk[16 + i] / steps / 100.0 for i, n in enumerate(names)}}
    return best, m.last_decode_path


PHASES = None


out = {"steps": STEPS, "Ti": 100}
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    for name, prec, pers in (("persistent_bf16", "bf16", True), ("persistent_fp32", "fp32", True),
                             ("launch_chain_bf16", "bf16", False), ("launch_chain_fp32", "fp32", False)):
        full, path = timed(prec, pers, STEPS)
        one, _ = timed(prec, pers, 1)
        if pers:
            PHASES = None
            timed(prec, pers, STEPS, phases=True)             # separate run: the phase clock is not in the timed numbers
            if PHASES:
                out[name + "_phases"] = PHASES
        es = 2.0 if prec == "bf16" else 4.0
        step_bytes = es * (18189969 + 640 * 100.0)
        loop = max(full - one, 1e-9)
        out[name] = {"path": path, "seconds_whole_call": full, "seconds_1_step_call": one,
                     "decode_steps_per_s_whole_call": STEPS / full, "us_per_step_loop_only": 1e6 * loop / (STEPS - 1),
                     "hbm_roofline_frac_whole_call": step_bytes * STEPS / full / 8e12}
engine.PERSISTENT_DECODE = True
print(json.dumps(out))
