"""Alternating A/B of one module-level switch of tacotron2_amd.engine over whole training steps, in one process, same model and
batches (bf16 mode, B = 64, the bench's synthetic batches):

    timeout 600 python tools/ab_engine_flag.py ENCODER_BATCH_PERSISTENT_TRAIN [--steps 6] [--blocks 4] [--tag name]

Prints / writes gpurun_out/ab_<flag>.json: ms per training step (fwd + loss + bwd + clip + Adam) with the flag off and on, per
block, and whether the loss of the same step is bit-identical either way."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import engine, native
from tacotron2_amd.hparams import create_hparams
from tacotron2_amd.loss_function import Tacotron2Loss
from tacotron2_amd.model import Tacotron2
from tacotron2_amd.optim import FusedAdam
from tacotron2_amd.synth import synth_batch

ap = argparse.ArgumentParser()
ap.add_argument("flag")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--tag", default=None)
ap.add_argument("--values", default=None, help="off,on values of a non-boolean switch, e.g. --values 0,3")
a = ap.parse_args()
VALS = (False, True)
if a.values:
    VALS = tuple(int(v) for v in a.values.split(","))
else:
    assert isinstance(getattr(engine, a.flag), bool), "engine.%s is not a bool switch (use --values)" % a.flag
native.load()
dev = torch.device("cuda", 0)
hp = create_hparams()
torch.manual_seed(1234)
m = Tacotron2(hp).to(dev).train()
m.precision = "bf16"
crit = Tacotron2Loss()
opt = FusedAdam(m.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
batches = [tuple(t.to(dev) for t in synth_batch(64, 1234 + i)) for i in range(a.steps)]


def step(i):
    m.zero_grad()
    x, y = m.parse_batch(batches[i])
    loss = crit(m(x), y)
    loss.backward()
    opt.step(clip_norm=1.0)
    return loss


def same_loss():
    state = {k: v.clone() for k, v in m.state_dict().items()}
    vals = []
    for on in VALS:
        setattr(engine, a.flag, on)
        m.load_state_dict(state)
        torch.manual_seed(5)
        m.zero_grad()
        x, y = m.parse_batch(batches[0])
        vals.append(float(crit(m(x), y)))
    m.load_state_dict(state)
    return vals


def block(on):
    setattr(engine, a.flag, on)
    step(0); step(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / a.steps


losses = same_loss()
res = {"flag": a.flag, "loss_off_on": losses, "loss_bit_identical": losses[0] == losses[1], "off": [], "on": []}
for _ in range(a.blocks):
    res["off"].append(block(VALS[0]))
    res["on"].append(block(VALS[1]))
res["mean_off"], res["mean_on"] = sum(res["off"]) / a.blocks, sum(res["on"]) / a.blocks
res["give_ups"] = native.attn_handoff_timeouts(reset=False) + native.encoder_handoff_timeouts(reset=False)
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/ab_%s.json" % (a.tag or a.flag), "w") as fh:
    json.dump(res, fh, indent=1)
