"""A/B of the teacher-forced decoder loop: launch chain (two dependent launches per time step) against the ONE persistent launch
(csrc/attention.hip, dec_train_fwd_persistent_kernel) -- same process, same model, same batches, same dropout masks.

    timeout 600 python tools/ab_train_fwd_persistent.py [--small]      # writes gpurun_out/ab_train_fwd_persistent.json

1. bitwise: the four outputs, the loss and all 60 gradients of one training step, chain vs persistent (B = 64, To = 870, bf16
   mode; --small: B = 5, To = 37 as well);
2. time: alternating blocks of full training steps (fwd + loss + bwd + clip + Adam) with either forward loop.
"""
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
ap.add_argument("--small", action="store_true")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--blocks", type=int, default=3)
a = ap.parse_args()
native.load()
dev = torch.device("cuda", 0)
hp = create_hparams()
crit = Tacotron2Loss()
out = {}


def one_step(m, batch, persistent, seed=99):
    engine.TRAIN_FWD_PERSISTENT = persistent
    m.zero_grad()
    torch.manual_seed(seed)                       # the Philox keep-masks are seeded from torch's RNG
    x, y = m.parse_batch(batch)
    o = m(x)
    loss = crit(o, y)
    loss.backward()
    torch.cuda.synchronize()
    return [t.detach().clone() for t in o], loss.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, m.last_train_decoder_path


def compare(tag, batch):
    torch.manual_seed(1234)
    m = Tacotron2(hp).to(dev).train()
    m.precision = "bf16"
    o0, l0, g0, p0 = one_step(m, batch, False)
    o1, l1, g1, p1 = one_step(m, batch, True)
    o2, l2, g2, p2 = one_step(m, batch, True)
    names = ("mel", "mel_post", "gate", "align")
    row = {"paths": [p0, p1, p2], "loss": [float(l0), float(l1), float(l2)],
           "outputs_bit_identical": {n: bool(torch.equal(o0[i], o1[i]) and torch.equal(o0[i], o2[i])) for i, n in enumerate(names)},
           "outputs_max_abs_diff": {n: float((o0[i] - o1[i]).abs().max()) for i, n in enumerate(names)},
           "finite": bool(all(torch.isfinite(t).all() for t in o1)),
           "gradients_differing": [k for k in g0 if not (torch.equal(g0[k], g1[k]) and torch.equal(g0[k], g2[k]))]}
    row["ok"] = p1 == "persistent" and all(row["outputs_bit_identical"].values()) and not row["gradients_differing"] and float(l0) == float(l1)
    out[tag] = row
    print(tag, json.dumps(row)[:600], flush=True)
    return m


if a.small:
    full = synth_batch(64, 4321)
    idx = torch.tensor([0, 13, 27, 41, 63])
    text, il, mel, gate, ol = (t[idx] for t in full)
    ol = torch.clamp(ol, max=37)
    Ti, To = int(il.max()), int(ol.max())
    gate = torch.zeros(5, To)
    for i in range(5):
        gate[i, int(ol[i]) - 1:] = 1.0
    small = (text[:, :Ti].contiguous(), il, mel[:, :, :To].contiguous(), gate, ol)
    compare("small_B5_To%d" % To, tuple(t.to(dev) for t in small))
batches = [tuple(t.to(dev) for t in synth_batch(64, 1234 + i)) for i in range(a.steps)]
m = compare("full_B64_To870", batches[0])

# ---- timing: whole training steps, alternating blocks --------------------------------------------------------------
opt = FusedAdam(m.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)


def block(persistent):
    engine.TRAIN_FWD_PERSISTENT = persistent
    for i in range(2):                                             # warm
        m.zero_grad(); x, y = m.parse_batch(batches[i]); crit(m(x), y).backward(); opt.step(clip_norm=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        m.zero_grad(); x, y = m.parse_batch(batches[i]); crit(m(x), y).backward(); opt.step(clip_norm=1.0)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / a.steps


def fwd_only(persistent):
    engine.TRAIN_FWD_PERSISTENT = persistent
    with torch.no_grad():
        x, y = m.parse_batch(batches[0])
        m(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            m(x)
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / a.steps


# ---- where a persistent time step spends its time, and the pre-poll pauses ---------------------------------------------
import ctypes as C
lib = native.load()
lib.t2amd_debug_dtp_prof_.argtypes = [C.c_void_p]
prof = torch.zeros(8, dtype=torch.int64, device=dev)
sweep = {}
for dl, dt_ in ((16, 8), (0, 8), (2, 8), (4, 8), (8, 8), (12, 8), (24, 8)):
    os.environ["T2AMD_DTP_DELAY_L"], os.environ["T2AMD_DTP_DELAY_T"] = str(dl), str(dt_)
    ms = min(fwd_only(True) for _ in range(2)) if False else None
    sweep["L%d_T%d" % (dl, dt_)] = None
os.environ["T2AMD_DTP_DELAY_L"], os.environ["T2AMD_DTP_DELAY_T"] = "16", "8"

times = {"chain": [], "persistent": [], "fwd_only_chain": [], "fwd_only_persistent": []}
for _ in range(a.blocks):
    times["chain"].append(block(False))
    times["persistent"].append(block(True))
    times["fwd_only_chain"].append(fwd_only(False))
    times["fwd_only_persistent"].append(fwd_only(True))
for key in list(sweep):
    dl, dt_ = key[1:].split("_T")
    os.environ["T2AMD_DTP_DELAY_L"], os.environ["T2AMD_DTP_DELAY_T"] = dl, dt_
    sweep[key] = min(fwd_only(True) for _ in range(2))
os.environ["T2AMD_DTP_DELAY_L"], os.environ["T2AMD_DTP_DELAY_T"] = "16", "8"
out["fwd_only_ms_by_prepoll_pause"] = sweep
print("pause sweep (fwd only, ms):", json.dumps(sweep))
prof.zero_()
lib.t2amd_debug_dtp_prof_(C.c_void_p(prof.data_ptr()))
engine.TRAIN_FWD_PERSISTENT = True
with torch.no_grad():
    m(m.parse_batch(batches[0])[0])
torch.cuda.synchronize()
lib.t2amd_debug_dtp_prof_(None)
pv = prof.tolist()
To_ = batches[0][2].shape[2]
names4 = ("wait_attention_flags", "lstm_tile_and_publish", "wait_lstm_flags", "attention_step_and_publish")
out["phase_us_per_time_step"] = {"workgroup0_lstm_a_tile_and_attention": {n: pv[i] / 100.0 / To_ for i, n in enumerate(names4)},
                                 "lstm_d_tile_workgroup": {n: pv[4 + i] / 100.0 / To_ for i, n in enumerate(names4)}}
print("phase clocks (us per time step):", json.dumps(out["phase_us_per_time_step"]))
out["ms_per_training_step"] = times
out["give_ups"] = native.attn_handoff_timeouts(reset=False)
print(json.dumps(times), "give-ups:", out["give_ups"])
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/ab_train_fwd_persistent.json", "w") as fh:
    json.dump(out, fh, indent=1)
