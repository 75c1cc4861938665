"""A/B of BPTT through the teacher-forced decoder loop: launch chain (two dependent launches per time step) against the ONE
persistent launch (csrc/attention.hip, dec_train_bwd_persistent_kernel) -- same process, same model, same batches, same
dropout masks; the forward loop is the persistent launch in both legs.

    timeout 600 python tools/ab_train_bwd_persistent.py [--small]      # writes gpurun_out/ab_train_bwd_persistent.json

1. bitwise: the loss and all 60 gradients of one training step, chain vs persistent (B = 64, To = 870, bf16 mode; --small: a
   5-utterance batch as well);
2. time: alternating blocks of full training steps (fwd + loss + bwd + clip + Adam) with either backward loop;
3. the pre-poll pauses of the two waits (T2AMD_DBP_DELAY_1 / _2) and workgroup 0's phase clocks.
"""
import argparse
import ctypes as C
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
ap.add_argument("--no-sweep", action="store_true")
a = ap.parse_args()
lib = native.load()
dev = torch.device("cuda", 0)
hp = create_hparams()
crit = Tacotron2Loss()
out = {}
engine.TRAIN_FWD_PERSISTENT = True


def one_step(m, batch, persistent, seed=99):
    engine.TRAIN_BWD_PERSISTENT = persistent
    m.zero_grad()
    torch.manual_seed(seed)                       # the Philox keep-masks are seeded from torch's RNG
    x, y = m.parse_batch(batch)
    loss = crit(m(x), y)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, m.last_train_decoder_bwd_path


def compare(tag, batch):
    torch.manual_seed(1234)
    m = Tacotron2(hp).to(dev).train()
    m.precision = "bf16"
    l0, g0, p0 = one_step(m, batch, False)
    l1, g1, p1 = one_step(m, batch, True)
    l2, g2, p2 = one_step(m, batch, True)
    diff = [k for k in g0 if not (torch.equal(g0[k], g1[k]) and torch.equal(g0[k], g2[k]))]
    row = {"paths": [p0, p1, p2], "loss": [float(l0), float(l1), float(l2)], "gradients_differing": diff,
           "max_rel_diff": {k: float(((g0[k] - g1[k]).abs().max() / (g0[k].abs().max() + 1e-30))) for k in diff[:8]},
           "finite": bool(all(torch.isfinite(v).all() for v in g1.values())), "give_ups": native.attn_handoff_timeouts(reset=False)}
    row["ok"] = p0 == "launch chain" and p1 == "persistent" and not diff and float(l0) == float(l1)
    out[tag] = row
    print(tag, json.dumps(row)[:900], flush=True)
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
opt = FusedAdam(m.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)


def block(persistent):
    engine.TRAIN_BWD_PERSISTENT = persistent
    for i in range(2):                                             # warm
        m.zero_grad(); x, y = m.parse_batch(batches[i]); crit(m(x), y).backward(); opt.step(clip_norm=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        m.zero_grad(); x, y = m.parse_batch(batches[i]); crit(m(x), y).backward(); opt.step(clip_norm=1.0)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / a.steps


times = {"chain": [], "persistent": []}
for _ in range(a.blocks):
    times["chain"].append(block(False))
    times["persistent"].append(block(True))
out["ms_per_training_step"] = times
print(json.dumps(times), flush=True)
if not a.no_sweep:
    sweep = {}
    for d1, d2 in ((4, 4), (0, 0), (0, 4), (4, 0), (8, 8), (16, 16), (2, 2)):
        os.environ["T2AMD_DBP_DELAY_1"], os.environ["T2AMD_DBP_DELAY_2"] = str(d1), str(d2)
        sweep["K1_%d_K2_%d" % (d1, d2)] = min(block(True) for _ in range(2))
    del os.environ["T2AMD_DBP_DELAY_1"], os.environ["T2AMD_DBP_DELAY_2"]
    out["ms_per_training_step_by_prepoll_pause"] = sweep
    print("pause sweep (whole step, ms):", json.dumps(sweep), flush=True)
# phase clocks of workgroup 0 (100 MHz wall clock)
lib.t2amd_debug_dtp_prof_.argtypes = [C.c_void_p]
prof = torch.zeros(8, dtype=torch.int64, device=dev)
engine.TRAIN_FWD_PERSISTENT = False               # (the forward launch would add its own clocks to the same slots)
engine.TRAIN_BWD_PERSISTENT = True
m.zero_grad(); x, y = m.parse_batch(batches[0]); loss = crit(m(x), y)
torch.cuda.synchronize()
lib.t2amd_debug_dtp_prof_(C.c_void_p(prof.data_ptr()))
loss.backward()
torch.cuda.synchronize()
lib.t2amd_debug_dtp_prof_(None)
To = int(batches[0][4].max())
pc = [v / 100.0 / To for v in prof.tolist()[:4]]
out["phase_clocks_us_per_time_step_workgroup0"] = {"attention_backward_and_cells_incl_wait_for_dgrad_flags": pc[0],
                                                   "dgrad_tile_incl_drain_flag1_and_wait_for_attention_flags": pc[2],
                                                   "drain_and_flag2": pc[3], "sum": sum(pc)}
print("phase clocks (us per time step):", json.dumps(out["phase_clocks_us_per_time_step_workgroup0"]))
out["give_ups"] = native.attn_handoff_timeouts(reset=False)
print("give-ups:", out["give_ups"])
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/ab_train_bwd_persistent.json", "w") as fh:
    json.dump(out, fh, indent=1)
