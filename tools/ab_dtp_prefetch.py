"""A/B of the persistent forward loop with and without the next step's first k-tiles fetched from inside the attention phase
(csrc/skinny_wide.h skinny_wide_prefetch4; T2AMD_DTP_PREFETCH=0/1 is read per call) -- one process, same model, batches, masks.

    timeout 600 python tools/ab_dtp_prefetch.py          # writes gpurun_out/ab_dtp_prefetch.json

1. bitwise: outputs, loss and all 60 gradients of one B = 64 / To = 870 training step: launch chain vs persistent without vs with;
2. time: forward only and whole training steps, alternating blocks; in-kernel phase clocks of workgroup 0 and of LSTM_d tile 0.
"""
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

STEPS = int(os.environ.get("AB_STEPS", "6"))
BLOCKS = int(os.environ.get("AB_BLOCKS", "3"))
lib = native.load()
dev = torch.device("cuda", 0)
hp = create_hparams()
crit = Tacotron2Loss()
out = {}


def select(form):
    engine.TRAIN_FWD_PERSISTENT = form != "chain"
    os.environ["T2AMD_DTP_PREFETCH"] = "1" if form == "prefetch" else "0"


def one_step(m, batch, form, seed=99):
    select(form)
    m.zero_grad()
    torch.manual_seed(seed)
    x, y = m.parse_batch(batch)
    o = m(x)
    loss = crit(o, y)
    loss.backward()
    torch.cuda.synchronize()
    return [t.detach().clone() for t in o], loss.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, m.last_train_decoder_path


batches = [tuple(t.to(dev) for t in synth_batch(64, 1234 + i)) for i in range(STEPS)]
torch.manual_seed(1234)
m = Tacotron2(hp).to(dev).train()
m.precision = os.environ.get("AB_PRECISION", "bf16")
res = {f: one_step(m, batches[0], f) for f in ("chain", "plain", "prefetch", "prefetch")}
ref = res["chain"]
row = {"paths": {f: r[3] for f, r in res.items()}, "loss": {f: float(r[1]) for f, r in res.items()}}
for f in ("plain", "prefetch"):
    r = res[f]
    row[f + "_bit_identical_to_chain"] = bool(all(torch.equal(a, b) for a, b in zip(ref[0], r[0])) and torch.equal(ref[1], r[1])
                                              and all(torch.equal(ref[2][k], r[2][k]) for k in ref[2]))
out["bitwise_B64_To870"] = row
print(json.dumps(row), flush=True)
opt = FusedAdam(m.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)


def block(form):
    select(form)
    for i in range(2):
        m.zero_grad(); x, y = m.parse_batch(batches[i]); crit(m(x), y).backward(); opt.step(clip_norm=1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        m.zero_grad(); x, y = m.parse_batch(batches[i]); crit(m(x), y).backward(); opt.step(clip_norm=1.0)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / STEPS


def fwd_only(form):
    select(form)
    with torch.no_grad():
        x, y = m.parse_batch(batches[0])
        m(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            m(x)
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / STEPS


times = {k: [] for k in ("step_chain", "step_plain", "step_prefetch", "fwd_chain", "fwd_plain", "fwd_prefetch")}
for _ in range(BLOCKS):
    for f in ("chain", "plain", "prefetch"):
        times["step_" + f].append(round(block(f), 3))
    for f in ("chain", "plain", "prefetch"):
        times["fwd_" + f].append(round(fwd_only(f), 3))
out["ms"] = times
print(json.dumps(times), flush=True)
# phase clocks (100 MHz ticks accumulated over the launch): [wg0: wait T flags, tile + publish, wait A flags, attention] [LSTM_d tile 0: same]
lib.t2amd_debug_dtp_prof_.argtypes = [C.c_void_p]
To = batches[0][2].shape[2]
for f in ("plain", "prefetch"):
    prof = torch.zeros(8, dtype=torch.int64, device=dev)
    lib.t2amd_debug_dtp_prof_(C.c_void_p(prof.data_ptr()))
    select(f)
    with torch.no_grad():
        m(m.parse_batch(batches[0])[0])
    torch.cuda.synchronize()
    lib.t2amd_debug_dtp_prof_(None)
    out["phase_us_per_time_step_" + f] = [round(v / 100.0 / To, 3) for v in prof.tolist()]
    print(f, out["phase_us_per_time_step_" + f], flush=True)
# TIMING ONLY (not legal: h_dec is never produced): the L phase with the attention LSTM alone -- the ceiling of taking the decoder
# LSTM off the loop's critical path
os.environ["T2AMD_DTP_TIMING_NO_D"] = "1"
out["timing_only_no_decoder_lstm_tiles"] = {"fwd_plain_ms": [round(fwd_only("plain"), 3) for _ in range(2)],
                                            "fwd_prefetch_ms": [round(fwd_only("prefetch"), 3) for _ in range(2)]}
prof = torch.zeros(8, dtype=torch.int64, device=dev)
lib.t2amd_debug_dtp_prof_(C.c_void_p(prof.data_ptr()))
select("prefetch")
with torch.no_grad():
    m(m.parse_batch(batches[0])[0])
torch.cuda.synchronize()
lib.t2amd_debug_dtp_prof_(None)
out["timing_only_no_decoder_lstm_tiles"]["phase_us_per_time_step_prefetch"] = [round(v / 100.0 / To, 3) for v in prof.tolist()]
del os.environ["T2AMD_DTP_TIMING_NO_D"]
print("timing only, no LSTM_d tiles:", json.dumps(out["timing_only_no_decoder_lstm_tiles"]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/ab_dtp_prefetch_%s.json" % m.precision, "w") as fh:
    json.dump(out, fh, indent=1)
