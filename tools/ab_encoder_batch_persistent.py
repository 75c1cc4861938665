"""Encoder bi-LSTM forward of a batch: the launch chain (T dependent launches) against the persistent launch with flag + data
hand-offs -- same inputs, outputs compared (gates, cell states, h), both timed.
    python tools/ab_encoder_batch_persistent.py [--B 64] [--T 170]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv      # noqa: E402


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


B, T, H, E = arg("--B", 64), arg("--T", 170), 256, 512
dev = torch.device("cuda")
nv.load()
g = torch.Generator().manual_seed(0)
lens = torch.randint(T // 3, T + 1, (B,), generator=g).sort(descending=True)[0].to(torch.int32)
lens[0] = T
lens = lens.to(dev)
Whh = [(torch.randn(4 * H, H, generator=g) * 0.06).to(dev) for _ in range(2)]
GX0 = [(torch.randn(B * T, 4 * H, generator=g) * 0.5).to(dev) for _ in range(2)]


def make(which):
    mem = torch.full((B, T, E), float('nan'), device=dev)
    descs, keep = [], []
    for d in range(2):
        GX, Cst = GX0[d].clone(), torch.full((T, B, H), float('nan'), device=dev)
        desc = nv.LstmSeq()
        desc.B, desc.T, desc.H, desc.reverse = B, T, H, d
        desc.Whh, desc.GX = nv.ptr(Whh[d]), nv.ptr(GX)
        ov = mem.view(B * T, E)[:, d * H:(d + 1) * H]
        desc.out, desc.ld_out = nv.ptr(ov), E
        desc.C, desc.lens = nv.ptr(Cst), nv.ptr(lens, torch.int32)
        descs.append(desc)
        keep.append((GX, Cst))
    return mem, descs, keep


flags = torch.zeros(2 * ((B + 31) // 32) * (H // 4), dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)


def run(which, descs):
    if which == "chain":
        nv.lstm_seq_fwd2(descs[0], descs[1])
    else:
        nv.lstm_seq_fwd2_batch_persistent(descs[0], descs[1], flags, status)


def timed(which, n=5):
    best = None
    for _ in range(n):
        mem, descs, keep = make(which)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(which, descs)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best, mem, keep


out = {"B": B, "T": T, "H": H}
tc, mc, kc = timed("chain")
tp, mp, kp = timed("persistent")
out["chain_ms"], out["persistent_ms"], out["status"] = tc, tp, int(status.item())
out["us_per_step"] = {"chain": 1e3 * tc / T, "persistent": 1e3 * tp / T}
out["max_abs_diff"] = {"h": float((mc - mp).abs().max()), "gates_fwd": float((kc[0][0] - kp[0][0]).abs().max()),
                       "gates_rev": float((kc[1][0] - kp[1][0]).abs().max()), "c_fwd": float((kc[0][1] - kp[0][1]).abs().max()),
                       "c_rev": float((kc[1][1] - kp[1][1]).abs().max())}
out["finite"] = bool(torch.isfinite(mp).all())
print(json.dumps(out))
