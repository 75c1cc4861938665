"""BPTT of the encoder bi-LSTM of a batch: the chain of 2 T launches against the ONE persistent launch, and the pre-poll pause
(T2AMD_EBB_DELAY) of the latter.  timeout 120 python tools/microbench_encoder_bwd_persistent.py  -> gpurun_out/microbench_encoder_bwd_persistent.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native as nv

nv.load()
DEV = "cuda"
out = {}
for B, T in ((64, 177), (64, 100), (32, 177)):
    H, E = 256, 512
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(T // 3, T + 1, (B,), generator=g).sort(descending=True)[0].to(torch.int32)
    lens[0] = T
    lens = lens.to(DEV)
    Whh = [(torch.randn(4 * H, H, generator=g) * 0.06).to(DEV) for _ in range(2)]
    WhhT = [w.t().contiguous() for w in Whh]
    GX = [(torch.rand(B * T, 4 * H, generator=g)).to(DEV) for _ in range(2)]
    Cst = [(torch.randn(T, B, H, generator=g) * 0.3).to(DEV) for _ in range(2)]
    dmem = (torch.randn(B, T, E, generator=g) * 0.3).to(DEV)
    descs, keep = [], []
    for d in range(2):
        DG = torch.zeros(B * T, 4 * H, device=DEV)
        dX, dc = torch.zeros(4, B, H, device=DEV), torch.zeros(B, H, device=DEV)
        desc = nv.LstmSeq()
        desc.B, desc.T, desc.H, desc.reverse = B, T, H, d
        desc.WhhT = nv.ptr(WhhT[d])
        desc.GX, desc.C, desc.lens = nv.ptr(GX[d]), nv.ptr(Cst[d]), nv.ptr(lens, torch.int32)
        desc.dout, desc.ld_dout = nv.ptr(dmem.view(B * T, E)[:, d * H:(d + 1) * H]), E
        desc.DG = nv.ptr(DG)
        desc.dX, desc.dc, desc.dx_splits = nv.ptr(dX), nv.ptr(dc), 4
        descs.append(desc); keep.append((DG, dX, dc))
    flags = torch.zeros(nv.lstm_seq_batch_persistent_flag_words(B, H, 2), dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    row = {"chain_ms": timed(lambda: nv.lstm_seq_bwd2(descs[0], descs[1]))}
    for delay in (0, 4, 8, 16, 32, 64):
        os.environ["T2AMD_EBB_DELAY"] = str(delay)
        row["persistent_ms_delay_%d" % delay] = timed(lambda: nv.lstm_seq_bwd2_batch_persistent(descs[0], descs[1], flags, status))
        assert int(status.item()) == 0
    row["us_per_step_chain"] = 1e3 * row["chain_ms"] / T
    row["us_per_step_persistent_best"] = 1e3 * min(v for k, v in row.items() if k.startswith("persistent")) / T
    out["B%d_T%d" % (B, T)] = row
    print("B=%d T=%d" % (B, T), json.dumps({k: round(v, 3) for k, v in row.items()}))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/microbench_encoder_bwd_persistent.json", "w"), indent=1)
