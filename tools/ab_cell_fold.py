"""A/B of the forms of the decoder BPTT step on the BASELINE configs[1] training step, in ONE process: the same model,
batches and dropout seeds run under each configuration "fold,gran[,fwd]" -- fold: the LSTM cell backwards as a launch of their
own (0) or inside the attention-backward launch (1, t2amd_set_bptt_cell_fold); gran: the first hand-off of that launch as
drained stores + token (0) or as {token, value} granules (1, t2amd_set_attn_bwd_granules); fwd: the attention forward
of a step as two launches (0) or one (1, t2amd_set_attn_fwd_fused) -- alternating, and the
gradients of every form are compared bit for bit with the first one at full size (B = 64, To <= 870).

    python tools/ab_cell_fold.py [--configs "0,0;1,0;1,1"] [--steps 6] [--rounds 3] [--precision bf16]

Prints one JSON line: ms per step of each form per round, the medians, and `bitwise_equal`.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--precision", default="bf16", choices=("fp32", "bf16"))
    ap.add_argument("--configs", default="0,0;1,0;1,1", help='";"-separated "fold,gran" pairs; the first is the reference')
    args = ap.parse_args()
    configs = [tuple(int(v) for v in c.split(",")) for c in args.configs.split(";")]
    configs = [c if len(c) == 3 else c + (0,) for c in configs]
    from tacotron2_amd import native
    native.load()
    from tacotron2_amd.hparams import create_hparams
    from tacotron2_amd.model import Tacotron2
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.optim import FusedAdam
    from tacotron2_amd.synth import synth_batch
    dev = torch.device("cuda", 0)
    hp = create_hparams()
    hp.batch_size = args.batch_size
    torch.manual_seed(hp.seed)
    model = Tacotron2(hp).to(dev)
    model.precision = args.precision
    model.train()
    criterion = Tacotron2Loss()
    batches = [tuple(t.to(dev) for t in synth_batch(args.batch_size, 1234 + i)) for i in range(args.steps)]
    start = native.get_bptt_cell_fold()

    def select(cfg):
        native.set_bptt_cell_fold(cfg[0])
        native.set_attn_bwd_granules(cfg[1])
        native.set_attn_fwd_fused(cfg[2])

    def grads_of(cfg, seed):
        select(cfg)
        torch.manual_seed(seed)                      # same dropout masks in every form
        model.zero_grad()
        x, y = model.parse_batch(batches[0])
        loss = criterion(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.item()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    try:
        l0, g0 = grads_of(configs[0], 5)
        equal, finite, worst, losses = True, True, 0.0, [l0]
        for cfg in configs[1:] + configs[-1:]:       # the last form twice: run-to-run determinism
            l1, g1 = grads_of(cfg, 5)
            losses.append(l1)
            worst = max(worst, max(float((g0[k] - g1[k]).abs().max()) for k in g0))
            equal = equal and l1 == l0 and all(torch.equal(g0[k], g1[k]) for k in g0)
            finite = finite and all(bool(torch.isfinite(g).all()) for g in g1.values())

        optimizer = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)

        def timed(cfg):
            select(cfg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches:
                model.zero_grad()
                x, y = model.parse_batch(b)
                criterion(model(x), y).backward()
                optimizer.step(clip_norm=hp.grad_clip_thresh)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / len(batches)

        for cfg in configs:
            timed(cfg)                                # warm-up of every form
        ms = {cfg: [] for cfg in configs}
        for _ in range(args.rounds):
            for cfg in configs:
                ms[cfg].append(timed(cfg))
    finally:
        native.set_bptt_cell_fold(start)
        native.set_attn_bwd_granules(-1)
        native.set_attn_fwd_fused(-1)
    frames = sum(int(b[4].sum()) for b in batches) / len(batches)
    out = {"workload": "BASELINE configs[1] training step, B=%d, %s" % (args.batch_size, args.precision),
           "steps_per_round": args.steps,
           "forms": {"fold=%d,granules=%d,fwd_fused=%d" % cfg: {"ms_per_step": v, "median_ms": statistics.median(v),
                                                   "frames_per_s": frames / statistics.median(v) * 1e3}
                     for cfg, v in ms.items()},
           "bitwise_equal": bool(equal), "finite": finite, "worst_abs_diff": worst, "loss": losses}
    print(json.dumps(out))
    return 0 if (equal and finite) else 1


if __name__ == "__main__":
    sys.exit(main())
