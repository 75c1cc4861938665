"""A/B of the BPTT cell fold (t2amd_set_bptt_cell_fold) on the BASELINE configs[1] training step, in ONE process:
the same model, batches and dropout seeds run with the LSTM cell backwards as a launch of their own (0) and inside the
attention-backward launch (1), alternating, and the gradients of the two forms are compared bit for bit at full size
(B = 64, To <= 870).

    python tools/ab_cell_fold.py [--steps 6] [--rounds 3] [--precision bf16]

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
    args = ap.parse_args()
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

    def grads_of(fold, seed):
        native.set_bptt_cell_fold(fold)
        torch.manual_seed(seed)                      # same dropout masks in both forms
        model.zero_grad()
        x, y = model.parse_batch(batches[0])
        loss = criterion(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.item()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    try:
        l0, g0 = grads_of(0, 5)
        l1, g1 = grads_of(1, 5)
        l2, g2 = grads_of(1, 5)
        worst = max(float((g0[k] - g1[k]).abs().max()) for k in g0)
        equal = all(torch.equal(g0[k], g1[k]) and torch.equal(g1[k], g2[k]) for k in g0) and l0 == l1 == l2
        finite = all(bool(torch.isfinite(g).all()) for g in g1.values())

        optimizer = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)

        def timed(fold):
            native.set_bptt_cell_fold(fold)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches:
                model.zero_grad()
                x, y = model.parse_batch(b)
                criterion(model(x), y).backward()
                optimizer.step(clip_norm=hp.grad_clip_thresh)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / len(batches)

        timed(0), timed(1)                            # warm-up of both forms
        ms = {0: [], 1: []}
        for _ in range(args.rounds):
            for fold in (0, 1):
                ms[fold].append(timed(fold))
    finally:
        native.set_bptt_cell_fold(start)
    frames = sum(int(b[4].sum()) for b in batches) / len(batches)
    out = {"workload": "BASELINE configs[1] training step, B=%d, %s" % (args.batch_size, args.precision),
           "steps_per_round": args.steps, "ms_per_step_unfolded": ms[0], "ms_per_step_folded": ms[1],
           "median_unfolded": statistics.median(ms[0]), "median_folded": statistics.median(ms[1]),
           "frames_per_s_unfolded": frames / statistics.median(ms[0]) * 1e3,
           "frames_per_s_folded": frames / statistics.median(ms[1]) * 1e3,
           "bitwise_equal": bool(equal), "finite": finite, "worst_abs_diff": worst, "loss": [l0, l1, l2]}
    print(json.dumps(out))
    return 0 if (equal and finite) else 1


if __name__ == "__main__":
    sys.exit(main())
