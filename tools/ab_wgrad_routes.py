"""Gradients of one BASELINE configs[1] training step (bf16 mode) under the two weight-gradient routes, in ONE process on the
same model, batch and dropout seed: engine.WGRAD_KK = True (K-major products straight from the slabs / halo images) against
False (round 2: transposed copies, f32-source convolution gradients).  Prints, per parameter, the difference relative to the
gradient's max -- both routes round the same operands to bf16, so they differ by summation order only.
    python tools/ab_wgrad_routes.py [--batch-size 64]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-size", type=int, default=64)
    args = ap.parse_args()
    from tacotron2_amd import engine, native
    native.load()
    from tacotron2_amd.hparams import create_hparams
    from tacotron2_amd.model import Tacotron2
    from tacotron2_amd.loss_function import Tacotron2Loss
    from tacotron2_amd.synth import synth_batch
    dev = torch.device("cuda", 0)
    hp = create_hparams()
    hp.batch_size = args.batch_size
    torch.manual_seed(hp.seed)
    model = Tacotron2(hp).to(dev)
    model.precision = "bf16"
    model.train()
    criterion = Tacotron2Loss()
    batch = tuple(t.to(dev) for t in synth_batch(args.batch_size, 1234))

    def grads_of(kk):
        engine.WGRAD_KK = kk
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        torch.manual_seed(5)
        model.zero_grad()
        x, y = model.parse_batch(batch)
        loss = criterion(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        g = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        model.load_state_dict(sd)                    # BatchNorm running statistics back to the start
        return float(loss.item()), g

    l1, g1 = grads_of(True)
    l0, g0 = grads_of(False)
    l1b, g1b = grads_of(True)
    rel = {k: float((g1[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)) for k in g0}
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:8]
    print(json.dumps({"loss_kk": l1, "loss_round2_route": l0, "kk_run_to_run_bitwise": all(torch.equal(g1[k], g1b[k]) for k in g1),
                      "worst_relative_difference": worst,
                      "finite": all(bool(torch.isfinite(v).all()) for v in g1.values())}))


if __name__ == "__main__":
    main()
