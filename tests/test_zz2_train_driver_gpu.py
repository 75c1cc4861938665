"""The training driver end to end on the engine."""
import json
import os

import pytest
import torch

import golden_util as gu
from tacotron2_amd.hparams import create_hparams

pytestmark = pytest.mark.gpu


def test_train_driver_runs_on_the_engine(native_lib, tmp_path):
    from tacotron2_amd import train as tr
    hpstr = gu.TINY_HP + ",batch_size=2,iters_per_checkpoint=2,epochs=2,training_files=synthetic:6:3:60," \
                         "validation_files=synthetic:3:4:60"
    out = tmp_path / "run"
    last = tr.train(str(out), "logs", None, False, 1, 0, "g", create_hparams(hpstr), max_iterations=3)
    assert last == 2 and os.path.exists(out / "checkpoint_2")
    recs = [json.loads(l) for l in open(out / "logs" / "scalars.jsonl")]
    tl = [r["training.loss"] for r in recs if "training.loss" in r]
    vl = [r["validation.loss"] for r in recs if "validation.loss" in r]
    assert len(tl) == 3 and len(vl) == 2 and all(0.0 < v < 100.0 for v in tl + vl)
    gn = [r["grad.norm"] for r in recs if "grad.norm" in r]
    assert all(0.0 < v < 1e6 for v in gn)
    # resume: optimiser state and iteration counter come from the checkpoint
    last = tr.train(str(out), "logs", str(out / "checkpoint_2"), False, 1, 0, "g", create_hparams(hpstr),
                    max_iterations=5)
    assert last == 4
    recs = [json.loads(l) for l in open(out / "logs" / "scalars.jsonl")]
    tl = [r["training.loss"] for r in recs if "training.loss" in r]
    assert len(tl) == 5 and all(0.0 < v < 100.0 for v in tl)
    ck = torch.load(out / "checkpoint_4", weights_only=False)
    assert ck["iteration"] == 4 and len(ck["state_dict"]) == 84
    # a reference-format checkpoint written here loads into a fresh model
    m = tr.load_model(create_hparams(hpstr))
    tr.warm_start_model(str(out / "checkpoint_4"), m, [])
    assert all(torch.equal(v.cpu(), ck["state_dict"][k].cpu()) for k, v in m.state_dict().items())


def test_train_driver_precision_from_the_environment(native_lib, tmp_path, monkeypatch):
    """T2AMD_PRECISION picks the engine's compute mode for the driver (the reference's hparams have no word for 'bf16x3').  Same
    seed, same data, same Philox dropout masks: the f32-class split mode follows the fp32 run's losses, iteration by iteration."""
    from tacotron2_amd import train as tr
    hpstr = gu.TINY_HP + ",batch_size=2,iters_per_checkpoint=100,epochs=2,training_files=synthetic:6:3:60," \
                         "validation_files=synthetic:3:4:60"
    losses = {}
    for prec in ("fp32", "bf16x3", "bf16"):
        monkeypatch.setenv("T2AMD_PRECISION", prec)
        assert tr.load_model(create_hparams(hpstr)).precision == prec
        out = tmp_path / prec
        tr.train(str(out), "logs", None, False, 1, 0, "g", create_hparams(hpstr), max_iterations=4)
        recs = [json.loads(l) for l in open(out / "logs" / "scalars.jsonl")]
        losses[prec] = [r["training.loss"] for r in recs if "training.loss" in r]
        assert len(losses[prec]) == 4
    for a, b in zip(losses["fp32"], losses["bf16x3"]):
        assert abs(a - b) <= 2e-4 * abs(a), losses
    for a, b in zip(losses["fp32"], losses["bf16"]):
        assert abs(a - b) <= 5e-2 * abs(a), losses
    monkeypatch.setenv("T2AMD_PRECISION", "fp16")
    with pytest.raises(Exception, match="T2AMD_PRECISION"):
        tr.load_model(create_hparams(hpstr))
