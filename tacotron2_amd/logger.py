"""Run log with the call surface ``train.py`` uses (reference logger.py:8-48):
``Tacotron2Logger(logdir).log_training(loss, grad_norm, lr, duration, iteration)`` and
``.log_validation(loss, model, y, y_pred, iteration)``.

Observability only — no arithmetic of the hot path.  TensorBoard is not on the MI355X image, so
scalars go to ``<logdir>/scalars.jsonl`` (one JSON object per call); when ``tensorboard`` is
importable the same scalars are mirrored into an event file.  The reference's per-iteration
parameter histograms and alignment/mel images (logger.py:24-48, plotting_utils.py) force a D2H
copy of all 28 M parameters per validation: here validation logs the scalar and, per tensor,
only its mean / rms (computed on the device, one small copy).
"""
import json
import os
import time


class Tacotron2Logger(object):
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.logdir = logdir
        self._fh = open(os.path.join(logdir, 'scalars.jsonl'), 'a', buffering=1)
        self._tb = None
        try:                                              # optional mirror
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(logdir)
        except Exception:
            self._tb = None

    def _emit(self, iteration, **scalars):
        rec = {'iteration': int(iteration), 'time': time.time()}
        rec.update({k: float(v) for k, v in scalars.items()})
        self._fh.write(json.dumps(rec) + '\n')
        if self._tb is not None:
            for k, v in scalars.items():
                self._tb.add_scalar(k, float(v), iteration)

    def log_training(self, reduced_loss, grad_norm, learning_rate, duration, iteration):
        self._emit(iteration, **{'training.loss': reduced_loss, 'grad.norm': grad_norm,
                                 'learning.rate': learning_rate, 'duration': duration})

    def log_validation(self, reduced_loss, model, y, y_pred, iteration):
        """``y`` / ``y_pred`` are accepted for signature compatibility (the reference plots them)."""
        import torch
        stats = {'validation.loss': reduced_loss}
        named = list(model.named_parameters()) if model is not None else []
        if named:
            with torch.no_grad():
                means = [p.detach().float().mean() for _, p in named]
                rms = [p.detach().float().pow(2).mean().sqrt() for _, p in named]
                packed = torch.stack(means + rms).cpu().tolist()       # one device-to-host copy
            for i, (name, _) in enumerate(named):
                stats['param.mean/' + name] = packed[i]
                stats['param.rms/' + name] = packed[len(named) + i]
        self._emit(iteration, **stats)

    def close(self):
        self._fh.close()
        if self._tb is not None:
            self._tb.close()
