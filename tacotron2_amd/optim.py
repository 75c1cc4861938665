"""Global-norm clipping + Adam as two HIP launches.

``FusedAdam`` is a ``torch.optim.Adam`` whose ``step`` runs ``csrc/optim.hip`` instead of torch's
~12 multi-tensor passes.  It *is* an Adam: same constructor, same ``param_groups`` and the same
per-parameter state (``step``, ``exp_avg``, ``exp_avg_sq``), so ``state_dict()`` / ``load_state_dict``
— the ``optimizer`` entry of the reference's checkpoints (train.py:112-118) — interchange with a plain
``torch.optim.Adam`` in both directions.

    optimizer = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
    loss.backward()
    grad_norm = optimizer.step(clip_norm=hp.grad_clip_thresh)   # replaces clip_grad_norm_ + step (train.py:233-236)

Differences from the torch pair, all deliberate:
  * the clip factor is applied while the gradient is read: ``p.grad`` itself is left unscaled (the
    reference's loop never looks at it again before ``zero_grad``);
  * ``step()`` without ``clip_norm`` is a plain Adam step; ``clip_norm`` returns the pre-clip global
    norm as a 0-d device tensor, like ``clip_grad_norm_``;
  * with ``clip_norm``, a non-finite global norm skips the update ON THE DEVICE (weights and moments untouched,
    no host synchronisation); the step counts (bias correction) are corrected by the next ``step()`` -- which reads the
    norm from an asynchronous pinned copy -- or by ``undo_step_count()`` if the caller read the norm first; either way only
    the counters of the parameters that step covered are touched.
Unsupported options raise (amsgrad, maximize, capturable, differentiable, decoupled weight decay,
sparse or non-f32 gradients): there is no silent fallback to torch's implementation.
"""
import math

import torch

from . import native as nv


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False)
        self._ws = None
        self._norm = None
        self._norm_host = None        # pinned copy of the last step's [norm, coef]: read without a synchronisation
        self._norm_event = None
        self._pending = None          # step counters the last clipped step incremented, until its norm is known

    def _check_group(self, group):
        for flag in ('amsgrad', 'maximize', 'capturable', 'differentiable', 'decoupled_weight_decay'):
            if group.get(flag):
                raise nv.NativeError("FusedAdam: option %s=True is not implemented by the HIP kernel" % flag)
        if isinstance(group['lr'], torch.Tensor):
            raise nv.NativeError("FusedAdam: tensor learning rates are not supported")

    @torch.no_grad()
    def step(self, closure=None, clip_norm=None):
        if closure is not None:
            raise nv.NativeError("FusedAdam: closures are not supported")
        self._settle_previous()
        work = []                                            # (group, params, grads, exp_avgs, exp_avg_sqs, steps)
        for group in self.param_groups:
            self._check_group(group)
            ps, gs, ms, vs, steps = [], [], [], [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.grad.dtype != torch.float32 or p.dtype != torch.float32:
                    raise nv.NativeError("FusedAdam: dense float32 parameters and gradients only")
                st = self.state[p]
                if len(st) == 0:                             # torch.optim.Adam's own lazy state layout
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not (p.is_contiguous() and st['exp_avg'].is_contiguous() and st['exp_avg_sq'].is_contiguous()):
                    raise nv.NativeError("FusedAdam: parameters and moments must be contiguous")
                ps.append(p)
                gs.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
                ms.append(st['exp_avg'])
                vs.append(st['exp_avg_sq'])
                steps.append(st['step'])
            if ps:
                work.append((group, ps, gs, ms, vs, steps))
        if not work:
            return None

        dev = work[0][1][0].device
        norm = None
        if clip_norm is not None:
            all_g = [g for w in work for g in w[2]]
            if len(all_g) > nv.MAX_TENSORS:
                raise nv.NativeError("FusedAdam: clipping covers at most %d tensors per step, got %d"
                                     % (nv.MAX_TENSORS, len(all_g)))
            L, blocks = nv.tensor_list(all_g)
            if self._ws is None or self._ws.numel() < blocks or self._ws.device != dev:
                self._ws = torch.empty(max(blocks, 1024), dtype=torch.float64, device=dev)
            self._norm = torch.empty(2, dtype=torch.float32, device=dev)
            nv.grad_norm(L, blocks, float(clip_norm), self._ws, self._norm)
            norm = self._norm
            # the host learns whether this step was skipped WITHOUT waiting for it: an asynchronous copy to pinned memory
            # and an event, looked at by the next step() (or by undo_step_count(), whichever comes first)
            if dev.type == 'cuda':
                if self._norm_host is None:
                    self._norm_host = torch.zeros(2, dtype=torch.float32).pin_memory()
                    self._norm_event = torch.cuda.Event()
                self._norm_host.copy_(self._norm, non_blocking=True)
                self._norm_event.record()
            else:                                 # host tensors (the CPU emulation of the kernels in tests/): nothing to wait for
                self._norm_host, self._norm_event = self._norm, None

        for group, ps, gs, ms, vs, steps in work:
            beta1, beta2 = group['betas']
            for lo in range(0, len(ps), nv.MAX_TENSORS):
                hi = min(lo + nv.MAX_TENSORS, len(ps))
                # all tensors of one launch must share the step count (they do unless state was edited by hand)
                t = float(steps[lo]) + 1.0
                if any(float(s) + 1.0 != t for s in steps[lo:hi]):
                    raise nv.NativeError("FusedAdam: parameters of one group are at different step counts")
                h = nv.AdamHyper()
                h.step_size = float(group['lr']) / (1.0 - beta1 ** t)
                h.bc2_sqrt = math.sqrt(1.0 - beta2 ** t)
                h.one_minus_beta1 = 1.0 - beta1
                h.beta2 = beta2
                h.one_minus_beta2 = 1.0 - beta2
                h.eps = float(group['eps'])
                h.weight_decay = float(group['weight_decay'])
                L, _ = nv.tensor_list(gs[lo:hi], ps[lo:hi], ms[lo:hi], vs[lo:hi])
                nv.adam_step(L, h, norm)
            for s in steps:
                s += 1.0
        if norm is not None:
            self._pending = dict(steps=[s for w in work for s in w[5]], undone=False)
        # the kernels update the weights through raw pointers: torch's version counters do not see it
        from . import engine
        engine.bump_weight_generation()
        return norm[0] if norm is not None else None

    def _undo(self):
        if self._pending is not None and not self._pending['undone']:
            for s in self._pending['steps']:             # exactly the counters that step incremented, no others
                s -= 1.0
            self._pending['undone'] = True

    def _settle_previous(self):
        """A clipped step whose global norm turned out non-finite was skipped on the device: take it out of the step counts
        (bias correction) of exactly the parameters it covered.  Its norm was copied to pinned memory when it ran; by the
        time the next step is enqueued that copy has landed (a whole forward / backward lies between), so this does not
        stall the host -- the wait below only ever triggers for back-to-back steps."""
        if self._pending is None or self._pending['undone']:
            return
        if self._norm_event is not None and not self._norm_event.query():
            self._norm_event.synchronize()
        if not math.isfinite(float(self._norm_host[0])):
            self._undo()
        self._pending = None

    def state_dict(self):
        """A checkpoint taken between a device-skipped step and the next ``step()`` must not carry step counts one too high
        (bias correction after resume): the skipped step is settled first."""
        self._settle_previous()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._pending = None             # its step tensors belong to the state that is being replaced
        return super().load_state_dict(state_dict)

    def undo_step_count(self):
        """The last ``step(clip_norm=...)`` was skipped on the device (non-finite norm): take it out of the counts.  Kept for
        callers that read the norm themselves (tacotron2_amd.train); since round 3 the next ``step()`` does this on its own,
        and only the counters of the parameters that step covered are touched."""
        self._undo()
