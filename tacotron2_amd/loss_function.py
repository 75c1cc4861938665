"""Tacotron2Loss for the MI355X engine.

Contract taken from reference loss_function.py:8-19: the scalar is
``mean((mel - y)^2) + mean((mel_postnet - y)^2) + mean(bce_with_logits(gate, g))``
with every mean taken over the *padded* tensors (padded frames
contribute exactly zero because ``Tacotron2.parse_output`` forced them to the
padding targets).  The criterion sits outside the model boundary (the training
loop applies it to the model's outputs).

On the GPU it is one ``torch.autograd.Function`` over two HIP kernels (csrc/loss.hip): one
reduction pass over the three mel-shaped tensors and the gate vectors (double-precision partial sums, fixed order:
bit-reproducible), one gradient pass scaled by the upstream gradient on the device.  CPU tensors (the kernels-off test
mode, the reference's own CPU use of a criterion) take the reference's three torch calls.
"""
import torch
import torch.nn.functional as F

from . import native as nv


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mel, post, gate, mel_target, gate_target):
        mel, post, gate = mel.contiguous(), post.contiguous(), gate.contiguous()
        mel_target, gate_target = mel_target.contiguous(), gate_target.contiguous()
        ws = torch.empty(nv.loss_workspace_doubles(), dtype=torch.float64, device=mel.device)
        out4 = torch.empty(4, dtype=torch.float32, device=mel.device)
        nv.tacotron2_loss_fwd(mel, post, mel_target, gate, gate_target, ws, out4)
        ctx.save_for_backward(mel, post, gate, mel_target, gate_target)
        return out4[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        mel, post, gate, mel_target, gate_target = ctx.saved_tensors
        d_mel, d_post, d_gate = torch.empty_like(mel), torch.empty_like(post), torch.empty_like(gate)
        nv.tacotron2_loss_bwd(mel, post, mel_target, gate, gate_target, grad_out.reshape(1).float().contiguous(),
                              d_mel, d_post, d_gate)
        return d_mel, d_post, d_gate, None, None


class Tacotron2Loss(torch.nn.Module):
    """``criterion(model_output, (mel_target, gate_target)) -> scalar``."""

    def forward(self, model_output, targets):
        mel_target, gate_target = targets
        mel_target = mel_target.detach()
        mel_dec, mel_post, gate_logits = model_output[0], model_output[1], model_output[2]
        if (mel_dec.is_cuda and not nv.validate_only() and mel_dec.dtype == torch.float32
                and mel_post.dtype == torch.float32 and gate_logits.dtype == torch.float32):
            return _FusedLoss.apply(mel_dec, mel_post, gate_logits, mel_target.float(), gate_target.detach().float())
        gate_target = gate_target.detach().reshape(-1, 1)
        loss_dec = F.mse_loss(mel_dec, mel_target, reduction='mean')
        loss_post = F.mse_loss(mel_post, mel_target, reduction='mean')
        loss_gate = F.binary_cross_entropy_with_logits(
            gate_logits.reshape(-1, 1), gate_target, reduction='mean')
        return loss_dec + loss_post + loss_gate
