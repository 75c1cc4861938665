"""Tacotron2Loss for the MI355X engine.

Contract taken from reference loss_function.py:8-19: the scalar is
``mean((mel - y)^2) + mean((mel_postnet - y)^2) + mean(bce_with_logits(gate, g))``
with every mean taken over the *padded* tensors (padded frames
contribute exactly zero because ``Tacotron2.parse_output`` forced them to the
padding targets).  The criterion sits outside the model boundary (the training
loop applies it to the model's outputs); it is device-side tensor arithmetic,
fused with clipping and Adam only in a later round.
"""
import torch
import torch.nn.functional as F


class Tacotron2Loss(torch.nn.Module):
    """``criterion(model_output, (mel_target, gate_target)) -> scalar``."""

    def forward(self, model_output, targets):
        mel_target, gate_target = targets
        mel_target = mel_target.detach()
        gate_target = gate_target.detach().reshape(-1, 1)
        mel_dec, mel_post, gate_logits = model_output[0], model_output[1], model_output[2]
        loss_dec = F.mse_loss(mel_dec, mel_target, reduction='mean')
        loss_post = F.mse_loss(mel_post, mel_target, reduction='mean')
        loss_gate = F.binary_cross_entropy_with_logits(
            gate_logits.reshape(-1, 1), gate_target, reduction='mean')
        return loss_dec + loss_post + loss_gate
