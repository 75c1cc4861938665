"""Fused Tacotron2Loss (csrc/loss.hip) against the reference's three torch calls (loss_function.py:8-19): value and
the three gradients, on a padded LJSpeech-shaped output set (gate = 1e3 on padded frames, as parse_output leaves it)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,To", [(3, 41), (64, 870), (1, 7)])
def test_fused_loss_matches_torch(native_lib, B, To):
    from tacotron2_amd.loss_function import Tacotron2Loss
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 1000 + To)
    mel = torch.randn(B, 80, To, generator=g).to(dev).requires_grad_(True)
    post = torch.randn(B, 80, To, generator=g).to(dev).requires_grad_(True)
    gate = (4 * torch.randn(B, To, generator=g))
    gate[:, To - max(1, To // 5):] = 1e3                            # padded frames after parse_output
    gate = gate.to(dev).requires_grad_(True)
    tgt = (-5 + 2 * torch.randn(B, 80, To, generator=g)).to(dev)
    gtgt = torch.zeros(B, To)
    gtgt[:, To - max(1, To // 5) - 1:] = 1.0
    gtgt = gtgt.to(dev)
    loss = Tacotron2Loss()([mel, post, gate, None], (tgt, gtgt))
    (3.0 * loss).backward()                                          # a non-unit upstream gradient
    m2, p2, g2 = (t.detach().clone().double().requires_grad_(True) for t in (mel, post, gate))
    ref = F.mse_loss(m2, tgt.double()) + F.mse_loss(p2, tgt.double()) + \
        F.binary_cross_entropy_with_logits(g2.reshape(-1, 1), gtgt.double().reshape(-1, 1))
    (3.0 * ref).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-6 * abs(float(ref.detach()))
    for a, b in ((mel.grad, m2.grad), (post.grad, p2.grad), (gate.grad, g2.grad)):
        assert (a.double() - b).abs().max().item() <= 2e-6 * b.abs().max().item() + 1e-12
    # bit-reproducible
    loss2 = Tacotron2Loss()([mel.detach(), post.detach(), gate.detach(), None], (tgt, gtgt))
    assert float(loss2) == float(loss)
