"""Shared helpers for the golden fixtures (no dependency on /root/reference).

Weights are never stored in fixtures: they are regenerated from a seed through
``tacotron2_amd.model.Tacotron2`` (whose initialisation is pinned bit-for-bit against the
reference by ``tests/golden/make_golden.py``); fixtures hold inputs, packed dropout masks and
the reference's outputs / gradient digests.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tacotron2_amd.hparams import create_hparams  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

TINY_HP = ("encoder_embedding_dim=128,symbols_embedding_dim=128,attention_rnn_dim=128,"
           "decoder_rnn_dim=128,prenet_dim=64,postnet_embedding_dim=128")

CASES = OrderedDict([
    ("tiny_train", dict(kind="train", hp=TINY_HP, seed=1234, in_lens=[12, 9, 5], out_lens=[20, 16, 11])),
    ("default_train", dict(kind="train", hp="", seed=1234, in_lens=[17, 11], out_lens=[30, 23])),
    # mask_padding=False (reference model.py:490, hparams.py:85): padded frames keep their decoded values and
    # contribute to the loss and to every gradient
    ("tiny_train_nomask", dict(kind="train", hp=TINY_HP + ",mask_padding=False", seed=77, in_lens=[10, 7, 4],
                               out_lens=[9, 18, 13])),
    ("default_infer", dict(kind="infer", hp="max_decoder_steps=40", seed=1234, in_lens=[13], steps=40)),
    ("tiny_infer_batched", dict(kind="infer_batched", hp=TINY_HP + ",max_decoder_steps=24", seed=4321,
                                in_lens=[11, 8, 4], steps=24)),
])


def make_hparams(overrides):
    return create_hparams(overrides if overrides else None)


def build_state_dict(hp, seed, perturb_bn=False):
    """Reference-identical initial state_dict (CPU fp32) under ``torch.manual_seed(seed)``."""
    from tacotron2_amd.model import Tacotron2
    torch.manual_seed(seed)
    model = Tacotron2(hp)
    sd = OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())
    if perturb_bn:
        g = torch.Generator().manual_seed(seed + 99)
        for k in sd:
            if k.endswith('running_mean'):
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
            elif k.endswith('running_var'):
                sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
            elif k.endswith('.1.weight'):
                sd[k] = 0.8 + 0.4 * torch.rand(sd[k].shape, generator=g)
            elif k.endswith('.1.bias'):
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        # at random init the stop gate is flat (sigmoid ~ 0.503 +- 3e-4); amplify it so that the
        # inference fixtures exercise the stop rule with a real margin
        sd['decoder.gate_layer.linear_layer.weight'] = sd['decoder.gate_layer.linear_layer.weight'] * 60.0
    return sd


def make_train_batch(in_lens, out_lens, n_mel, seed):
    """Seeded batch in the TextMelCollate layout (reference data_utils.py:73-111)."""
    g = torch.Generator().manual_seed(seed + 7)
    B, Ti, To = len(in_lens), max(in_lens), max(out_lens)
    text = torch.zeros(B, Ti, dtype=torch.long)
    mel = torch.zeros(B, n_mel, To)
    gate = torch.zeros(B, To)
    for b in range(B):
        text[b, :in_lens[b]] = torch.randint(1, 148, (in_lens[b],), generator=g)
        mel[b, :, :out_lens[b]] = -5.0 + 2.0 * torch.randn(n_mel, out_lens[b], generator=g)
        gate[b, out_lens[b] - 1:] = 1.0
    return text, torch.tensor(in_lens), mel, gate, torch.tensor(out_lens)


def make_text(in_lens, seed):
    g = torch.Generator().manual_seed(seed + 11)
    B, Ti = len(in_lens), max(in_lens)
    text = torch.zeros(B, Ti, dtype=torch.long)
    for b in range(B):
        text[b, :in_lens[b]] = torch.randint(1, 148, (in_lens[b],), generator=g)
    return text


# ---- mask packing ----------------------------------------------------------------------------
def pack_mask(m):
    a = m.to(torch.uint8).numpy()
    return dict(shape=list(a.shape), bits=torch.from_numpy(np.packbits(a.reshape(-1))))


def unpack_mask(d):
    n = int(np.prod(d['shape']))
    a = np.unpackbits(d['bits'].numpy())[:n].reshape(d['shape'])
    return torch.from_numpy(a.astype(np.uint8))


def pack_masks(masks):
    out = {}
    for k, v in masks.items():
        out[k] = [pack_mask(x) for x in v] if isinstance(v, list) else pack_mask(v)
    return out


def unpack_masks(packed):
    out = {}
    for k, v in packed.items():
        out[k] = [unpack_mask(x) for x in v] if isinstance(v, list) else unpack_mask(v)
    return out


def grad_digest(t):
    f = t.detach().double().reshape(-1)
    n = f.numel()
    idx = torch.linspace(0, n - 1, steps=min(n, 64)).long()
    return dict(sum=f.sum().item(), abssum=f.abs().sum().item(), l2=f.norm().item(),
                idx=idx, sample=f[idx].float())


def masks_to_engine(masks, device):
    """Oracle/reference-layout masks -> engine layout (channel-last), on ``device``."""
    out = {}
    if 'enc' in masks:
        out['enc'] = [m.permute(0, 2, 1).contiguous().to(device) for m in masks['enc']]
    if 'prenet' in masks:
        out['prenet'] = [m[:-1].contiguous().to(device) for m in masks['prenet']]   # drop the unused last frame
    if 'att' in masks:
        out['att'] = masks['att'].contiguous().to(device)
        out['dec'] = masks['dec'].contiguous().to(device)
    if 'post' in masks:
        out['post'] = [m.permute(0, 2, 1).contiguous().to(device) for m in masks['post']]
    if 'prenet_infer' in masks:
        out['prenet_infer'] = masks['prenet_infer'].contiguous().to(device)
    return out


def load_fixture(name):
    path = os.path.join(GOLDEN_DIR, name + ".pt")
    return torch.load(path, map_location='cpu', weights_only=False)
