"""Synthetic LJSpeech-shaped batches (SURVEY.md §8d / BASELINE.md §3).

Text length  Ti ~ clip(round(N(100, 32)), 15, 187)   (LJSpeech transcripts: median 102, max 187)
Mel frames   To ~ clip(round(5.66*Ti + N(0, 25)), 90, 870)   (1.1-10.1 s at 22,050 Hz / hop 256)
Utterances sorted by text length, descending, zero padded, gate target 1 from the last frame on:
the layout ``TextMelCollate`` produces (reference data_utils.py:73-111).
"""
import numpy as np
import torch


def synth_lengths(batch_size, seed):
    rs = np.random.RandomState(seed)
    ti = np.clip(np.round(rs.normal(100.0, 32.0, batch_size)), 15, 187).astype(np.int64)
    to = np.clip(np.round(5.66 * ti + rs.normal(0.0, 25.0, batch_size)), 90, 870).astype(np.int64)
    order = np.argsort(-ti, kind='stable')
    return ti[order], to[order]


def synth_batch(batch_size, seed, n_mel=80, n_symbols=148):
    """Returns the 5-tuple (text_padded, input_lengths, mel_padded, gate_padded, output_lengths), CPU."""
    ti, to = synth_lengths(batch_size, seed)
    g = torch.Generator().manual_seed(int(seed))
    Ti, To = int(ti.max()), int(to.max())
    text = torch.zeros(batch_size, Ti, dtype=torch.long)
    mel = torch.zeros(batch_size, n_mel, To)
    gate = torch.zeros(batch_size, To)
    for b in range(batch_size):
        text[b, :ti[b]] = torch.randint(1, n_symbols, (int(ti[b]),), generator=g)
        mel[b, :, :to[b]] = -5.0 + 2.0 * torch.randn(n_mel, int(to[b]), generator=g)
        gate[b, to[b] - 1:] = 1.0
    return text, torch.from_numpy(ti.copy()), mel, gate, torch.from_numpy(to.copy())
