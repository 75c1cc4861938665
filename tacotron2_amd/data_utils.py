"""Host-side data path: (text ids, mel) pairs and the padded batch layout the engine consumes.

Mirrors the interface of the reference's ``TextMelLoader`` / ``TextMelCollate``
(reference data_utils.py:11-111): a filelist of ``path|transcript`` lines, shuffled once with
``hparams.seed``; items are ``(IntTensor (Ti,), FloatTensor (n_mel, To))``; the collate sorts by
text length (descending), right-pads with zeros and emits the 5-tuple
``(text_padded i64 (B,Ti), input_lengths i64 (B), mel_padded f32 (B,n_mel,To), gate_padded f32
(B,To), output_lengths i64 (B))`` that ``Tacotron2.parse_batch`` takes.

What is different (MI355X-first):
  * wav -> mel runs on the GPU (``tacotron2_amd.audio.TacotronSTFT``: STFT as an MFMA GEMM against
    the windowed Fourier basis, fused magnitude / mel / log-compression kernels) instead of a
    ``num_workers=1`` CPU conv1d (reference train.py:55) that would starve an engine several
    hundred times faster than the CPU model; ``audio.precompute_mels`` writes the ``.npy`` files
    the reference's own ``load_mel_from_disk`` path (hparams.py:27) reads.
  * the text frontend (cleaners, number expansion, ARPAbet; reference text/*.py) is out of scope
    (SURVEY.md §2): pass ``text_to_sequence=`` (e.g. the reference's ``text.text_to_sequence``),
    or ship pre-tokenised filelists whose transcript field is space-separated symbol ids.
  * ``synthetic:N[:seed[:max_frames]]`` as the filelist name yields N LJSpeech-shaped utterances
    (``tacotron2_amd.synth``; ``max_frames`` caps their length): the datasets are not on the box,
    the loop still has to run.
"""
import random

import numpy as np
import torch
import torch.utils.data

from .utils import load_filepaths_and_text, load_wav_to_torch


PRETOKENISED = 'ids:'        # filelist marker: "path|ids: 12 7 33 ..." is a transcript that already is symbol ids


def _ids_from_transcript(text, cleaner_names):
    """Default tokeniser.  A transcript that starts with the marker ``ids:`` is a list of symbol ids (explicit
    opt-in: a plain transcript such as "1984" is text, not id 1984); anything else goes to the reference's
    ``text.text_to_sequence`` when that package is importable."""
    if text.lstrip().startswith(PRETOKENISED):
        fields = text.lstrip()[len(PRETOKENISED):].split()
        if not fields or not all(f.isdigit() for f in fields):
            raise ValueError("pre-tokenised transcript must be 'ids:' followed by non-negative integers, got %r" % (text,))
        return [int(f) for f in fields]
    try:                                                   # the reference's package, if importable
        from text import text_to_sequence                 # noqa: WPS433
    except Exception as e:
        raise RuntimeError(
            "TextMelLoader: no text frontend.  The cleaners/symbol table (reference text/*.py) are out "
            "of scope for the engine: pass text_to_sequence=<callable(text, cleaner_names) -> ids>, put the "
            "reference's `text` package on sys.path, or use a pre-tokenised filelist ('path|ids: 12 7 33').  "
            "(import text failed: %s)" % (e,))
    return text_to_sequence(text, cleaner_names)


class TextMelLoader(torch.utils.data.Dataset):
    """``TextMelLoader(filelist, hparams)[i] -> (text_ids IntTensor, mel FloatTensor (n_mel, To))``."""

    def __init__(self, audiopaths_and_text, hparams, text_to_sequence=None):
        self.text_cleaners = hparams.text_cleaners
        self.max_wav_value = hparams.max_wav_value
        self.sampling_rate = hparams.sampling_rate
        self.load_mel_from_disk = hparams.load_mel_from_disk
        self.n_mel_channels = hparams.n_mel_channels
        self.n_symbols = hparams.n_symbols
        self._hparams = hparams
        self._stft = None
        self._tokenise = text_to_sequence or _ids_from_transcript
        self.synthetic = None
        if isinstance(audiopaths_and_text, str) and audiopaths_and_text.startswith('synthetic:'):
            parts = audiopaths_and_text.split(':')
            n = int(parts[1])
            seed = int(parts[2]) if len(parts) > 2 else hparams.seed
            from .synth import synth_lengths
            ti, to = synth_lengths(n, seed)
            if len(parts) > 3:                              # synthetic:N:seed:max_frames -> short utterances
                cap = int(parts[3])
                k = np.arange(n)                                # keep the capped set ragged
                to = np.maximum(2, np.minimum(to, cap) - k % 7)
                ti = np.maximum(2, np.minimum(ti, max(2, cap // 5)) - k % 3)
            self.synthetic = (seed, ti, to)
            self.audiopaths_and_text = [['synthetic/%d' % i, ''] for i in range(n)]
        else:
            self.audiopaths_and_text = load_filepaths_and_text(audiopaths_and_text)
        # reference data_utils.py:28-29 seeds the global RNG and shuffles; a private Mersenne
        # Twister with the same seed yields the same permutation without touching global state
        random.Random(hparams.seed).shuffle(self.audiopaths_and_text)

    # -- mel ----------------------------------------------------------------------------------
    @property
    def stft(self):
        if self._stft is None:
            from .audio import TacotronSTFT
            hp = self._hparams
            self._stft = TacotronSTFT(hp.filter_length, hp.hop_length, hp.win_length, hp.n_mel_channels,
                                      hp.sampling_rate, hp.mel_fmin, hp.mel_fmax)
        return self._stft

    def get_mel(self, filename):
        if self.load_mel_from_disk:
            melspec = torch.from_numpy(np.load(filename))
            if melspec.size(0) != self.n_mel_channels:
                raise AssertionError('Mel dimension mismatch: given {}, expected {}'.format(
                    melspec.size(0), self.n_mel_channels))
            return melspec
        audio, sampling_rate = load_wav_to_torch(filename)
        if sampling_rate != self.sampling_rate:
            raise ValueError("{}: {} SR doesn't match target {} SR".format(
                filename, sampling_rate, self.sampling_rate))
        audio_norm = (audio / self.max_wav_value).unsqueeze(0)
        return self.stft.mel_spectrogram(audio_norm).squeeze(0).cpu()

    # -- text ---------------------------------------------------------------------------------
    def get_text(self, text):
        ids = self._tokenise(text, self.text_cleaners)
        if ids and not (0 <= min(ids) and max(ids) < self.n_symbols):
            raise ValueError("symbol id out of range for the embedding (n_symbols=%d): %r" % (self.n_symbols, text[:60]))
        return torch.IntTensor(ids)

    def get_mel_text_pair(self, audiopath_and_text):
        return self.get_text(audiopath_and_text[1]), self.get_mel(audiopath_and_text[0])

    def _synthetic_item(self, index):
        seed, ti, to = self.synthetic
        i = int(self.audiopaths_and_text[index][0].split('/')[1])
        g = torch.Generator().manual_seed(seed * 1000003 + i)
        text = torch.randint(1, self.n_symbols, (int(ti[i]),), generator=g, dtype=torch.int32)
        mel = -5.0 + 2.0 * torch.randn(self.n_mel_channels, int(to[i]), generator=g)
        return text, mel

    def __getitem__(self, index):
        if self.synthetic is not None:
            return self._synthetic_item(index)
        return self.get_mel_text_pair(self.audiopaths_and_text[index])

    def __len__(self):
        return len(self.audiopaths_and_text)


class TextMelCollate(object):
    """Zero-pads a list of ``(text_ids, mel)`` items into the engine's batch layout; the frame
    count is rounded up to a multiple of ``n_frames_per_step`` (reference data_utils.py:67-111)."""

    def __init__(self, n_frames_per_step):
        self.n_frames_per_step = n_frames_per_step

    def __call__(self, batch):
        n = len(batch)
        input_lengths, order = torch.sort(torch.LongTensor([item[0].numel() for item in batch]),
                                          dim=0, descending=True)
        order = order.tolist()
        frames = [batch[j][1].size(1) for j in order]
        n_mel = batch[0][1].size(0)
        To = max(frames)
        r = self.n_frames_per_step
        To = (To + r - 1) // r * r
        text_padded = torch.zeros(n, int(input_lengths[0]), dtype=torch.long)
        mel_padded = torch.zeros(n, n_mel, To, dtype=torch.float32)
        gate_padded = torch.zeros(n, To, dtype=torch.float32)
        for row, j in enumerate(order):
            text, mel = batch[j]
            text_padded[row, :text.numel()] = text
            mel_padded[row, :, :frames[row]] = mel
            gate_padded[row, frames[row] - 1:] = 1.0        # stop target from the last valid frame on
        output_lengths = torch.LongTensor(frames)
        return text_padded, input_lengths, mel_padded, gate_padded, output_lengths
