"""CPU oracle of the mel front end — TEST INFRASTRUCTURE ONLY.

Only ``tests/`` and ``tests/golden/make_golden_audio.py`` import this file; the product
(``tacotron2_amd/audio.py`` + csrc/audio.hip) never does.

Restates, with torch CPU float32 ops in the reference's own order:
  * ``STFT.__init__`` / ``STFT.transform``    reference stft.py:44-105 (Fourier basis from
    ``np.fft.fft(np.eye(L))``, periodic hann window centre-padded to L, reflect pad L/2,
    ``conv1d`` with stride hop, ``sqrt(re**2 + im**2)``);
  * ``TacotronSTFT.mel_spectrogram``          reference layers.py:63-80 (``mel_basis @ magnitudes``,
    ``log(clamp(x, min=1e-5))``, audio_processing.py:78-84);
  * ``librosa.filters.mel`` / ``librosa.util.pad_center``  — third-party, NOT in the reference tree
    (reference requirements.txt pins librosa==0.6.0) and not installed here: the published 0.6.0
    algorithm (Slaney mel scale, ``norm=1`` area normalisation) is restated below with scalar loops.

Pinning: ``tests/golden/make_golden_audio.py`` runs the reference's unmodified stft.py / layers.py on
CPU (with ``pad_center`` / ``filters.mel`` supplied by this file, because librosa is absent) on a
slice of the reference's demo.wav and asserts this oracle reproduces it; the fixture
``tests/golden/audio_demo.pt`` stores input and output.  The mel *filterbank* itself has no
artefact of the REFERENCE to be checked against (librosa is neither vendored nor installed); since
round 4 the table is pinned against an independent public implementation that is in this image --
``transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")``, documented as
librosa-equivalent: equal to 2e-16 on the reference's configuration
(tests/test_widening_cpu.py::test_mel_filterbank_against_an_independent_public_implementation).
What stays unpinned is only "librosa 0.6.0 itself produced these numbers".
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ---- librosa 0.6.0 restatements (third-party; see header) -----------------------------------
def hz_to_mel(f):
    f_sp = 200.0 / 3
    if f >= 1000.0:
        return 1000.0 / f_sp + math.log(f / 1000.0) / (math.log(6.4) / 27.0)
    return f / f_sp


def mel_to_hz(m):
    f_sp = 200.0 / 3
    min_log_mel = 1000.0 / f_sp
    if m >= min_log_mel:
        return 1000.0 * math.exp((math.log(6.4) / 27.0) * (m - min_log_mel))
    return f_sp * m


def librosa_mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1), scalar loops."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    fftfreqs = [i * (float(sr) / 2) / (n_bins - 1) for i in range(n_bins)]
    lo, hi = hz_to_mel(fmin), hz_to_mel(fmax)
    mel_f = [mel_to_hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
    weights = np.zeros((n_mels, n_bins), dtype=np.float64)
    for i in range(n_mels):
        for j in range(n_bins):
            lower = (fftfreqs[j] - mel_f[i]) / (mel_f[i + 1] - mel_f[i])
            upper = (mel_f[i + 2] - fftfreqs[j]) / (mel_f[i + 2] - mel_f[i + 1])
            weights[i, j] = max(0.0, min(lower, upper))
        weights[i] *= 2.0 / (mel_f[i + 2] - mel_f[i])
    return weights


def pad_center(data, size):
    """librosa.util.pad_center for 1-D input: zero-pad to ``size`` with the data centred."""
    n = data.shape[-1]
    lpad = int((size - n) // 2)
    return np.pad(data, (lpad, int(size - n - lpad)), mode='constant')


# ---- reference stft.py:44-105 ------------------------------------------------------------------
def forward_basis(filter_length, win_length):
    from scipy.signal import get_window
    fb = np.fft.fft(np.eye(filter_length))
    cutoff = int(filter_length / 2 + 1)
    fb = np.vstack([np.real(fb[:cutoff, :]), np.imag(fb[:cutoff, :])])
    basis = torch.FloatTensor(fb[:, None, :])
    win = torch.from_numpy(pad_center(get_window('hann', win_length, fftbins=True), filter_length)).float()
    basis *= win
    return basis.float()


def stft_magnitude(y, filter_length=1024, hop_length=256, win_length=1024):
    """y (B, T) float32 -> (B, L/2+1, T//hop + 1)."""
    B, T = y.shape
    x = F.pad(y.view(B, 1, 1, T), (filter_length // 2, filter_length // 2, 0, 0), mode='reflect').squeeze(1)
    ft = F.conv1d(x, forward_basis(filter_length, win_length), stride=hop_length, padding=0)
    cutoff = filter_length // 2 + 1
    re, im = ft[:, :cutoff, :], ft[:, cutoff:, :]
    return torch.sqrt(re ** 2 + im ** 2)


def mel_spectrogram(y, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80,
                    sampling_rate=22050, mel_fmin=0.0, mel_fmax=8000.0):
    """reference layers.py:63-80."""
    assert float(y.min()) >= -1 and float(y.max()) <= 1
    mag = stft_magnitude(y, filter_length, hop_length, win_length)
    basis = torch.from_numpy(librosa_mel(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)).float()
    mel = torch.matmul(basis, mag)
    return torch.log(torch.clamp(mel, min=1e-5))
