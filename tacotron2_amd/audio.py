"""wav -> log-mel on the MI355X.

Interface of the reference's ``layers.TacotronSTFT`` (reference layers.py:42-80) and the forward
half of ``stft.STFT`` (reference stft.py:42-105): ``TacotronSTFT(filter_length, hop_length,
win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax).mel_spectrogram(y)`` with ``y``
(B, T) in [-1, 1] returns (B, n_mel_channels, T // hop + 1) log-mels, buffers ``mel_basis`` and
``stft_fn.forward_basis`` as in the reference.

Arithmetic (same as the reference, different schedule):
  1. reflect-pad by filter_length/2 on both sides                       (csrc/audio.hip)
  2. spec = frames . forward_basis^T on the exact-f32 MFMA GEMM; the frame matrix is never built:
     the padded signal is the A operand with row stride ``hop`` (< K, rows overlap)   (gemm.hip)
  3. magnitude sqrt(re^2 + im^2)                                        (csrc/audio.hip)
  4. mel = mag . mel_basis^T                                            (gemm.hip)
  5. log(clamp(mel, 1e-5)) and the transpose to (B, n_mel, frames)      (csrc/audio.hip)
No torch arithmetic touches the samples; torch allocates the buffers.  There is no CPU path.

The mel filterbank is librosa 0.6.0's ``filters.mel(sr, n_fft, n_mels, fmin, fmax)`` (htk=False,
norm=1: Slaney's Auditory-Toolbox scale, area-normalised triangles), which the reference calls at
layers.py:50-51; librosa is not vendored in the reference and not installed here, so the published
algorithm is restated in ``mel_filterbank`` (parity for this table is unpinned by any reference
artefact; the STFT/magnitude/log part is pinned against the reference's own stft.py run on CPU,
tests/golden/make_golden_audio.py).  The inverse STFT / Griffin-Lim (stft.py:107-141) is only
used by the notebook's vocoder hand-off and is out of scope.
"""
import os

import numpy as np
import torch

from . import native as nv

_F_SP = 200.0 / 3.0                  # Slaney: linear below 1 kHz, 200/3 Hz per mel
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP   # = 15
_LOGSTEP = np.log(6.4) / 27.0        # log-spaced above: 27 mels per factor 6.4


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / _F_SP
    log = _MIN_LOG_MEL + np.log(np.maximum(f, _MIN_LOG_HZ) / _MIN_LOG_HZ) / _LOGSTEP
    return np.where(f >= _MIN_LOG_HZ, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = _F_SP * m
    log = _MIN_LOG_HZ * np.exp(_LOGSTEP * (np.maximum(m, _MIN_LOG_MEL) - _MIN_LOG_MEL))
    return np.where(m >= _MIN_LOG_MEL, log, lin)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """(n_mels, 1 + n_fft//2) float64 triangular filters, Slaney scale, each scaled by
    2 / (its band width in Hz) — librosa 0.6.0 ``filters.mel`` with its defaults."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    bin_hz = np.linspace(0.0, sr / 2.0, n_bins)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    dist = edges[:, None] - bin_hz[None, :]                 # (n_mels+2, n_bins)
    rising = -dist[:-2] / width[:-1, None]
    falling = dist[2:] / width[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling))
    return tri * (2.0 / (edges[2:] - edges[:-2]))[:, None]


def fourier_basis(filter_length, win_length, window='hann'):
    """(2F, filter_length) float32: rows [cos | -sin](2 pi f k / L), F = L/2 + 1, each multiplied in float32
    by the periodic window zero-padded (centred) to filter_length — what stft.py:53-70 registers."""
    L = int(filter_length)
    if win_length > L:
        raise AssertionError("filter_length must be >= win_length")
    F = L // 2 + 1
    k = np.arange(L, dtype=np.float64)
    ang = 2.0 * np.pi * np.outer(np.arange(F, dtype=np.float64), k) / L
    basis = np.vstack([np.cos(ang), -np.sin(ang)]).astype(np.float32)
    if window is not None:
        if window != 'hann':
            raise ValueError("only the periodic hann window of the reference is built in")
        n = np.arange(win_length, dtype=np.float64)
        w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
        lpad = (L - win_length) // 2
        wfull = np.zeros(L, dtype=np.float64)
        wfull[lpad:lpad + win_length] = w
        basis = basis * wfull.astype(np.float32)[None, :]
    return basis


class STFT(torch.nn.Module):
    """Forward transform (magnitude) of reference stft.py:42-105 on the GPU."""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window='hann'):
        super().__init__()
        self.filter_length, self.hop_length, self.win_length, self.window = filter_length, hop_length, win_length, window
        self.cutoff = filter_length // 2 + 1
        basis = torch.from_numpy(fourier_basis(filter_length, win_length, window))
        self.register_buffer('forward_basis', basis[:, None, :].contiguous())      # (2F, 1, L) like the reference

    def magnitude_rows(self, y):
        """y (B, T) device f32 -> (mag (B*n, Fpad) with zero columns beyond F, n)."""
        B, T = y.shape
        L, hop, F = self.filter_length, self.hop_length, self.cutoff
        if T <= L // 2:
            raise ValueError("signal of %d samples is too short to reflect-pad by %d" % (T, L // 2))
        n = T // hop + 1
        ldo = (T + L + 3) // 4 * 4
        padded = torch.empty(B, ldo, dtype=torch.float32, device=y.device)
        nv.reflect_pad(y, padded, L // 2)
        spec = torch.empty(B * n, 2 * F, dtype=torch.float32, device=y.device)
        frames0 = padded.as_strided((n, L), (hop, 1))                              # utterance 0; rows overlap
        nv.gemm(spec[:n], frames0, self.forward_basis.view(2 * F, L), batch=B, strides=(ldo, 0, n * 2 * F))
        Fpad = (F + 15) // 16 * 16
        mag = torch.empty(B * n, Fpad, dtype=torch.float32, device=y.device)
        nv.stft_magnitude(spec, mag, F)
        return mag, n

    def transform_magnitude(self, y):
        """(B, F, n) magnitudes, the first return value of the reference's ``transform``."""
        y = _device_signal(y, self.forward_basis)
        mag, n = self.magnitude_rows(y)
        out = torch.empty(y.shape[0], self.cutoff, n, dtype=torch.float32, device=y.device)
        nv.transpose(out.view(-1, n)[:self.cutoff], mag[:n, :self.cutoff], batch=y.shape[0],
                     sstride=n * mag.shape[1], dstride=self.cutoff * n)
        return out


def _device_signal(y, like):
    if y.dim() != 2:
        raise ValueError("expected (B, T) samples, got shape %s" % (tuple(y.shape),))
    if not like.is_cuda and not nv.validate_only():
        raise nv.NativeError("tacotron2_amd.audio: move the module to the MI355X first (.cuda()); there is no CPU path")
    return y.detach().to(device=like.device, dtype=torch.float32).contiguous()


class TacotronSTFT(torch.nn.Module):
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80,
                 sampling_rate=22050, mel_fmin=0.0, mel_fmax=8000.0):
        super().__init__()
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        self.stft_fn = STFT(filter_length, hop_length, win_length)
        fb = mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)
        self.register_buffer('mel_basis', torch.from_numpy(fb).float())
        F = self.stft_fn.cutoff
        Fpad = (F + 15) // 16 * 16
        padded = torch.zeros(n_mel_channels, Fpad, dtype=torch.float32)
        padded[:, :F] = self.mel_basis
        self.register_buffer('_mel_basis_padded', padded, persistent=False)
        self.clip_val = 1e-5
        if torch.cuda.is_available():
            self.cuda()

    def spectral_normalize(self, magnitudes):
        """log(clamp(x, 1e-5)) (reference audio_processing.py:78-84) — elementwise convenience for callers
        outside the mel path; ``mel_spectrogram`` fuses it into its last kernel."""
        return torch.log(torch.clamp(magnitudes, min=self.clip_val))

    def spectral_de_normalize(self, magnitudes):
        return torch.exp(magnitudes)

    def mel_spectrogram(self, y, check_range=True):
        """y (B, T) float in [-1, 1] -> (B, n_mel_channels, T // hop + 1) on the GPU."""
        y = _device_signal(y, self.mel_basis)
        if check_range and not nv.validate_only():
            lo, hi = (float(v) for v in torch.stack(torch.aminmax(y)).tolist())
            if lo < -1.0 or hi > 1.0:
                raise AssertionError("samples must lie in [-1, 1] (got [%g, %g]); divide by max_wav_value" % (lo, hi))
        mag, n = self.stft_fn.magnitude_rows(y)
        B = y.shape[0]
        mel_rows = torch.empty(B * n, self.n_mel_channels, dtype=torch.float32, device=y.device)
        nv.gemm(mel_rows, mag, self._mel_basis_padded)
        out = torch.empty(B, self.n_mel_channels, n, dtype=torch.float32, device=y.device)
        nv.mel_log_compress(mel_rows, out, self.clip_val)
        return out


def precompute_mels(filelist, hparams, out_dir, out_filelist=None):
    """Run every wav of a ``path|text`` filelist through the GPU front end once and write
    ``<out_dir>/<stem>.npy`` (n_mel, frames) float32 — the files the reference's
    ``load_mel_from_disk=True`` path reads (hparams.py:27, data_utils.py:50-55).  Optionally writes
    the matching filelist.  Returns the number of utterances."""
    from .utils import load_filepaths_and_text, load_wav_to_torch
    stft = TacotronSTFT(hparams.filter_length, hparams.hop_length, hparams.win_length, hparams.n_mel_channels,
                        hparams.sampling_rate, hparams.mel_fmin, hparams.mel_fmax)
    os.makedirs(out_dir, exist_ok=True)
    rows = load_filepaths_and_text(filelist)
    lines = []
    for fields in rows:
        audio, sr = load_wav_to_torch(fields[0])
        if sr != hparams.sampling_rate:
            raise ValueError("%s: sampling rate %d, expected %d" % (fields[0], sr, hparams.sampling_rate))
        mel = stft.mel_spectrogram((audio / hparams.max_wav_value).unsqueeze(0)).squeeze(0)
        dst = os.path.join(out_dir, os.path.splitext(os.path.basename(fields[0]))[0] + '.npy')
        np.save(dst, mel.cpu().numpy())
        lines.append('|'.join([dst] + fields[1:]))
    if out_filelist:
        with open(out_filelist, 'w', encoding='utf-8') as fh:
            fh.write('\n'.join(lines) + '\n')
    return len(rows)
