"""Mel front end on the GPU against the fixture produced by the reference's stft.py /
layers.py and against the CPU oracle."""
import json
import os

import pytest
import torch

import golden_util as gu
from tacotron2_amd.hparams import create_hparams

pytestmark = pytest.mark.gpu

# float32 summation-order noise of this pipeline measured on the fixture (conv1d vs framed matmul on CPU,
# f32 vs f64): max 2e-5, mean 4e-7 in the log-mel domain.  Bounds: 10x that.
LOGMEL_MAX_ABS = 2e-4
LOGMEL_MEAN_ABS = 5e-6


def _golden(name):
    return torch.load(os.path.join(gu.GOLDEN_DIR, name), weights_only=False)


def test_mel_spectrogram_matches_reference_fixture(native_lib):
    from tacotron2_amd.audio import TacotronSTFT
    g = _golden("audio_demo.pt")
    stft = TacotronSTFT().cuda()
    for key_in, key_out in (("y", "mel"), ("y_odd", "mel_odd")):
        out = stft.mel_spectrogram(g[key_in].cuda())
        torch.cuda.synchronize()
        assert out.is_cuda and tuple(out.shape) == tuple(g[key_out].shape)
        d = (out.cpu() - g[key_out]).abs()
        assert d.max().item() < LOGMEL_MAX_ABS and d.mean().item() < LOGMEL_MEAN_ABS, (d.max().item(), d.mean().item())
    # batch of two == each utterance alone, bitwise (rows are independent GEMM problems)
    both = stft.mel_spectrogram(g["y"].cuda())
    for b in range(2):
        one = stft.mel_spectrogram(g["y"][b:b + 1].cuda())
        assert torch.equal(one[0], both[b])
    with pytest.raises(AssertionError):
        stft.mel_spectrogram(2.0 * g["y"].cuda())


def test_mel_spectrogram_matches_oracle_on_seeded_signals(native_lib):
    from oracle import audio_oracle as ao
    from tacotron2_amd.audio import TacotronSTFT
    gen = torch.Generator().manual_seed(5)
    t = torch.arange(22050, dtype=torch.float32) / 22050.0
    chirp = 0.5 * torch.sin(2 * torch.pi * (200.0 + 3000.0 * t) * t)
    noise = 0.1 * torch.randn(3, 22050, generator=gen)
    y = torch.stack([chirp + noise[0], noise[1], torch.clamp(5 * noise[2], -1, 1)])
    for hp in (dict(), dict(filter_length=512, hop_length=128, win_length=400, n_mel_channels=40,
                            sampling_rate=16000, mel_fmin=50.0, mel_fmax=7600.0)):
        ref = ao.mel_spectrogram(y, **hp)
        out = TacotronSTFT(**hp).cuda().mel_spectrogram(y.cuda()).cpu()
        d = (out - ref).abs()
        assert d.max().item() < LOGMEL_MAX_ABS and d.mean().item() < LOGMEL_MEAN_ABS, (hp, d.max().item(), d.mean().item())
    mag = TacotronSTFT().cuda().stft_fn.transform_magnitude(y.cuda()).cpu()
    ref = ao.stft_magnitude(y)
    assert tuple(mag.shape) == tuple(ref.shape)
    assert ((mag - ref).abs() <= 1e-4 + 1e-4 * ref.abs()).all()
    # all-zero signal: every mel bin sits on the clamp, log(1e-5) exactly like the reference's clamp(min=1e-5)
    z = TacotronSTFT().cuda().mel_spectrogram(torch.zeros(1, 4096).cuda()).cpu()
    # the device's logf and the host's differ by up to 2 ulp of |log(1e-5)| = 11.5 (1.9e-6 measured by tools/probe/gpu_selftest)
    assert (z - torch.log(torch.full_like(z, 1e-5))).abs().max().item() < 6e-6


def test_precompute_mels_and_loader_roundtrip(native_lib, tmp_path):
    import numpy as np
    from scipy.io.wavfile import write
    from oracle import audio_oracle as ao
    from tacotron2_amd.audio import precompute_mels
    from tacotron2_amd.data_utils import TextMelLoader
    g = _golden("audio_demo.pt")
    hp = create_hparams()
    pcm = (g["y"][0] * 32767.0).round().to(torch.int16)
    wav = tmp_path / "utt0.wav"
    write(str(wav), 22050, pcm.numpy())
    fl = tmp_path / "list.txt"
    fl.write_text("%s|ids: 5 6 7\n" % wav, encoding="utf-8")
    n = precompute_mels(str(fl), hp, str(tmp_path / "mels"), out_filelist=str(tmp_path / "mels.txt"))
    assert n == 1
    ref = ao.mel_spectrogram((pcm.float() / hp.max_wav_value).unsqueeze(0))[0]
    disk = TextMelLoader(str(tmp_path / "mels.txt"), create_hparams("load_mel_from_disk=True"))
    text, mel = disk[0]
    assert text.tolist() == [5, 6, 7] and (mel - ref).abs().max().item() < LOGMEL_MAX_ABS
    live = TextMelLoader(str(fl), hp)                       # wav -> mel on the GPU inside the dataset
    text2, mel2 = live[0]
    assert not mel2.is_cuda and torch.equal(mel2, mel)
    assert np.load(tmp_path / "mels" / "utt0.npy").shape == (80, 9000 // 256 + 1)
