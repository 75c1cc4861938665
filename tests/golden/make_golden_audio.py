"""Golden fixtures for the widened rows: mel front end and batch collation.

    python tests/golden/make_golden_audio.py        # writes tests/golden/audio_demo.pt, collate.pt

Runs only in the build container.  Executes the UNMODIFIED reference files on CPU:
  * /root/reference/stft.py + layers.py (``TacotronSTFT.mel_spectrogram``) on two slices of the
    reference's own demo.wav.  librosa is absent, so ``librosa.util.pad_center`` and
    ``librosa.filters.mel`` are supplied by oracle/audio_oracle.py's restatement of librosa 0.6.0
    (the filterbank table is therefore NOT pinned by a reference artefact; everything else is);
  * /root/reference/data_utils.py ``TextMelCollate`` on seeded ragged items, and the filelist
    shuffle order of ``TextMelLoader`` (``random.seed(seed); random.shuffle``).
Before writing, asserts that oracle/audio_oracle.py reproduces the reference's mel output and that
tacotron2_amd's Fourier basis / mel filterbank / collate / shuffle equal the reference's.
"""
import os
import random
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import audio_oracle as ao  # noqa: E402

REF = "/root/reference"


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub('librosa')
    stub('librosa.filters', mel=ao.librosa_mel)
    stub('librosa.util', pad_center=ao.pad_center, tiny=lambda x: np.finfo(np.float32).tiny, normalize=None)
    sys.modules['librosa'].filters = sys.modules['librosa.filters']
    sys.modules['librosa'].util = sys.modules['librosa.util']
    stub('unidecode', unidecode=lambda s: s)
    stub('inflect', engine=lambda: types.SimpleNamespace(number_to_words=lambda *a, **k: 'num'))
    sys.path.insert(0, REF)
    import layers as ref_layers
    import data_utils as ref_data
    sys.path.remove(REF)
    return ref_layers, ref_data


def main():
    ref_layers, ref_data = import_reference()
    from scipy.io.wavfile import read
    from tacotron2_amd import audio
    from tacotron2_amd.data_utils import TextMelCollate

    # ---- mel front end -------------------------------------------------------------------
    sr, wav = read(os.path.join(REF, "demo.wav"))
    assert sr == 22050 and wav.dtype == np.float32
    y = torch.zeros(2, 9000)
    y[0] = torch.from_numpy(wav[20000:29000].copy())
    y[1] = torch.from_numpy(wav[50000:59000].copy())
    ref_stft = ref_layers.TacotronSTFT()                       # defaults = hparams.py:35-42
    mel_ref = ref_stft.mel_spectrogram(y)
    mag_ref, _ = ref_stft.stft_fn.transform(y)
    mel_orc = ao.mel_spectrogram(y)
    mag_orc = ao.stft_magnitude(y)
    assert torch.equal(mag_orc, mag_ref.data), (mag_orc - mag_ref).abs().max()
    assert torch.equal(mel_orc, mel_ref), (mel_orc - mel_ref).abs().max()
    # the product's tables against the reference's buffers
    fb_ref = ref_stft.stft_fn.forward_basis[:, 0, :].numpy()
    fb_new = audio.fourier_basis(1024, 1024)
    d = np.abs(fb_ref - fb_new).max()
    assert d <= 2 ** -23, d                                    # cos/sin vs fft(eye): <= 1 ulp near 1.0
    mb_new = audio.mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    assert np.abs(mb_new - ao.librosa_mel(22050, 1024, 80, 0.0, 8000.0)).max() < 1e-15
    # odd length (T % hop != 0), mono
    y1 = torch.from_numpy(wav[70000:70000 + 4321].copy()).unsqueeze(0)
    mel1 = ref_stft.mel_spectrogram(y1)
    assert torch.equal(ao.mel_spectrogram(y1), mel1)
    torch.save({"y": y, "mel": mel_ref.clone(), "mag_row_sums": mag_ref.data.sum(dim=1).clone(),
                "y_odd": y1, "mel_odd": mel1.clone(),
                "basis_digest": torch.from_numpy(fb_ref).double().sum(dim=1)},
               os.path.join(HERE, "audio_demo.pt"))
    print("audio_demo.pt: mel", tuple(mel_ref.shape), "basis max diff", d)

    # ---- collate + shuffle ---------------------------------------------------------------
    g = torch.Generator().manual_seed(99)
    items = []
    for Ti, To in [(7, 19), (12, 31), (7, 25), (3, 8), (12, 30)]:          # ties in the text length on purpose
        items.append((torch.randint(1, 148, (Ti,), generator=g, dtype=torch.int32), torch.randn(80, To, generator=g)))
    cases = {}
    for r in (1, 2, 3):
        ref_out = ref_data.TextMelCollate(r)(items)
        new_out = TextMelCollate(r)(items)
        for a, b in zip(ref_out, new_out):
            assert a.dtype == b.dtype and torch.equal(a, b), r
        cases[r] = [t.clone() for t in ref_out]
    lines = [["f%d.wav" % i, "t%d" % i] for i in range(50)]
    ref_order = list(lines)
    random.seed(1234)
    random.shuffle(ref_order)
    torch.save({"items": items, "collated": cases, "shuffle_1234": [int(r[0][1:-4]) for r in ref_order]},
               os.path.join(HERE, "collate.pt"))
    print("collate.pt written")


if __name__ == "__main__":
    main()
