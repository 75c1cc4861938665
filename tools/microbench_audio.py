"""Mel front end throughput (python tools/microbench_audio.py, on the GPU box): frames/s of
TacotronSTFT.mel_spectrogram on LJSpeech-length clips, against the f32-MFMA bound of DESIGN.md §8.1
(2.19 MFLOP and ~13.5 KB per frame).  Prints one JSON line; run it under rocprofv3 --kernel-trace --stats for the
per-kernel split (gemm_f32_kernel x2, reflect_pad / magnitude / mel_log kernels)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacotron2_amd import native
from tacotron2_amd.audio import TacotronSTFT

native.load()
stft = TacotronSTFT().cuda()
out = {}
for name, B, seconds in (("one_clip_6.6s", 1, 6.6), ("batch64_6.6s", 64, 6.6), ("batch64_10s", 64, 10.0)):
    T = int(seconds * 22050)
    y = (0.3 * torch.randn(B, T, generator=torch.Generator().manual_seed(1))).clamp_(-1, 1).cuda()
    for _ in range(3):
        mel = stft.mel_spectrogram(y, check_range=False)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        mel = stft.mel_spectrogram(y, check_range=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    frames = B * mel.shape[2]
    out[name] = {"B": B, "samples": T, "frames": frames, "ms": dt * 1e3, "frames_per_s": frames / dt,
                 "tflops_f32": frames * 2.19e6 / dt / 1e12, "realtime_factor": B * seconds / dt}
print(json.dumps(out))
