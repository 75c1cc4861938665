"""Small host helpers with the reference's names (reference utils.py:6-10, 24-29)."""
import torch


def get_mask_from_lengths(lengths):
    """(B,) lengths -> (B, max_len) bool, True on valid positions.

    Same result as reference utils.py:6-10 but device-agnostic (the reference
    hard-codes ``torch.cuda.LongTensor``)."""
    max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, dtype=torch.long, device=lengths.device)
    return ids < lengths.unsqueeze(1)


def to_gpu(x):
    """reference utils.py:24-29."""
    x = x.contiguous()
    if torch.cuda.is_available():
        x = x.cuda(non_blocking=True)
    return x


def load_filepaths_and_text(filename, split="|"):
    """One list of fields per filelist line, e.g. ``["DUMMY/LJ001-0001.wav", "transcript"]``
    (reference utils.py:18-21; the reference's filelists are pipe-separated)."""
    rows = []
    with open(filename, encoding='utf-8') as fh:
        for line in fh:
            rows.append(line.strip().split(split))
    return rows


def load_wav_to_torch(full_path):
    """``(samples as float32 tensor in the file's integer range, sampling_rate)`` — reference
    utils.py:13-15 (scipy.io.wavfile.read, no resampling, no scaling)."""
    import numpy as np
    from scipy.io.wavfile import read
    sampling_rate, data = read(full_path)
    return torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)), sampling_rate
