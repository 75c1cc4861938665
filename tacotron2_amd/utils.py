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
