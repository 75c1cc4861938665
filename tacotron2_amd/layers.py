"""Parameter containers with the reference's initialisation.

``LinearNorm`` / ``ConvNorm`` keep the attribute names (``linear_layer``,
``conv``) and the Xavier-uniform-with-gain initialisation of the reference
(reference layers.py:8-39), so that the ``state_dict`` key set and — under the
same ``torch.manual_seed`` — the initial values are identical.  They are
containers only: the arithmetic of the hot path runs in the HIP engine
(``tacotron2_amd.engine``), never through ``nn.Linear.forward``.
"""
import torch


class LinearNorm(torch.nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, w_init_gain='linear'):
        super(LinearNorm, self).__init__()
        self.linear_layer = torch.nn.Linear(in_dim, out_dim, bias=bias)
        torch.nn.init.xavier_uniform_(
            self.linear_layer.weight,
            gain=torch.nn.init.calculate_gain(w_init_gain))


class ConvNorm(torch.nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1,
                 padding=None, dilation=1, bias=True, w_init_gain='linear'):
        super(ConvNorm, self).__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        if stride != 1 or dilation != 1:
            raise ValueError("the MI355X engine implements stride=1, dilation=1 "
                             "convolutions only (all the reference uses)")
        self.conv = torch.nn.Conv1d(in_channels, out_channels,
                                    kernel_size=kernel_size, stride=stride,
                                    padding=padding, dilation=dilation,
                                    bias=bias)
        torch.nn.init.xavier_uniform_(
            self.conv.weight, gain=torch.nn.init.calculate_gain(w_init_gain))


def __getattr__(name):
    """``from layers import TacotronSTFT, STFT`` (reference layers.py:42-80, inference.ipynb cell 2): the GPU mel
    front end lives in ``tacotron2_amd.audio``; resolved lazily so that the model containers do not import it."""
    if name in ('TacotronSTFT', 'STFT'):
        from . import audio
        return getattr(audio, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
