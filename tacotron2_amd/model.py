"""Tacotron 2 module surface for the MI355X engine.

Drop-in for ``model.Tacotron2`` of NVIDIA/tacotron2 (reference model.py:457-529):
same constructor (``hparams``), same ``parse_batch`` / ``forward`` / ``inference`` /
``parse_output`` signatures and return layouts, same sub-module attribute names and
*parameter creation order*, hence the same ``state_dict`` key set (60 parameters +
24 BatchNorm buffers) and — under the same ``torch.manual_seed`` — bit-identical
initial values (tests/test_boundary.py checks both against the reference's dump).

What is different: the sub-modules are parameter *containers*.  No arithmetic of the hot
path runs through ``torch.nn`` forward methods; ``Tacotron2.forward`` / ``.inference``
hand the parameters to ``tacotron2_amd.engine`` which executes hand-written HIP kernels
(``include/tacotron2_amd.h``).  Calling it with CPU tensors raises: there is no fallback.
"""
from math import sqrt

import torch
from torch import nn

from .layers import ConvNorm, LinearNorm
from .utils import to_gpu
from . import engine


def _conv_bn(cin, cout, ksize, gain):
    return nn.Sequential(
        ConvNorm(cin, cout, kernel_size=ksize, stride=1, padding=int((ksize - 1) / 2),
                 dilation=1, w_init_gain=gain),
        nn.BatchNorm1d(cout))


class LocationLayer(nn.Module):
    """Parameters of reference model.py:10-20 (2->F conv k=31 no bias, F->A dense no bias)."""

    def __init__(self, attention_n_filters, attention_kernel_size, attention_dim):
        super().__init__()
        self.location_conv = ConvNorm(2, attention_n_filters, kernel_size=attention_kernel_size,
                                      padding=int((attention_kernel_size - 1) / 2), bias=False,
                                      stride=1, dilation=1)
        self.location_dense = LinearNorm(attention_n_filters, attention_dim, bias=False,
                                         w_init_gain='tanh')


class Attention(nn.Module):
    """Parameters of reference model.py:29-41.  ``score_mask_value`` is kept because
    train.py:76 assigns it; the kernel excludes padded positions from the softmax, which is
    what any very negative fill value does in fp32."""

    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim,
                 attention_location_n_filters, attention_location_kernel_size):
        super().__init__()
        self.query_layer = LinearNorm(attention_rnn_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.v = LinearNorm(attention_dim, 1, bias=False)
        self.location_layer = LocationLayer(attention_location_n_filters,
                                            attention_location_kernel_size, attention_dim)
        self.score_mask_value = -float("inf")


class Prenet(nn.Module):
    """Parameters of reference model.py:89-95."""

    def __init__(self, in_dim, sizes):
        super().__init__()
        dims = [in_dim] + list(sizes)
        self.layers = nn.ModuleList(
            [LinearNorm(dims[i], dims[i + 1], bias=False) for i in range(len(sizes))])


class Postnet(nn.Module):
    """Parameters of reference model.py:108-139: n convs of width postnet_embedding_dim, k=5."""

    def __init__(self, hparams):
        super().__init__()
        n, k = hparams.postnet_n_convolutions, hparams.postnet_kernel_size
        mel, dim = hparams.n_mel_channels, hparams.postnet_embedding_dim
        chans = [mel] + [dim] * (n - 1) + [mel]
        self.convolutions = nn.ModuleList()
        for i in range(n):
            self.convolutions.append(_conv_bn(chans[i], chans[i + 1], k, 'tanh' if i < n - 1 else 'linear'))


class Encoder(nn.Module):
    """Parameters of reference model.py:154-171: 3 x (conv k=5 + BN) and a 1-layer bi-LSTM."""

    def __init__(self, hparams):
        super().__init__()
        dim = hparams.encoder_embedding_dim
        self.convolutions = nn.ModuleList(
            [_conv_bn(dim, dim, hparams.encoder_kernel_size, 'relu')
             for _ in range(hparams.encoder_n_convolutions)])
        self.lstm = nn.LSTM(dim, int(dim / 2), 1, batch_first=True, bidirectional=True)


class Decoder(nn.Module):
    """Parameters and scalar attributes of reference model.py:205-241."""

    def __init__(self, hparams):
        super().__init__()
        for name in ('n_mel_channels', 'n_frames_per_step', 'encoder_embedding_dim',
                     'attention_rnn_dim', 'decoder_rnn_dim', 'prenet_dim', 'max_decoder_steps',
                     'gate_threshold', 'p_attention_dropout', 'p_decoder_dropout'):
            setattr(self, name, getattr(hparams, name))
        mel_dim = hparams.n_mel_channels * hparams.n_frames_per_step
        enc = hparams.encoder_embedding_dim
        self.prenet = Prenet(mel_dim, [hparams.prenet_dim, hparams.prenet_dim])
        self.attention_rnn = nn.LSTMCell(hparams.prenet_dim + enc, hparams.attention_rnn_dim)
        self.attention_layer = Attention(hparams.attention_rnn_dim, enc, hparams.attention_dim,
                                         hparams.attention_location_n_filters,
                                         hparams.attention_location_kernel_size)
        self.decoder_rnn = nn.LSTMCell(hparams.attention_rnn_dim + enc, hparams.decoder_rnn_dim, 1)
        self.linear_projection = LinearNorm(hparams.decoder_rnn_dim + enc, mel_dim)
        self.gate_layer = LinearNorm(hparams.decoder_rnn_dim + enc, 1, bias=True, w_init_gain='sigmoid')


# attributes distributed.apply_gradient_allreduce() leaves on a module: never copied or pickled with it
_DP_STATE = ('_grad_sync', '_hook_sync', '_t2amd_dp_applied')


class Tacotron2(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        engine.check_hparams(hparams)
        self.hparams = hparams
        self.mask_padding = hparams.mask_padding
        self.fp16_run = hparams.fp16_run
        self.n_mel_channels = hparams.n_mel_channels
        self.n_frames_per_step = hparams.n_frames_per_step
        self.embedding = nn.Embedding(hparams.n_symbols, hparams.symbols_embedding_dim)
        std = sqrt(2.0 / (hparams.n_symbols + hparams.symbols_embedding_dim))
        val = sqrt(3.0) * std
        self.embedding.weight.data.uniform_(-val, val)
        self.encoder = Encoder(hparams)
        self.decoder = Decoder(hparams)
        self.postnet = Postnet(hparams)
        # test hook: dict of injected keep-masks in ENGINE layout (see engine.MaskSource);
        # None = Philox masks seeded from torch's RNG.
        self.dropout_masks = None
        self.last_inference_lengths = None
        # 'fp32': exact-f32 MFMA forward (parity mode).  'bf16': matrix operands rounded to bf16, f32
        # accumulation, f32 master weights / cell state / saved activations (throughput mode, training only).
        # 'bf16x3' (round 6): the decoder's LSTM products on split-bf16 operand pairs (hi*hi + lo*hi + hi*lo on the bf16 MFMA,
        # ~2^-17 relative), everything else as 'fp32' -- inside the 1e-4 mel bound with every gate stop equal, ~18 % faster.
        self.precision = 'fp32'
        self._output_dtype = None

    # -- the engine's packed weight images (engine._Run.cached) live on the module but are not part of it ------------
    def invalidate_weight_cache(self):
        """Drop the packed / transposed / bf16 weight images; the next call rebuilds them (needed after ``param.data``
        edits -- everything torch tracks, ``load_state_dict`` and device / dtype moves invalidate by themselves)."""
        engine.invalidate_weight_cache(self)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_weight_cache()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_weight_cache()
        return out

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_weight_cache', None)          # device images of the weights: rebuilt on demand, never pickled
        # process-group state of the data-parallel wrapper -- AND the flag that says it is there: a copy that kept
        # `_t2amd_dp_applied` without the exchange would make apply_gradient_allreduce() return early and then train
        # without any all-reduce, ranks drifting apart silently
        for k in _DP_STATE:
            state.pop(k, None)
        return state

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == '_weight_cache' or k in _DP_STATE:
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    # -- inference.ipynb cell 7: ``model.cuda().eval().half()`` -----------------------------------
    def half(self):
        """Reduced-precision mode without reduced-precision *storage*: the parameters stay f32 master
        weights (the engine refuses anything else), the matrix products switch to the bf16 compute mode
        and ``inference`` returns float16 tensors, so that the notebook's half-precision WaveGlow takes
        ``mel_outputs_postnet`` unchanged (inference.ipynb cells 7, 13, 15)."""
        self.precision = 'bf16'
        self._output_dtype = torch.float16
        return self

    def bfloat16(self):
        self.precision = 'bf16'
        self._output_dtype = torch.bfloat16
        return self

    def float(self):
        self.precision = 'fp32'
        self._output_dtype = None
        return super().float()

    # -- reference model.py:473-485 ---------------------------------------------------------
    def parse_batch(self, batch):
        text_padded, input_lengths, mel_padded, gate_padded, output_lengths = batch
        text_padded = to_gpu(text_padded).long()
        input_lengths = to_gpu(input_lengths).long()
        max_len = torch.max(input_lengths.data).item()
        mel_padded = to_gpu(mel_padded).float()
        gate_padded = to_gpu(gate_padded).float()
        output_lengths = to_gpu(output_lengths).long()
        return ((text_padded, input_lengths, mel_padded, max_len, output_lengths),
                (mel_padded, gate_padded))

    # -- reference model.py:487-497 ---------------------------------------------------------
    def parse_output(self, outputs, output_lengths=None):
        """Kept for API compatibility.  ``forward`` already returns masked outputs (the masking is
        fused into the engine's output kernel); applying it again is idempotent."""
        if self.mask_padding and output_lengths is not None:
            To = outputs[0].size(2)
            pad = torch.arange(To, device=output_lengths.device).unsqueeze(0) >= output_lengths.unsqueeze(1)
            outputs[0].data.masked_fill_(pad.unsqueeze(1), 0.0)
            outputs[1].data.masked_fill_(pad.unsqueeze(1), 0.0)
            outputs[2].data.masked_fill_(pad, 1e3)
        return outputs

    # -- reference model.py:499-515 ---------------------------------------------------------
    def forward(self, inputs):
        text_inputs, text_lengths, mels, max_len, output_lengths = inputs
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        buffers = dict(self.named_buffers())
        outs = engine.Tacotron2TrainFunction.apply(
            self, names, buffers, text_inputs, text_lengths.data, mels, int(max_len),
            output_lengths.data, *params)
        return list(outs)

    # -- reference model.py:517-529 ---------------------------------------------------------
    def inference(self, inputs, input_lengths=None):
        """``inputs``: (B, Ti) token ids.  B == 1 with ``input_lengths=None`` is the reference
        contract.  With B > 1 pass ``input_lengths`` (descending or not): every utterance is
        decoded as the reference would decode it alone; outputs are padded to the
        longest utterance and the per-utterance frame counts are left in
        ``self.last_inference_lengths``."""
        params = dict(self.named_parameters())
        buffers = dict(self.named_buffers())
        with torch.no_grad():
            outs, lengths, hit_max = engine.infer(self, params, buffers, inputs, input_lengths)
        self.last_inference_lengths = lengths
        if hit_max:
            print("Warning! Reached max decoder steps")
        out_dtype = self._output_dtype
        if out_dtype is None and self.embedding.weight.dtype in (torch.float16, torch.bfloat16):
            out_dtype = self.embedding.weight.dtype      # parameters stored in reduced precision (model.to(dtype))
        if out_dtype is not None:
            outs = [o.to(out_dtype) for o in outs]
        return outs
