"""CPU oracle for the Tacotron 2 mel hot path  — TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) *restatement* of the
algorithm of NVIDIA/tacotron2 ``model.py`` for the path named in
BASELINE.json (Encoder -> teacher-forced / free-running Decoder -> Postnet ->
output masking -> loss).  It exists to check the HIP engine and to serve as
``bench.py``'s ``cpu_baseline`` leg.  Nothing under ``tacotron2_amd/`` may
import it: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``.

Pinning: the reference ships no golden vectors.  The
oracle is pinned against the *reference itself*, executed in the build
container through the import shim of ``tests/golden/make_golden.py``; the
outputs of that run are committed as ``tests/golden/*.pt`` and re-checked by
``tests/test_oracle_golden.py`` on every machine (the reference tree does not
travel to the GPU box).  Third-party arithmetic the reference relies on and
this file restates: ``torch.nn.{Embedding,Conv1d,BatchNorm1d,LSTM,LSTMCell,
Linear}``, ``F.{relu,softmax,dropout}``, ``torch.bmm`` (reference pins
"PyTorch 1.0", README.md:27; behaviour is stable for fp32).

Differences from the reference's code shape (not its arithmetic):
  * functional: weights come from a ``state_dict``-style mapping with the
    reference's key names, nothing is stored on ``self``;
  * every dropout site takes an explicit keep-mask (uint8/bool, reference
    tensor layout) — the reference draws them from the global RNG
    (``F.dropout``), which makes parity impossible without injection
    (SURVEY.md H2.1);
  * the bi-LSTM is written as explicit per-step cell arithmetic with
    length masking instead of ``pack_padded_sequence`` + cuDNN/ATen LSTM.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def get_mask_from_lengths(lengths, max_len=None):
    """reference utils.py:6-10 — True on valid positions."""
    if max_len is None:
        max_len = int(lengths.max().item())
    ids = torch.arange(max_len, device=lengths.device)
    return ids.unsqueeze(0) < lengths.unsqueeze(1)


def apply_dropout(x, p, keep):
    """``F.dropout(x, p, training=True)`` with an injected keep-mask.

    ATen computes ``x * (bernoulli(1-p) / (1-p))``; the scale is formed in the
    tensor dtype, which is what is done here."""
    if keep is None:
        return x
    scale = torch.ones((), dtype=x.dtype) / torch.tensor(1.0 - p, dtype=x.dtype)
    return x * (keep.to(x.dtype) * scale)


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """``torch.nn.LSTMCell`` semantics: gate chunk order i, f, g, o; both
    biases added (reference model.py:352-354, 366-369 call sites)."""
    gates = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, dim=1)
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c_new = f * c + i * g
    h_new = o * torch.tanh(c_new)
    return h_new, c_new


def draw_keep(shape, p, generator):
    """One Bernoulli(1-p) keep-mask, uint8."""
    return (torch.rand(shape, generator=generator) >= p).to(torch.uint8)


def draw_masks_train(hp, B, Ti, To, generator):
    """Keep-masks for one training forward, in the reference's tensor layouts
    and draw order: 3x encoder (B,C,Ti) -> 2x prenet
    (To+1,B,P) -> per step att (B,H) then dec (B,H) -> 5x postnet (B,C,To)."""
    m = OrderedDict()
    E = hp.encoder_embedding_dim
    m['enc'] = [draw_keep((B, E, Ti), 0.5, generator)
                for _ in range(hp.encoder_n_convolutions)]
    m['prenet'] = [draw_keep((To + 1, B, hp.prenet_dim), 0.5, generator) for _ in range(2)]
    att, dec = [], []
    for _ in range(To):
        att.append(draw_keep((B, hp.attention_rnn_dim), hp.p_attention_dropout, generator))
        dec.append(draw_keep((B, hp.decoder_rnn_dim), hp.p_decoder_dropout, generator))
    m['att'] = torch.stack(att)
    m['dec'] = torch.stack(dec)
    chans = [hp.postnet_embedding_dim] * (hp.postnet_n_convolutions - 1) + [hp.n_mel_channels]
    m['post'] = [draw_keep((B, c, To), 0.5, generator) for c in chans]
    return m


def draw_masks_infer(hp, B, max_steps, generator):
    """Prenet keep-masks for free-running decoding: (steps, 2, B, P)
    (the Prenet's dropout is active in eval(), reference model.py:99)."""
    return draw_keep((max_steps, 2, B, hp.prenet_dim), 0.5, generator)


# ----------------------------------------------------------------------------
# Encoder  (reference model.py:150-201)
# ----------------------------------------------------------------------------
def _conv_bn(x, sd, prefix, training, new_buffers):
    """``nn.Sequential(ConvNorm, BatchNorm1d)`` (reference model.py:160-167,
    116-139).  x: (B, C, T)."""
    w = sd[prefix + '.0.conv.weight']
    b = sd[prefix + '.0.conv.bias']
    pad = (w.shape[2] - 1) // 2
    y = F.conv1d(x, w, b, stride=1, padding=pad)
    rm = sd[prefix + '.1.running_mean'].clone()
    rv = sd[prefix + '.1.running_var'].clone()
    y = F.batch_norm(y, rm, rv, sd[prefix + '.1.weight'], sd[prefix + '.1.bias'],
                     training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    if training and new_buffers is not None:
        new_buffers[prefix + '.1.running_mean'] = rm
        new_buffers[prefix + '.1.running_var'] = rv
        new_buffers[prefix + '.1.num_batches_tracked'] = \
            sd[prefix + '.1.num_batches_tracked'] + 1
    return y


def bilstm(x, lengths, sd, prefix='encoder.lstm'):
    """1-layer bidirectional LSTM over a right-padded batch with packed-sequence
    semantics (reference model.py:180-188): the reverse direction starts at each
    utterance's own last token and outputs are zero at padded positions.
    x: (B, T, C) -> (B, T, 2H)."""
    B, T, _ = x.shape
    outs = []
    for suffix, reverse in (('', False), ('_reverse', True)):
        w_ih = sd['%s.weight_ih_l0%s' % (prefix, suffix)]
        w_hh = sd['%s.weight_hh_l0%s' % (prefix, suffix)]
        b_ih = sd['%s.bias_ih_l0%s' % (prefix, suffix)]
        b_hh = sd['%s.bias_hh_l0%s' % (prefix, suffix)]
        H = w_hh.shape[1]
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        ys = [None] * T
        steps = range(T - 1, -1, -1) if reverse else range(T)
        for t in steps:
            valid = (t < lengths).to(x.dtype).unsqueeze(1)          # (B,1)
            h_new, c_new = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
            h = valid * h_new + (1 - valid) * h
            c = valid * c_new + (1 - valid) * c
            ys[t] = valid * h_new
        outs.append(torch.stack(ys, dim=1))
    return torch.cat(outs, dim=2)


def encoder_forward(emb, lengths, sd, hp, masks, training, new_buffers=None):
    """reference model.py:173-190 (training / teacher-forced path) and
    :192-201 (``lengths is None``: inference path, no packing).
    emb: (B, C, Ti) = embedding(text).transpose(1,2)."""
    x = emb
    for i in range(hp.encoder_n_convolutions):
        x = F.relu(_conv_bn(x, sd, 'encoder.convolutions.%d' % i, training, new_buffers))
        if training:
            x = apply_dropout(x, 0.5, masks['enc'][i])
    x = x.transpose(1, 2)
    if lengths is None:
        lengths = torch.full((x.shape[0],), x.shape[1], dtype=torch.long)
    T = int(lengths.max().item())
    return bilstm(x[:, :T], lengths, sd)


# ----------------------------------------------------------------------------
# Decoder  (reference model.py:204-454)
# ----------------------------------------------------------------------------
def prenet(x, sd, keep0, keep1):
    """reference model.py:97-100 — dropout p=0.5 ALWAYS on."""
    x = apply_dropout(F.relu(F.linear(x, sd['decoder.prenet.layers.0.linear_layer.weight'])), 0.5, keep0)
    x = apply_dropout(F.relu(F.linear(x, sd['decoder.prenet.layers.1.linear_layer.weight'])), 0.5, keep1)
    return x


def attention_step(h_att, memory, processed_memory, w_prev, w_cum, mask, sd, score_mask_value):
    """Location-sensitive attention, one step (reference model.py:43-86 with
    LocationLayer :22-26).  mask: (B,Ti) True on PADDED positions or None."""
    pfx = 'decoder.attention_layer.'
    q = F.linear(h_att.unsqueeze(1), sd[pfx + 'query_layer.linear_layer.weight'])      # (B,1,A)
    cat = torch.stack((w_prev, w_cum), dim=1)                                           # (B,2,Ti)
    wc = sd[pfx + 'location_layer.location_conv.conv.weight']
    loc = F.conv1d(cat, wc, None, padding=(wc.shape[2] - 1) // 2)                       # (B,F,Ti)
    loc = F.linear(loc.transpose(1, 2), sd[pfx + 'location_layer.location_dense.linear_layer.weight'])
    e = F.linear(torch.tanh(q + loc + processed_memory), sd[pfx + 'v.linear_layer.weight']).squeeze(-1)
    if mask is not None:
        e = e.masked_fill(mask, score_mask_value)
    w = F.softmax(e, dim=1)
    ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
    return ctx, w


def decoder_step(x, state, memory, processed_memory, mask, sd, hp, keep_att, keep_dec,
                 score_mask_value=-float('inf')):
    """``Decoder.decode`` (reference model.py:340-379).  ``state`` is the tuple
    the reference keeps on ``self`` (model.py:258-289)."""
    h_a, c_a, h_d, c_d, w, w_cum, ctx = state
    h_a, c_a = lstm_cell(torch.cat((x, ctx), -1), h_a, c_a,
                         sd['decoder.attention_rnn.weight_ih'], sd['decoder.attention_rnn.weight_hh'],
                         sd['decoder.attention_rnn.bias_ih'], sd['decoder.attention_rnn.bias_hh'])
    h_a = apply_dropout(h_a, hp.p_attention_dropout, keep_att)      # stored state is the dropped one
    ctx, w = attention_step(h_a, memory, processed_memory, w, w_cum, mask, sd, score_mask_value)
    w_cum = w_cum + w
    h_d, c_d = lstm_cell(torch.cat((h_a, ctx), -1), h_d, c_d,
                         sd['decoder.decoder_rnn.weight_ih'], sd['decoder.decoder_rnn.weight_hh'],
                         sd['decoder.decoder_rnn.bias_ih'], sd['decoder.decoder_rnn.bias_hh'])
    h_d = apply_dropout(h_d, hp.p_decoder_dropout, keep_dec)
    hc = torch.cat((h_d, ctx), dim=1)
    mel = F.linear(hc, sd['decoder.linear_projection.linear_layer.weight'],
                   sd['decoder.linear_projection.linear_layer.bias'])
    gate = F.linear(hc, sd['decoder.gate_layer.linear_layer.weight'],
                    sd['decoder.gate_layer.linear_layer.bias'])
    return mel, gate, (h_a, c_a, h_d, c_d, w, w_cum, ctx)


def _init_state(memory, hp):
    """reference model.py:258-285."""
    B, Ti, E = memory.shape
    z = memory.new_zeros
    return (z(B, hp.attention_rnn_dim), z(B, hp.attention_rnn_dim),
            z(B, hp.decoder_rnn_dim), z(B, hp.decoder_rnn_dim),
            z(B, Ti), z(B, Ti), z(B, E))


def decoder_train_forward(memory, mels, memory_lengths, sd, hp, masks,
                          score_mask_value=-float('inf')):
    """Teacher-forced ``Decoder.forward`` (reference model.py:381-416).
    mels: (B, n_mel, To).  Returns mel (B,n_mel,To), gate (B,To), align (B,To,Ti)."""
    B = memory.shape[0]
    frames = mels.transpose(1, 2).transpose(0, 1)                         # (To,B,n_mel)
    frames = torch.cat((memory.new_zeros(1, B, hp.n_mel_channels), frames), dim=0)
    x_all = prenet(frames, sd, masks['prenet'][0], masks['prenet'][1])     # (To+1,B,P)
    pad_mask = ~get_mask_from_lengths(memory_lengths, memory.shape[1])
    pm = F.linear(memory, sd['decoder.attention_layer.memory_layer.linear_layer.weight'])
    state = _init_state(memory, hp)
    mel_o, gate_o, ali_o = [], [], []
    training = masks.get('att') is not None
    for t in range(frames.shape[0] - 1):
        mel, gate, state = decoder_step(
            x_all[t], state, memory, pm, pad_mask, sd, hp,
            masks['att'][t] if training else None, masks['dec'][t] if training else None,
            score_mask_value)
        mel_o.append(mel)
        gate_o.append(gate.squeeze(1))
        ali_o.append(state[4])
    return (torch.stack(mel_o).transpose(0, 1).transpose(1, 2),
            torch.stack(gate_o).transpose(0, 1).contiguous(),
            torch.stack(ali_o).transpose(0, 1))


def decoder_inference(memory, sd, hp, prenet_keep, max_decoder_steps=None, gate_threshold=None,
                      memory_lengths=None):
    """Free-running ``Decoder.inference`` (reference model.py:418-454), B == 1 in
    the reference.  For B > 1 (BASELINE config 5) every utterance
    follows the B == 1 semantics independently: padded memory positions are
    masked, an utterance stops after the first frame whose sigmoid(gate) >
    threshold (strict, stopping frame included), the others continue.
    Returns mel (B,n_mel,T), gate (B,T,1), align (B,T,Ti), lengths (B)."""
    max_steps = hp.max_decoder_steps if max_decoder_steps is None else max_decoder_steps
    thr = hp.gate_threshold if gate_threshold is None else gate_threshold
    B = memory.shape[0]
    pm = F.linear(memory, sd['decoder.attention_layer.memory_layer.linear_layer.weight'])
    mask = None
    if memory_lengths is not None:
        mask = ~get_mask_from_lengths(memory_lengths, memory.shape[1])
    state = _init_state(memory, hp)
    x = memory.new_zeros(B, hp.n_mel_channels)
    mel_o, gate_o, ali_o = [], [], []
    done = torch.zeros(B, dtype=torch.bool)
    lengths = torch.zeros(B, dtype=torch.long)
    hit_max = False
    while True:
        t = len(mel_o)
        xin = prenet(x, sd, prenet_keep[t, 0], prenet_keep[t, 1])
        mel, gate, state = decoder_step(xin, state, memory, pm, mask, sd, hp, None, None)
        mel_o.append(mel)
        gate_o.append(gate)
        ali_o.append(state[4])
        fired = (torch.sigmoid(gate.squeeze(1)) > thr) & ~done
        lengths[fired] = t + 1
        done |= fired
        if bool(done.all()):
            break
        if len(mel_o) == max_steps:
            hit_max = True
            lengths[~done] = max_steps
            break
        x = mel
    mel_out = torch.stack(mel_o).transpose(0, 1).transpose(1, 2)
    gate_out = torch.stack(gate_o).transpose(0, 1).contiguous()
    align = torch.stack(ali_o).transpose(0, 1)
    return mel_out, gate_out, align, lengths, hit_max


# ----------------------------------------------------------------------------
# Postnet  (reference model.py:103-146)
# ----------------------------------------------------------------------------
def postnet_forward(x, sd, hp, masks, training, new_buffers=None, frame_valid=None):
    """``frame_valid`` (B,1,T) is only used by batched inference: frames beyond an
    utterance's own stop are zeroed before every convolution, which is the zero padding a B == 1
    run of that utterance sees."""
    n = hp.postnet_n_convolutions
    for i in range(n):
        if frame_valid is not None:
            x = x * frame_valid
        x = _conv_bn(x, sd, 'postnet.convolutions.%d' % i, training, new_buffers)
        if i < n - 1:
            x = torch.tanh(x)
        if training:
            x = apply_dropout(x, 0.5, masks['post'][i])
    return x


# ----------------------------------------------------------------------------
# Tacotron2  (reference model.py:457-529)
# ----------------------------------------------------------------------------
def parse_output(outputs, output_lengths, hp):
    """reference model.py:487-497.  The fills are in-place on ``.data`` exactly
    like the reference: autograd does not see them, and the tensor saved by the
    first Postnet convolution (the decoder mel) is zeroed at padded frames
    before backward runs."""
    if hp.mask_padding and output_lengths is not None:
        pad = ~get_mask_from_lengths(output_lengths, outputs[0].shape[2])    # (B,To)
        m3 = pad.unsqueeze(1).expand(-1, hp.n_mel_channels, -1)
        outputs[0].data.masked_fill_(m3, 0.0)
        outputs[1].data.masked_fill_(m3, 0.0)
        outputs[2].data.masked_fill_(pad, 1e3)
    return outputs


def tacotron2_forward(sd, hp, inputs, masks, training=True, new_buffers=None,
                      score_mask_value=-float('inf')):
    """``Tacotron2.forward`` (reference model.py:499-515).
    inputs = (text_padded, input_lengths, mel_padded, max_len, output_lengths)."""
    text, text_lengths, mels, _max_len, output_lengths = inputs
    emb = F.embedding(text, sd['embedding.weight']).transpose(1, 2)
    memory = encoder_forward(emb, text_lengths, sd, hp, masks, training, new_buffers)
    mel, gate, align = decoder_train_forward(memory, mels, text_lengths, sd, hp, masks,
                                             score_mask_value)
    post = postnet_forward(mel, sd, hp, masks, training, new_buffers)
    mel_post = mel + post
    return parse_output([mel, mel_post, gate, align], output_lengths, hp)


def tacotron2_inference(sd, hp, text, prenet_keep, max_decoder_steps=None, gate_threshold=None,
                        input_lengths=None):
    """``Tacotron2.inference`` (reference model.py:517-529), eval mode: BatchNorm
    uses running statistics, only the Prenet dropout is live.  With
    ``input_lengths`` (B > 1) the encoder follows SURVEY.md H3: activations beyond
    each utterance's length are zeroed before every convolution so that every
    utterance sees exactly what a B == 1 run on its unpadded text would see."""
    emb = F.embedding(text, sd['embedding.weight']).transpose(1, 2)
    if input_lengths is None:
        memory = encoder_forward(emb, None, sd, hp, None, False)
    else:
        valid = get_mask_from_lengths(input_lengths, emb.shape[2]).unsqueeze(1).to(emb.dtype)
        x = emb
        for i in range(hp.encoder_n_convolutions):
            x = F.relu(_conv_bn(x * valid, sd, 'encoder.convolutions.%d' % i, False, None))
        T = int(input_lengths.max().item())
        memory = bilstm(x.transpose(1, 2)[:, :T], input_lengths, sd)
    mel, gate, align, lengths, hit_max = decoder_inference(
        memory, sd, hp, prenet_keep, max_decoder_steps, gate_threshold, input_lengths)
    frame_valid = None
    if input_lengths is not None:
        frame_valid = get_mask_from_lengths(lengths, mel.shape[2]).unsqueeze(1).to(mel.dtype)
        mel = mel * frame_valid
    post = postnet_forward(mel, sd, hp, None, False, frame_valid=frame_valid)
    mel_post = mel + post
    if frame_valid is not None:
        mel_post = mel_post * frame_valid
    return [mel, mel_post, gate, align], lengths, hit_max


def tacotron2_loss(outputs, targets):
    """reference loss_function.py:8-19."""
    mel_t, gate_t = targets
    mel, mel_post, gate = outputs[0], outputs[1], outputs[2]
    return (F.mse_loss(mel, mel_t) + F.mse_loss(mel_post, mel_t) +
            F.binary_cross_entropy_with_logits(gate.reshape(-1, 1), gate_t.reshape(-1, 1)))


def train_step_grads(sd, hp, batch, masks):
    """One forward+backward (reference train.py:214-226 without the optimiser):
    returns (loss, outputs, grads-by-name, new BN buffers)."""
    text, in_len, mel_t, gate_t, out_len = batch
    leaf = OrderedDict()
    for k, v in sd.items():
        if v.dtype.is_floating_point and not k.endswith(('running_mean', 'running_var')):
            leaf[k] = v.detach().clone().requires_grad_(True)
        else:
            leaf[k] = v
    new_buffers = {}
    inputs = (text, in_len, mel_t, int(in_len.max().item()), out_len)
    out = tacotron2_forward(leaf, hp, inputs, masks, True, new_buffers)
    loss = tacotron2_loss(out, (mel_t, gate_t))
    loss.backward()
    grads = OrderedDict((k, v.grad) for k, v in leaf.items()
                        if isinstance(v, torch.Tensor) and v.requires_grad)
    return loss.detach(), [o.detach() for o in out], grads, new_buffers
