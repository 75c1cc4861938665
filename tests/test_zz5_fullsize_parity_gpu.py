"""Parity at BASELINE scale against the LIVE oracle (shared dropout masks), both precision modes.

  (i)   training: the WHOLE of ``synth_batch(64, 1234)`` -- B = 64, Ti_max = 177, To = 870: BASELINE configs[1] itself,
        the batch bench.py times (870 dependent decoder steps forward and through BPTT, every row of every 64-row
        tile and every workgroup of the 4 x B attention grids live), reference model.py:405-411 under autograd;
        ``T2AMD_FULLSIZE_B=16`` restores round 2's every-4th-utterance sub-batch (a quarter of the oracle time);
  (ii)  inference, B = 1, Ti = 100 (BASELINE configs[3]): greedy decode to a REAL gate stop beyond 300 steps,
        stop index exact, once with a comfortable and once with a SMALL crossing margin, reference
        model.py:435-449;
  (iii) batched ragged inference with configs[4]'s length distribution (32 texts) against per-utterance oracle runs;
  (iv)  BASELINE configs[4] itself: 256 ragged texts, >= 400 decoder steps with real stops -- the engine's stop vector
        against the batched oracle's and sampled utterances against per-utterance B = 1 oracle runs (fp32 mode: exact
        stops, 1e-4), then the bf16 mode (the skinny_wide64 / attn_energy4 kernels of B >= 256) against the same
        oracle with the bf16 tolerance and exact stops wherever the oracle's own crossing margin exceeds the measured
        bf16 gate noise; reference model.py:418-454 per utterance.

Tolerances (stated here, measured values land in gpurun_out/parity_fullsize_*.json):
  fp32 mode   outputs mean |diff| < 1e-4 (north star: mel L1 < 1e-4), max < 5e-4 * max(1, max|ref|);
              loss 1e-4 relative; gradients max |diff| < 1e-3 * max|ref| + 2e-6 per tensor (isolated ReLU-kink elements
              excepted as stated at the check: confined to <= 3 output channels, each < 2e-2 * max, tensor L2 error < 1e-3 and < 5e-4 without them); stop
              index exact.
  bf16 mode   decoder mel / gate mean |diff| < 4e-3, postnet mel < 6e-2, alignments < 2e-3, loss 2 % relative,
              whole-gradient cosine > 0.995, per-tensor relative L2 < 0.35 (the reference's own bf16-autocast drift
              at a 30-frame horizon is 6.7e-4 / 1.6e-2; 870 steps accumulate more of it).
"""
import json
import os

import pytest
import torch

import golden_util as gu
from oracle import tacotron2_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
OUT = os.path.join(gu.ROOT, "gpurun_out")


def _stats(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    d = (a - b).abs()
    return d.mean().item(), d.max().item(), b.abs().max().item()


def _report(name, rows):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_fullsize_%s.json" % name), "w") as f:
        json.dump(rows, f, indent=1)


def _model(hp, sd):
    from tacotron2_amd.model import Tacotron2
    m = Tacotron2(hp)
    m.load_state_dict(sd)
    return m.to(DEV)


# ---------------------------------------------------------------------------------------------------
# (i) training step at To = 870
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_train_case():
    from tacotron2_amd.synth import synth_batch
    hp = gu.make_hparams("")
    sd = gu.build_state_dict(hp, 1234)
    full = synth_batch(64, 1234)
    Bs = int(os.environ.get("T2AMD_FULLSIZE_B", "64"))
    idx = torch.arange(0, 64, 64 // Bs)[:Bs]
    text, il, mel, gate, ol = (t[idx] for t in full)
    Ti, To = int(il.max()), int(ol.max())
    assert To == 870 and Ti == 177
    batch = (text[:, :Ti].contiguous(), il, mel[:, :, :To].contiguous(), gate[:, :To].contiguous(), ol)
    masks = orc.draw_masks_train(hp, Bs, Ti, To, torch.Generator().manual_seed(1234))
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    import time
    t0 = time.perf_counter()
    oloss, oout, ograds, obufs = orc.train_step_grads(sd, hp, batch, masks)
    if Bs == 64:
        # the oracle THIS box just ran is held to the digest of the reference's own results at this size
        # (tests/golden/make_golden_fullsize.py; /root/reference does not exist here): reference -> oracle -> engine
        import sys
        sys.path.insert(0, gu.GOLDEN_DIR)
        try:
            import make_golden_fullsize as mf
        finally:
            sys.path.remove(gu.GOLDEN_DIR)
        dg = torch.load(os.path.join(gu.GOLDEN_DIR, mf.NAME + ".pt"), weights_only=False)
        worst = mf.compare_to_digest(dg, oout, oloss, ograds)
        _report("oracle_vs_reference_digest_B64", dict(worst_relative_deviation=worst, digest_meta={
            k: v for k, v in dg['meta'].items() if not isinstance(v, dict)}))
    return dict(hp=hp, sd=sd, batch=batch, masks=masks, oloss=oloss, oout=oout, ograds=ograds, obufs=obufs, B=Bs,
                shape="B=%d of synth_batch(64,1234) (BASELINE configs[1]%s), Ti=%d, To=%d; oracle step %.1f s"
                      % (Bs, "" if Bs == 64 else ": every %dth utterance" % (64 // Bs), Ti, To, time.perf_counter() - t0))


@pytest.fixture(scope="module")
def second_train_case():
    """A SECOND batch at the full LJSpeech horizon (round 6; VERDICT r05 weak 3: the ReLU-kink carve-out had only ever met the
    seed-1234 batch): every 4th utterance of synth_batch(64, 4321) with its own dropout masks -- other lengths, other kinks (or
    none).  The fp32 test below applies the SAME rules to it: every gradient inside 1e-3 of its tensor's max, or a single-row
    ReLU kink that the adjudication recognises."""
    from tacotron2_amd.synth import synth_batch
    hp = gu.make_hparams("")
    sd = gu.build_state_dict(hp, 1234)
    full = synth_batch(64, 4321)
    idx = torch.arange(0, 64, 4)
    text, il, mel, gate, ol = (t[idx] for t in full)
    Ti, To = int(il.max()), int(ol.max())
    batch = (text[:, :Ti].contiguous(), il, mel[:, :, :To].contiguous(), gate[:, :To].contiguous(), ol)
    masks = orc.draw_masks_train(hp, 16, Ti, To, torch.Generator().manual_seed(4321))
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    oloss, oout, ograds, obufs = orc.train_step_grads(sd, hp, batch, masks)
    return dict(hp=hp, sd=sd, batch=batch, masks=masks, oloss=oloss, oout=oout, ograds=ograds, obufs=obufs, B=16,
                shape="every 4th utterance of synth_batch(64, 4321), Ti=%d, To=%d" % (Ti, To), tag="second_batch_seed4321")


def _engine_step(case, precision):
    from tacotron2_amd.loss_function import Tacotron2Loss
    model = _model(case['hp'], case['sd']).train()
    model.precision = precision
    model.dropout_masks = gu.masks_to_engine(case['masks'], DEV)
    x, y = model.parse_batch(tuple(t.clone() for t in case['batch']))
    out = model(x)
    loss = Tacotron2Loss()(out, y)
    loss.backward()
    torch.cuda.synchronize()
    return model, out, loss


def _adjudicate_relu_kink(case, key, g_engine, g_oracle, over):
    """Direct adjudication of the ReLU-kink carve-out below, from the gradient difference itself.

    If engine and oracle disagree about relu'(z) for ONE pre-activation z[b*, co, t*] ~ 0 of encoder layer i, then dz differs by
    a scalar delta at that one row and, through the BatchNorm backward (biased batch statistics over all N = B Ti rows,
    reference model.py:160-167 under autograd),
        dW_engine[co] - dW_oracle[co] = c * ( window(r*) - mean_r window(r) - zhat[r*] mean_r(zhat[r] window(r)) ),
    window(r)[ci, tap] = x_in[b, ci, t + tap - pad]: ONE direction in the (Ci x k)-dimensional space of that output channel,
    computable from the ORACLE's own activations.  So: recompute the oracle's encoder stack up to layer i (CPU, f32), take the
    rows of channel co whose |z| is smallest, and test whether the observed difference of channel co is parallel to the
    direction of one of them.  Returns the finding (|z| of the row, cosine); the caller asserts on it."""
    import torch.nn.functional as F
    i = int(key.split('.')[2])
    hp, sd, masks = case['hp'], case['sd'], case['masks']
    text = case['batch'][0]
    x = F.embedding(text, sd['embedding.weight']).transpose(1, 2)
    for l in range(i):
        x = orc.apply_dropout(F.relu(orc._conv_bn(x, sd, 'encoder.convolutions.%d' % l, True, None)), 0.5, masks['enc'][l])
    z = orc._conv_bn(x, sd, 'encoder.convolutions.%d' % i, True, None)                 # (B, C, Ti): the pre-activation of relu
    gamma, beta = sd['encoder.convolutions.%d.1.weight' % i], sd['encoder.convolutions.%d.1.bias' % i]
    B, C, Ti = z.shape
    k = sd[key].shape[2]
    pad = (k - 1) // 2
    N = B * Ti
    xp = F.pad(x.double(), (pad, pad))                                                   # (B, Ci, Ti + 2 pad)
    D = (g_engine.detach().cpu().double() - g_oracle.double())
    findings = []
    for co in torch.nonzero(over.reshape(over.shape[0], -1).any(1)).flatten().tolist():
        zc = z[:, co, :].double()
        zhat = (zc - float(beta[co])) / float(gamma[co])
        mean_w = torch.stack([xp[:, :, tap:tap + Ti].sum((0, 2)) for tap in range(k)], 1) / N            # (Ci, k)
        mean_zw = torch.stack([(zhat.unsqueeze(1) * xp[:, :, tap:tap + Ti]).sum((0, 2)) for tap in range(k)], 1) / N
        d = D[co]
        best = None
        for flat in torch.argsort(zc.abs().flatten())[:6].tolist():                      # the rows nearest the kink
            b_, t_ = flat // Ti, flat % Ti
            v = xp[b_, :, t_:t_ + k] - mean_w - float(zhat[b_, t_]) * mean_zw
            cos = float((d * v).sum() / (d.norm() * v.norm()).clamp_min(1e-300))
            if best is None or abs(cos) > abs(best['cosine']):
                best = dict(channel=co, utterance=b_, position=t_, z_oracle=float(zc[b_, t_]), cosine=cos)
        best['z_scale'] = float(zc.abs().mean())
        findings.append(best)
    return findings


def test_train_step_To870_fp32(native_lib, full_train_case):
    _check_fp32_step(full_train_case)


def test_train_step_fp32_on_a_second_batch(native_lib, second_train_case):
    _check_fp32_step(second_train_case)


def _check_fp32_step(c):
    model, out, loss = _engine_step(c, 'fp32')
    rows, bad = [], []
    for i, nm in enumerate(('mel', 'mel_post', 'gate', 'align')):
        mean, mx, rmax = _stats(out[i], c['oout'][i])
        rows.append(dict(what=nm, mean=mean, max=mx, refmax=rmax))
        if not (mean < 1e-4 and mx < 5e-4 * max(1.0, rmax)):
            bad.append(rows[-1])
    el, ol = float(loss.detach()), float(c['oloss'])
    rows.append(dict(what='loss', engine=el, oracle=ol))
    if not abs(el - ol) < 1e-4 * max(1.0, abs(ol)):
        bad.append(rows[-1])
    for k, p in model.named_parameters():
        ref = c['ograds'][k]
        mean, mx, rmax = _stats(p.grad, ref)
        rows.append(dict(what='grad ' + k, mean=mean, max=mx, refmax=rmax))
        if k.endswith('.0.conv.bias'):
            # analytically zero (a bias in front of a BatchNorm): both sides hold rounding noise
            wmax = c['ograds'][k.replace('.bias', '.weight')].abs().max().item()
            if not p.grad.abs().max().item() < 1e-4 * wmax + 1e-5:
                bad.append(rows[-1])
            continue
        if not mx < 1e-3 * rmax + 2e-6:
            # ReLU kinks: of the 5.8 M (row, channel) pre-activations of an encoder layer at B = 64 a few land within
            # rounding noise of zero; engine and oracle then disagree about relu'(0+-) for that ONE element, and the whole
            # contribution dy[r, co] * x[r + tap, :] of that row flips in dW[co, :, :] (up to Ci * k elements of ONE output
            # channel) and in the BatchNorm gradients (seen at B = 64: 273 elements of one channel of a 1.3 M-element
            # tensor up to 8e-3 * max, while the tensor's mean error is 1e-5 * max).  Accepted when the elements beyond
            # the bound sit in at most 3 output channels, each stays below 2e-2 * max, the tensor as a whole agrees to
            # 1e-3 in L2 and, without those channels (where every element is inside the max bound again), to 5e-4 (measured
            # 1.3e-4: the split-bf16 gradient GEMMs); anything spread wider is a real discrepancy.
            d = (p.grad.detach().cpu().double() - ref.double()).abs()
            over = d > 1e-3 * rmax + 2e-6
            n_out = int(over.sum())
            chans = int(over.reshape(over.shape[0], -1).any(1).sum()) if over.dim() > 1 else n_out
            rel_l2 = float(d.norm() / ref.double().norm().clamp_min(1e-30))
            rest = d.clone()
            if over.dim() > 1:
                rest[over.reshape(over.shape[0], -1).any(1)] = 0.0          # the tensor without the kinked channels
            else:
                rest[over] = 0.0
            rel_l2_rest = float(rest.norm() / ref.double().norm().clamp_min(1e-30))
            rows[-1].update(outliers=n_out, outlier_channels=chans, rel_l2=rel_l2, rel_l2_without_those_channels=rel_l2_rest)
            if not (chans <= 3 and mx < 2e-2 * rmax and rel_l2 < 1e-3 and rel_l2_rest < 5e-4):
                bad.append(rows[-1])
            elif k.startswith('encoder.convolutions.') and k.endswith('.0.conv.weight'):
                # ... and the reading itself is put to the test (round 5): the difference of every flagged channel must BE the
                # contribution of one row whose pre-activation sits at the kink -- parallel (|cos| > 0.99) to the direction that
                # row's relu' flip takes through the BatchNorm backward, with |z| of that row inside the rounding noise of the
                # K = 2560-term convolution sum (2e-5 of the layer's mean |z|); anything else is NOT accepted as a kink.
                found = _adjudicate_relu_kink(c, k, p.grad, ref, over)
                rows[-1].update(kink_rows=found)
                # (measured at B = 64: |cos| 0.9995 / 0.99996, |z| 1.3e-6 / 1.2e-6 against a mean |z| of 0.73: profiles/r05_relu_kink_adjudicated.json)
                if not all(abs(f['cosine']) > 0.99 and abs(f['z_oracle']) < 2e-5 * f['z_scale'] for f in found):
                    bad.append(rows[-1])
    msd = model.state_dict()
    for k, v in c['obufs'].items():
        mean, mx, rmax = _stats(msd[k].float(), v.float())
        if not mx < 1e-5 * max(1.0, rmax):
            bad.append(dict(what='buffer ' + k, max=mx, refmax=rmax))
    _report("train_B%d_fp32%s" % (c['B'], "_" + c['tag'] if c.get('tag') else ""), dict(shape=c['shape'], rows=rows, bad=bad))
    assert not bad, bad[:8]


def test_train_step_To870_bf16(native_lib, full_train_case):
    c = full_train_case
    model, out, loss = _engine_step(c, 'bf16')
    rows = []
    # Limits = 3 x what is measured at B = 64 (round 6; measured in rounds 3-5: decoder mel 2.5e-4, postnet mel
    # 6.7e-3, gate 2.2e-4, alignments 2.1e-5; loss 2.9e-6 relative; whole-gradient cosine 0.99998; worst tensor 5.5 %).  Until
    # round 5 they were 9-16 x looser: the headline mode could lose an order of magnitude and stay green.
    lim = [8e-4, 2e-2, 7e-4, 7e-5]
    full = c['B'] == 64                                # (a T2AMD_FULLSIZE_B subset has other statistics: worst tensor 11 % at B = 16)
    fails = []
    for i, nm in enumerate(("mel", "mel_post", "gate", "align")):
        mean, mx, rmax = _stats(out[i], c['oout'][i])
        rows.append(dict(what="bf16 " + nm, mean=mean, max=mx, refmax=rmax, limit=lim[i]))
        if not mean < lim[i]:
            fails.append(rows[-1])
    worst, worst_k, dot, n1, n2 = 0.0, None, 0.0, 0.0, 0.0
    for k, p in model.named_parameters():
        ref = c['ograds'][k].double()
        g = p.grad.cpu().double()
        assert torch.isfinite(g).all(), k
        if k.endswith('.0.conv.bias'):
            continue
        rel = ((g - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        rows.append(dict(what="bf16 grad " + k, rel_l2=rel))
        if rel > worst:
            worst, worst_k = rel, k
        dot += float((g * ref).sum()); n1 += float((g * g).sum()); n2 += float((ref * ref).sum())
    cos = dot / (n1 ** 0.5 * n2 ** 0.5)
    el, ol = float(loss.detach()), float(c['oloss'])
    rows.append(dict(what="bf16 summary", engine_loss=el, oracle_loss=ol, grad_cosine=cos, worst_rel_l2=worst,
                     worst_tensor=worst_k))
    _report("train_B%d_bf16" % c['B'], dict(shape=c['shape'], rows=rows, fails=fails))
    assert not fails, fails
    assert abs(el - ol) < 1e-4 * abs(ol), (el, ol)
    assert cos > 0.9999, cos
    assert worst < (0.15 if full else 0.33), (worst_k, worst)


def test_train_step_To870_bf16x3(native_lib, full_train_case):
    """Round 6: the accurate-fast mode at BASELINE configs[1] itself -- B = 64, Ti = 177, 870 dependent decoder
    steps forward and through BPTT with the LSTM tiles on split-bf16 operand images and the dense products on the split-bf16 GEMM.
    It must meet what the north star asks of the fp32 parity mode (mel mean |diff| < 1e-4 is the stated tolerance; the verdict's bar
    for this mode is 1e-5) -- limits below are 3 x what was measured on the first run of the round (profiles/r06_*)."""
    c = full_train_case
    model, out, loss = _engine_step(c, 'bf16x3')
    rows, fails = [], []
    lim = X3_LIMITS
    for i, nm in enumerate(("mel", "mel_post", "gate", "align")):
        mean, mx, rmax = _stats(out[i], c['oout'][i])
        rows.append(dict(what="bf16x3 " + nm, mean=mean, max=mx, refmax=rmax, limit=lim['out'][i]))
        if not mean < lim['out'][i]:
            fails.append(rows[-1])
    worst, worst_k, dot, n1, n2, over = 0.0, None, 0.0, 0.0, 0.0, []
    for k, p in model.named_parameters():
        ref = c['ograds'][k].double()
        g = p.grad.cpu().double()
        assert torch.isfinite(g).all(), k
        if k.endswith('.0.conv.bias'):
            continue
        rel = ((g - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        mx = float((g - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        rows.append(dict(what="bf16x3 grad " + k, rel_l2=rel, max_over_refmax=mx))
        if mx > 1e-3:
            over.append((k, mx))
        if rel > worst:
            worst, worst_k = rel, k
        dot += float((g * ref).sum()); n1 += float((g * g).sum()); n2 += float((ref * ref).sum())
    cos = dot / (n1 ** 0.5 * n2 ** 0.5)
    el, ol = float(loss.detach()), float(c['oloss'])
    rows.append(dict(what="bf16x3 summary", engine_loss=el, oracle_loss=ol, grad_cosine=cos, worst_rel_l2=worst,
                     worst_tensor=worst_k, tensors_beyond_1e3_of_max=over))
    _report("train_B%d_bf16x3" % c['B'], dict(shape=c['shape'], rows=rows, fails=fails))
    assert not fails, fails
    assert abs(el - ol) < lim['loss'] * abs(ol), (el, ol)
    assert 1.0 - cos < lim['one_minus_cos'], cos
    assert worst < lim['worst'], (worst_k, worst)
    # per-element bar of the fp32 mode (1e-3 of the tensor's max): held by every tensor outside the encoder convolutions' ReLU-kink
    # channels (the encoder and the prenet stay on the exact-f32 product in this mode for exactly that reason: engine._fg `exact`)
    stray = [(k, v) for k, v in over if not k.startswith('encoder.convolutions.')]
    assert not stray and len(over) <= lim['kink_tensors'] and all(v < lim['kink_max'] for _, v in over), over


# Measured in round 6 (profiles/r06_d_parity_fullsize_train_B64_bf16x3.json: decoder mel 3.7e-7, postnet mel 9.2e-6, gate 3.3e-7,
# alignments 2.6e-8; loss equal to the oracle's to every printed digit; whole-gradient cosine 1 - 7e-10; worst tensor 3.6e-4 relative
# L2; beyond 1e-3 of the tensor's max only the three encoder tensors of the fp32 mode's own ReLU-kink rows, at the fp32 mode's values
# 8.0e-3 / 1.3e-3 / 2.0e-3): limits = 3 x measured, far inside the verdict's bar for this mode (decoder mel 1e-5).
X3_LIMITS = dict(out=[1.2e-6, 3e-5, 1.1e-6, 8e-8], loss=1e-6, one_minus_cos=5e-9, worst=1.2e-3, kink_tensors=4, kink_max=2.5e-2)


# ---------------------------------------------------------------------------------------------------
# (ii) B = 1, Ti = 100, real gate stop beyond 300 steps
# ---------------------------------------------------------------------------------------------------
def _choose_thresholds(sig, lo_step):
    """sig: sigmoid(gate) trajectory of ONE utterance decoded without stopping.  Returns a list of
    (threshold, stop_length, margin) with the first crossing at index >= lo_step: one with the widest margin the
    trajectory offers, one with a small margin (1e-4: two orders above the engine's f32 noise on the gate)."""
    T = sig.numel()
    best = None
    for t in range(lo_step, T):
        m = float(sig[:t].max())
        if float(sig[t]) > m:                    # a first-crossing candidate for any thr in (m, sig[t])
            margin = (float(sig[t]) - m) / 2
            if best is None or margin > best[2]:
                best = (m + margin, t + 1, margin)
    assert best is not None, "no first-crossing candidate beyond step %d" % lo_step
    out = [best]
    t = best[1] - 1
    m = float(sig[:t].max())
    if float(sig[t]) - m > 4e-4:
        out.append((float(sig[t]) - 1e-4, t + 1, 1e-4))
    return out


def test_inference_B1_Ti100_real_gate_stop(native_lib):
    hp = gu.make_hparams("max_decoder_steps=520")
    sd = gu.build_state_dict(hp, 1234, perturb_bn=True)
    # Random weights give the gate an early transient and then a slow decay (its running maximum falls in the first
    # ~100 steps whatever the seed).  Negating the CONTEXT half of the gate weight turns the attention drift into a
    # slow rise with the decoder-LSTM noise on top: the first crossing of a threshold lands hundreds of steps in,
    # the regime a trained model stops in.
    wg = sd['decoder.gate_layer.linear_layer.weight'].clone()
    wg[:, hp.decoder_rnn_dim:] *= -1.0
    sd['decoder.gate_layer.linear_layer.weight'] = wg
    text = gu.make_text([100], 4242)
    keep = orc.draw_masks_infer(hp, 1, 520, torch.Generator().manual_seed(9))
    # the oracle decoded without a stop: the whole gate trajectory
    (mel_o, _, gate_o, al_o), _, _ = orc.tacotron2_inference(sd, hp, text, keep, 520, 2.0)
    sig = torch.sigmoid(gate_o.reshape(-1))
    cases = _choose_thresholds(sig, 300)
    rows = []
    for thr, L, margin in cases:
        ohp = gu.make_hparams("max_decoder_steps=520")
        ohp.gate_threshold = thr
        oref, olen, _ = orc.tacotron2_inference(sd, ohp, text, keep)
        assert olen.tolist() == [L]
        model = _model(ohp, sd).eval()
        model.dropout_masks = dict(prenet_infer=keep.to(DEV))
        out = model.inference(text.to(DEV))
        got = model.last_inference_lengths.tolist()
        row = dict(threshold=thr, oracle_stop=L, engine_stop=got[0], margin=margin)
        for i, nm in enumerate(('mel', 'mel_post', 'gate', 'align')):
            if got == [L]:
                mean, mx, rmax = _stats(out[i], oref[i])
                row[nm] = dict(mean=mean, max=mx, refmax=rmax)
        rows.append(row)
        _report("infer_B1", rows)
        assert got == [L], row                                             # bit-exact stop index, >= 300 steps
        for nm in ('mel', 'mel_post', 'gate', 'align'):
            assert row[nm]['mean'] < 1e-4 and row[nm]['max'] < 5e-4 * max(1.0, row[nm]['refmax']), (nm, row)
    # bf16 mode on the same case: reported, with its own tolerance on the frames both runs produced
    thr, L, margin = cases[0]
    ohp = gu.make_hparams("max_decoder_steps=520")
    ohp.gate_threshold = thr
    model = _model(ohp, sd).eval()
    model.precision = 'bf16'
    model.dropout_masks = dict(prenet_infer=keep.to(DEV))
    out = model.inference(text.to(DEV))
    Lb = int(model.last_inference_lengths[0])
    n = min(L, Lb)
    mean, mx, rmax = _stats(out[0][:, :, :n], mel_o[:, :, :n])
    rows.append(dict(what="bf16 mode", oracle_stop=L, engine_stop=Lb, mel_mean=mean, mel_max=mx, refmax=rmax))
    _report("infer_B1", rows)
    assert mean < 2e-2 * max(float(mel_o.abs().mean()), 1e-3), rows[-1]


# ---------------------------------------------------------------------------------------------------
# (iii) ragged batch with configs[4] lengths against per-utterance oracle runs
# ---------------------------------------------------------------------------------------------------
def test_batched_inference_config5_lengths(native_lib):
    from tacotron2_amd.synth import synth_lengths
    B, steps = 32, 240
    ti, _ = synth_lengths(256, 1234)
    lens = [int(v) for v in ti[::8]]                       # 32 of the 256 config-5 lengths, descending (187 .. 15)
    hp = gu.make_hparams("max_decoder_steps=%d" % steps)
    sd = gu.build_state_dict(hp, 1234, perturb_bn=True)
    wg = sd['decoder.gate_layer.linear_layer.weight'].clone()      # slow rise of the gate, as in the B = 1 test above
    wg[:, hp.decoder_rnn_dim:] *= -1.0
    sd['decoder.gate_layer.linear_layer.weight'] = wg
    text = gu.make_text(lens, 777)
    keep = orc.draw_masks_infer(hp, B, steps, torch.Generator().manual_seed(10))
    il = torch.tensor(lens)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    (_, _, gate_o, _), _, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, 2.0, input_lengths=il)
    sig = torch.sigmoid(gate_o.reshape(B, steps))
    # threshold: at least a third of the utterances stop before max_decoder_steps, at most two of them inside the
    # initial transient (< 30 steps), widest worst-case margin
    best = None
    for thr in torch.linspace(float(sig.min()), float(sig.max()), 800)[1:-1].tolist():
        over = sig > thr
        stop = torch.where(over.any(1), over.float().argmax(1) + 1, torch.full((B,), steps))
        if int((stop < steps).sum()) < B // 3 or int((stop < 30).sum()) > 2:
            continue
        marg = min(float((sig[b, :int(stop[b])] - thr).abs().min()) for b in range(B))
        if best is None or marg > best[1]:
            best = (thr, marg, stop)
    assert best is not None
    thr, marg, stop = best
    ohp = gu.make_hparams("max_decoder_steps=%d" % steps)
    ohp.gate_threshold = thr
    model = _model(ohp, sd).eval()
    model.dropout_masks = dict(prenet_infer=keep.to(DEV))
    out = model.inference(text.to(DEV), il.to(DEV))
    got = model.last_inference_lengths.tolist()
    rows = dict(threshold=thr, margin=marg, expected=stop.tolist(), engine=got, per_utterance=[])
    _report("infer_config5", rows)
    assert got == stop.tolist()
    # per-utterance oracle runs (reference semantics: B == 1 on the unpadded text) on a spread of utterances
    early = [b for b in range(B) if int(stop[b]) < steps]
    for b in sorted(set([0, B - 1] + early[:8])):
        L = int(stop[b])
        oref, olen, _ = orc.tacotron2_inference(sd, ohp, text[b:b + 1, :lens[b]], keep[:, :, b:b + 1])
        assert olen.tolist() == [L]
        r = dict(b=b, Ti=lens[b], stop=L)
        for nm, g_, w_ in (('mel', out[0][b, :, :L], oref[0][0]), ('mel_post', out[1][b, :, :L], oref[1][0]),
                           ('align', out[3][b, :L, :lens[b]], oref[3][0])):
            mean, mx, rmax = _stats(g_, w_)
            r[nm] = dict(mean=mean, max=mx, refmax=rmax)
        rows['per_utterance'].append(r)
        _report("infer_config5", rows)
        for nm in ('mel', 'mel_post', 'align'):
            assert r[nm]['mean'] < 1e-4 and r[nm]['max'] < 5e-4 * max(1.0, r[nm]['refmax']), r
        assert out[0][b, :, L:].abs().sum().item() == 0
    # Early-exit compaction: the run above kept all 32 rows to the end (fewer than 64 finished rows never
    # free a tile).  Forcing a compaction at every poll that saw >= 3 finished utterances must change nothing but the
    # batch the kernels see: same stop frames, same outputs (rows are computed independently of their neighbours).
    assert 'compacted' not in model.last_decode_path
    for prec in ('fp32', 'bf16'):
        model.precision = prec
        model.compact_min_rows = None
        ref = [o.clone() for o in model.inference(text.to(DEV), il.to(DEV))]
        ref_len = model.last_inference_lengths.tolist()
        model.compact_min_rows = 3
        got2 = model.inference(text.to(DEV), il.to(DEV))
        assert 'compacted' in model.last_decode_path, model.last_decode_path
        assert model.last_inference_lengths.tolist() == ref_len
        rows['compaction_' + prec] = dict(path=model.last_decode_path,
                                          mel_max_diff=float((got2[0] - ref[0]).abs().max()),
                                          align_max_diff=float((got2[3] - ref[3]).abs().max()))
        _report("infer_config5", rows)
        assert rows['compaction_' + prec]['mel_max_diff'] < 1e-5 and rows['compaction_' + prec]['align_max_diff'] < 1e-6
    model.compact_min_rows = None
    model.precision = 'fp32'


# ---------------------------------------------------------------------------------------------------
# (iv) BASELINE configs[4] at its real batch size: 256 ragged texts, real stops
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config5_case():
    """256 texts with configs[4]'s length distribution, the gate made to rise slowly (see (ii)), decoded by the BATCHED
    oracle without stopping; the threshold is then put where at least a third of the utterances stop before the cap, at
    most eight inside the initial transient, with the widest worst-case margin over every (utterance, step) the
    stop rule looks at."""
    from tacotron2_amd.synth import synth_lengths
    B, steps = 256, 440
    ti, _ = synth_lengths(B, 1234)
    lens = [int(v) for v in ti]
    hp = gu.make_hparams("max_decoder_steps=%d" % steps)
    sd = gu.build_state_dict(hp, 1234, perturb_bn=True)
    wg = sd['decoder.gate_layer.linear_layer.weight'].clone()
    wg[:, hp.decoder_rnn_dim:] *= -1.0
    sd['decoder.gate_layer.linear_layer.weight'] = wg
    text = gu.make_text(lens, 778)
    keep = orc.draw_masks_infer(hp, B, steps, torch.Generator().manual_seed(11))
    il = torch.tensor(lens)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    import time
    t0 = time.perf_counter()
    (mel_o, post_o, gate_o, al_o), _, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, 2.0, input_lengths=il)
    oracle_s = time.perf_counter() - t0
    sig = torch.sigmoid(gate_o.reshape(B, steps).double())
    best = None
    for thr in torch.linspace(float(sig.min()), float(sig.max()), 1200)[1:-1].tolist():
        over = sig > thr
        stop = torch.where(over.any(1), over.float().argmax(1) + 1, torch.full((B,), steps))
        if int((stop < steps).sum()) < B // 3 or int((stop < 30).sum()) > 8:
            continue
        looked = torch.arange(steps).unsqueeze(0) < stop.unsqueeze(1)          # what the stop rule reads
        per_utt = torch.where(looked, (sig - thr).abs(), torch.full_like(sig, 9.0)).min(1).values
        marg = float(per_utt.min())
        if best is None or marg > best[1]:
            best = (thr, marg, stop, per_utt)
    assert best is not None
    thr, marg, stop, per_utt = best
    return dict(B=B, steps=steps, lens=lens, hp_str="max_decoder_steps=%d" % steps, sd=sd, text=text, keep=keep, il=il,
                thr=thr, margin=marg, stop=stop, per_utt_margin=per_utt, mel=mel_o, post=post_o, align=al_o, sig=sig,
                oracle_s=oracle_s)


def _config5_engine(c, precision):
    ohp = gu.make_hparams(c['hp_str'])
    ohp.gate_threshold = c['thr']
    model = _model(ohp, c['sd']).eval()
    model.precision = precision
    model.dropout_masks = dict(prenet_infer=c['keep'].to(DEV))
    with torch.no_grad():
        out = model.inference(c['text'].to(DEV), c['il'].to(DEV))
    torch.cuda.synchronize()
    return ohp, model, out


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_config5_B256_fp32_stops_and_per_utterance_oracle(native_lib, config5_case, precision):
    """(precision 'bf16x3', round 6: the accurate-fast mode is held to the fp32 mode's bar -- every one of the 256 stops exact at a
    worst-case margin of ~1e-4, the same output tolerances.)"""
    c = config5_case
    B, steps, stop = c['B'], c['steps'], c['stop']
    ohp, model, out = _config5_engine(c, precision)
    got = model.last_inference_lengths.tolist()
    rows = dict(shape="B=256 ragged (synth_lengths(256,1234): Ti %d..%d), max_decoder_steps=%d" % (min(c['lens']), max(c['lens']), steps),
                threshold=c['thr'], margin=c['margin'], batched_oracle_seconds=c['oracle_s'],
                stopped_before_cap=int((stop < steps).sum()), earliest_stop=int(stop.min()),
                stops_equal=(got == stop.tolist()), decode_path=model.last_decode_path, per_utterance=[])
    _report("infer_config5_B256_" + precision, rows)
    assert got == stop.tolist(), [(b, got[b], int(stop[b])) for b in range(B) if got[b] != int(stop[b])][:8]
    # the whole batch against the batched oracle, which was decoded WITHOUT stopping: the decoder mel is causal, so every
    # frame before an utterance's stop must agree; the postnet looks 10 frames ahead (5 layers x k = 5), so its output is
    # compared up to 10 frames before the stop (the last frames, which see the zero padding behind the stop, are pinned by
    # the per-utterance oracle runs below)
    Tout = out[0].shape[2]
    assert Tout == int(stop.max())
    tt = torch.arange(Tout).unsqueeze(0)
    for i, (nm, ref, look) in enumerate((('mel', c['mel'], 0), ('mel_post', c['post'], 10))):
        valid = (tt < (stop - look).clamp(min=0).unsqueeze(1)).unsqueeze(1).to(ref.dtype)
        mean, mx, rmax = _stats(out[i].cpu() * valid, ref[:, :, :Tout] * valid)
        rows[nm + '_vs_batched_oracle'] = dict(mean=mean, max=mx, refmax=rmax)
        assert mean < 1e-4 and mx < 5e-4 * max(1.0, rmax), (nm, rows[nm + '_vs_batched_oracle'])
    # >= 16 sampled utterances against per-utterance B = 1 oracle runs on the unpadded text (reference semantics)
    early = [b for b in range(B) if int(stop[b]) < steps]
    sample = sorted(set([0, 1, B // 2, B - 2, B - 1] + early[::max(1, len(early) // 12)][:12] + list(range(7, B, 37))))
    assert len(sample) >= 16
    for b in sample:
        L = int(stop[b])
        Lb = c['lens'][b]
        oref, olen, _ = orc.tacotron2_inference(c['sd'], ohp, c['text'][b:b + 1, :Lb], c['keep'][:, :, b:b + 1])
        assert olen.tolist() == [L], (b, olen.tolist(), L)
        r = dict(b=b, Ti=Lb, stop=L)
        for nm, g_, w_ in (('mel', out[0][b, :, :L], oref[0][0]), ('mel_post', out[1][b, :, :L], oref[1][0]),
                           ('align', out[3][b, :L, :Lb], oref[3][0])):
            mean, mx, rmax = _stats(g_, w_)
            r[nm] = dict(mean=mean, max=mx, refmax=rmax)
        rows['per_utterance'].append(r)
        _report("infer_config5_B256_" + precision, rows)
        for nm in ('mel', 'mel_post', 'align'):
            assert r[nm]['mean'] < 1e-4 and r[nm]['max'] < 5e-4 * max(1.0, r[nm]['refmax']), r
        assert out[0][b, :, L:].abs().sum().item() == 0


def test_config5_B256_bf16_kernels_against_the_oracle(native_lib, config5_case):
    """The bf16 mode at B = 256 runs skinny_wide64_kernel / attn_energy4_kernel (csrc/rnn.hip, csrc/attention.hip): the
    first END-TO-END check of those two against the oracle.  With these weights (gate layer amplified x 60 so that the
    stop rule has something to cross) bf16 operands move sigmoid(gate) by up to ~1e-2 over 440 steps, and every early
    stop is a slow rise through the threshold whose own margin is ~1e-3: bit-exact stops cannot be asked of this mode.
    What is asked: (i) the gate trajectory stays within BF16_GATE_NOISE of the oracle's on every frame both sides
    produced; (ii) wherever the oracle's margin exceeds that bound the stop decision is the oracle's; (iii) every stop
    that differs is EXPLAINED by the bound -- at the earlier of the two stop frames the oracle's own gate is within
    BF16_GATE_NOISE of the threshold; (iv) the mels agree to the bf16 tolerance on the common frames."""
    BF16_GATE_NOISE = 2.5e-2
    c = config5_case
    B, steps, stop, thr = c['B'], c['steps'], c['stop'], c['thr']
    ohp, model, out = _config5_engine(c, 'bf16')
    got = model.last_inference_lengths.cpu()
    safe = c['per_utt_margin'] > BF16_GATE_NOISE
    gate_e = torch.sigmoid(out[2].float().cpu().reshape(B, -1).double())
    Tg = gate_e.shape[1]
    both = (torch.arange(Tg).unsqueeze(0) < torch.minimum(got, stop).unsqueeze(1))
    gate_noise = float(((gate_e - c['sig'][:, :Tg]).abs() * both).max())
    mism, unexplained = [], []
    for b in range(B):
        if int(got[b]) == int(stop[b]):
            continue
        first = min(int(got[b]), int(stop[b]))               # the frame (1-based) where one side stopped and the other went on
        gap = abs(float(c['sig'][b, first - 1]) - thr)        # the oracle's own distance from the threshold there
        mism.append((b, int(got[b]), int(stop[b]), gap))
        if gap > BF16_GATE_NOISE:
            unexplained.append(mism[-1])
    rows = dict(shape="B=256 ragged, max_decoder_steps=%d, bf16 mode" % steps, decode_path=model.last_decode_path,
                threshold=thr, utterances_with_safe_margin=int(safe.sum()), stops_equal=B - len(mism),
                stop_mismatches=mism[:60], unexplained_mismatches=unexplained,
                measured_sigmoid_gate_noise_max=gate_noise, BF16_GATE_NOISE=BF16_GATE_NOISE)
    scale = max(float(c['mel'].abs().mean()), 1e-3)
    num = den = 0.0
    worst = (0.0, -1)
    for b in range(B):
        n = int(min(got[b], stop[b]))
        d = (out[0][b, :, :n].float().cpu() - c['mel'][b, :, :n]).abs()
        num += float(d.sum()); den += d.numel()
        if float(d.mean()) > worst[0]:
            worst = (float(d.mean()), b)
    rows.update(mel_mean_abs_diff=num / den, oracle_mel_mean_abs=scale, worst_utterance_mel_mean=worst[0], worst_utterance=worst[1])
    _report("infer_config5_B256_bf16", rows)
    assert torch.isfinite(out[1]).all()
    assert gate_noise < BF16_GATE_NOISE, rows                                             # (i)
    assert all(int(got[b]) == int(stop[b]) for b in range(B) if bool(safe[b])), rows      # (ii)
    assert int(safe.sum()) >= B // 4, "the margin rule leaves too few utterances to pin: %d" % int(safe.sum())
    assert not unexplained, unexplained[:8]                                               # (iii)
    assert B - len(mism) >= B // 2, rows
    assert num / den < 2e-2 * scale, rows                                                 # (iv)
    assert worst[0] < 0.1 * scale, rows
