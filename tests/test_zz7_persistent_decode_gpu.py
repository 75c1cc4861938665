"""The persistent, weight-stationary single-utterance decode kernel (csrc/decode_persist.hip, BASELINE configs[3]).

It runs the whole of Decoder.inference's loop (reference model.py:435-449) as ONE launch; its arithmetic is the
launch chain's bf16 mode for B <= 8 (bf16 LSTM weight rows against f32 inputs, everything else f32) with different
summation orders and prenet layer 0 folded through the frame projection.  Checked here:
  * against the launch chain on the same weights / text / prenet dropout stream (tight: same operand precision);
  * against the f32 oracle with the bf16-mode tolerance;
  * a real gate stop hundreds of steps in lands on the same frame as the launch chain;
  * the engine reports which path ran (a silent fall back to the chain would otherwise pass every check).
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


def _model(hp, sd, precision='bf16'):
    from tacotron2_amd.model import Tacotron2
    m = Tacotron2(hp)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.precision = precision
    return m


def _run(model, text, keep, persistent):
    from tacotron2_amd import engine
    old = engine.PERSISTENT_DECODE
    engine.PERSISTENT_DECODE = persistent
    try:
        model.dropout_masks = dict(prenet_infer=keep.to(DEV))
        with torch.no_grad():
            out = model.inference(text.to(DEV))
        torch.cuda.synchronize()
    finally:
        engine.PERSISTENT_DECODE = old
    return [o.float().cpu() for o in out], int(model.last_inference_lengths[0]), model.last_decode_path


@pytest.mark.parametrize("hpstr,Ti,steps", [(gu.TINY_HP, 23, 48), ("", 100, 160), ("", 187, 64), ("", 7, 40)])
def test_persistent_matches_launch_chain_and_oracle(native_lib, hpstr, Ti, steps):
    hp = gu.make_hparams((hpstr + "," if hpstr else "") + "max_decoder_steps=%d" % steps)
    hp.gate_threshold = 2.0                                     # forced length: compare every frame
    sd = gu.build_state_dict(hp, 321, perturb_bn=True)
    text = gu.make_text([Ti], 55)
    keep = orc.draw_masks_infer(hp, 1, steps, torch.Generator().manual_seed(4))
    model = _model(hp, sd)
    model.persist_trace = True
    pout, plen, ppath = _run(model, text, keep, True)
    cout, clen, cpath = _run(model, text, keep, False)
    assert ppath == 'persistent', ppath
    assert cpath.startswith('launch chain'), cpath
    assert plen == clen == steps
    (omel, opost, ogate, oalign), _, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, 2.0)
    rows = dict(shape="H=%d Ti=%d steps=%d" % (hp.attention_rnn_dim, Ti, steps))
    scale = max(float(omel.abs().mean()), 1e-3)
    for i, nm in enumerate(("mel", "mel_post", "gate", "align")):
        d = (pout[i] - cout[i]).abs()
        rows[nm + " persistent vs chain"] = dict(mean=float(d.mean()), max=float(d.max()))
    dm = (pout[0] - omel).abs()
    rows["mel persistent vs oracle"] = dict(mean=float(dm.mean()), max=float(dm.max()), oracle_mean_abs=scale)
    dc = (cout[0] - omel).abs()
    rows["mel chain vs oracle"] = dict(mean=float(dc.mean()), max=float(dc.max()))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_persistent_H%d_Ti%d.json" % (hp.attention_rnn_dim, Ti)), "w") as f:
        json.dump(rows, f, indent=1)
    # same operand precision, different summation order: agreement far inside the bf16-vs-f32 error
    assert rows["mel persistent vs chain"]["mean"] < 2e-2 * rows["mel chain vs oracle"]["mean"] + 2e-6, rows
    assert rows["align persistent vs chain"]["max"] < 1e-4, rows
    assert rows["mel persistent vs oracle"]["mean"] < 2e-2 * scale, rows
    assert torch.isfinite(pout[1]).all()


def test_persistent_real_gate_stop(native_lib):
    """Greedy decode to the gate stop (threshold chosen on the f32 oracle's trajectory, first crossing beyond 300 steps):
    the persistent kernel, the launch chain and the oracle stop on the same frame."""
    hp = gu.make_hparams("max_decoder_steps=520")
    sd = gu.build_state_dict(hp, 1234, perturb_bn=True)
    wg = sd['decoder.gate_layer.linear_layer.weight'].clone()
    wg[:, hp.decoder_rnn_dim:] *= -1.0                        # slow rise of the gate (see test_zz5)
    sd['decoder.gate_layer.linear_layer.weight'] = wg
    text = gu.make_text([100], 4242)
    keep = orc.draw_masks_infer(hp, 1, 520, torch.Generator().manual_seed(9))
    (_, _, gate_o, _), _, _ = orc.tacotron2_inference(sd, hp, text, keep, 520, 2.0)
    sig = torch.sigmoid(gate_o.reshape(-1))
    best = None
    for t in range(300, 520):
        m = float(sig[:t].max())
        if float(sig[t]) > m and (best is None or (float(sig[t]) - m) / 2 > best[2]):
            best = (m + (float(sig[t]) - m) / 2, t + 1, (float(sig[t]) - m) / 2)
    assert best is not None
    hp.gate_threshold = best[0]
    model = _model(hp, sd)
    pout, plen, ppath = _run(model, text, keep, True)
    cout, clen, cpath = _run(model, text, keep, False)
    with open(os.path.join(OUT, "parity_persistent_gate_stop.json"), "w") as f:
        json.dump(dict(threshold=best[0], margin=best[2], oracle_stop=best[1], persistent_stop=plen, chain_stop=clen,
                       path=ppath), f)
    assert ppath == 'persistent'
    assert plen == clen == best[1], (plen, clen, best)
    assert pout[0].shape == cout[0].shape
    assert (pout[0] - cout[0]).abs().mean().item() < 1e-4


def test_persistent_gate_stop_small_margin(native_lib):
    """The stop rule under a SMALL crossing margin (1e-4 on sigmoid(gate), reference model.py:443).  The persistent kernel
    computes in the launch chain's bf16 mode, so the threshold is put on the bf16 launch chain's OWN forced trajectory:
    1e-4 below a first-crossing value beyond 300 steps whose running maximum lies at least 1e-4 below the threshold.
    Persistent kernel and chain must stop on that very frame.  The same is then asked against the f32 ORACLE's trajectory
    with the margin the measured bf16-vs-f32 gate noise allows (recorded; the 1e-4-margin case against the oracle is
    the fp32 chain's, tests/test_zz5)."""
    hp = gu.make_hparams("max_decoder_steps=520")
    sd = gu.build_state_dict(hp, 1234, perturb_bn=True)
    wg = sd['decoder.gate_layer.linear_layer.weight'].clone()
    wg[:, hp.decoder_rnn_dim:] *= -1.0
    sd['decoder.gate_layer.linear_layer.weight'] = wg
    text = gu.make_text([100], 4242)
    keep = orc.draw_masks_infer(hp, 1, 520, torch.Generator().manual_seed(9))
    hp.gate_threshold = 2.0
    model = _model(hp, sd)
    pf, _, ppath = _run(model, text, keep, True)
    cf, _, cpath = _run(model, text, keep, False)
    assert ppath == 'persistent' and cpath.startswith('launch chain')
    sig_p = torch.sigmoid(pf[2].reshape(-1).double())
    sig_c = torch.sigmoid(cf[2].reshape(-1).double())
    (_, _, gate_o, _), _, _ = orc.tacotron2_inference(sd, hp, text, keep, 520, 2.0)
    sig_o = torch.sigmoid(gate_o.reshape(-1).double())
    rec = dict(persistent_vs_chain_sigmoid_gate_max=float((sig_p - sig_c).abs().max()),
               persistent_vs_oracle_sigmoid_gate_max=float((sig_p - sig_o).abs().max()), cases=[])
    # (a) 1e-4 margin on the chain's own trajectory
    cand = None
    for t in range(300, 520):
        m = float(sig_c[:t].max())
        if float(sig_c[t]) - m >= 2e-4:
            cand = (float(sig_c[t]) - 1e-4, t + 1)
            break
    assert cand is not None, "no first crossing with a 2e-4 step beyond frame 300"
    hp.gate_threshold = cand[0]
    _, plen, ppath = _run(model, text, keep, True)
    _, clen, _ = _run(model, text, keep, False)
    rec['cases'].append(dict(kind="1e-4 below the bf16 chain's crossing value", threshold=cand[0], expected_stop=cand[1],
                             persistent_stop=plen, chain_stop=clen, path=ppath))
    # (b) against the oracle's trajectory: the smallest margin that is still 4x the measured bf16-vs-f32 gate noise
    noise = rec['persistent_vs_oracle_sigmoid_gate_max']
    want = max(1e-4, 4.0 * noise)
    ocand = None
    for t in range(300, 520):
        m = float(sig_o[:t].max())
        if float(sig_o[t]) - m >= 2.0 * want:
            ocand = (float(sig_o[t]) - want, t + 1)
            break
    olen = None
    if ocand is not None:
        hp.gate_threshold = ocand[0]
        _, olen, _ = _run(model, text, keep, True)
        rec['cases'].append(dict(kind="oracle trajectory, margin = max(1e-4, 4 x measured gate noise)", margin=want,
                                 threshold=ocand[0], expected_stop=ocand[1], persistent_stop=olen))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_persistent_gate_stop_small_margin.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert ppath == 'persistent'
    assert plen == clen == cand[1], rec
    assert rec['persistent_vs_chain_sigmoid_gate_max'] < 2e-5, rec
    if ocand is not None:
        assert olen == ocand[1], rec


def test_persistent_gives_up_and_the_launch_chain_takes_over(native_lib, capsys, monkeypatch):
    """A workgroup that never arrives (shared GPU, fewer than H/4 free CUs): every spin is bounded, the kernel reports
    T2AMD_PERSIST_TIMEOUT, and Tacotron2.inference decodes the utterance on the launch chain -- same result as asking for
    the chain directly, with a line on stderr; nothing hangs."""
    hp = gu.make_hparams("max_decoder_steps=32")
    hp.gate_threshold = 2.0
    sd = gu.build_state_dict(hp, 321, perturb_bn=True)
    text = gu.make_text([40], 56)
    keep = orc.draw_masks_infer(hp, 1, 32, torch.Generator().manual_seed(5))
    model = _model(hp, sd)
    cout, clen, cpath = _run(model, text, keep, False)
    monkeypatch.setenv("T2AMD_PB_TIMEOUT_TICKS", "1")
    pout, plen, ppath = _run(model, text, keep, True)
    monkeypatch.delenv("T2AMD_PB_TIMEOUT_TICKS")
    assert ppath == 'launch chain (persistent kernel timed out)', ppath
    assert "gave up" in capsys.readouterr().err
    assert plen == clen
    for a, b in zip(pout, cout):
        assert torch.equal(a, b)
    # after a timeout the model stays on the launch chain for a few calls (exponential back-off: a shared GPU would time
    # out again, 30 ms each), then tries the persistent kernel again
    gout, glen, gpath = _run(model, text, keep, True)
    assert gpath.startswith('launch chain (persistent kernel timed out recently') and glen == clen, gpath
    assert model._persist_backoff == 3
    model._persist_backoff = 0
    gout, glen, gpath = _run(model, text, keep, True)
    assert gpath == 'persistent' and glen == clen and model._persist_timeouts == 0


# ---------------------------------------------------------------------------------------------------------------------
# fp32 parity mode on the persistent kernel (round 3): exact f32 LSTM rows held on the CU -- attention rows in LDS, decoder
# rows in REGISTERS (t2amd_dec_persist.weights_f32) -- so BASELINE configs[3] no longer has to choose between the 1e-4 /
# bit-exact-stop mode and the one-launch decode loop.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Ti,steps", [(100, 160), (187, 64), (7, 40)])
def test_persistent_fp32_matches_launch_chain_and_oracle(native_lib, Ti, steps):
    hp = gu.make_hparams("max_decoder_steps=%d" % steps)
    hp.gate_threshold = 2.0
    sd = gu.build_state_dict(hp, 321, perturb_bn=True)
    text = gu.make_text([Ti], 55)
    keep = orc.draw_masks_infer(hp, 1, steps, torch.Generator().manual_seed(4))
    model = _model(hp, sd, 'fp32')
    pout, plen, ppath = _run(model, text, keep, True)
    cout, clen, cpath = _run(model, text, keep, False)
    assert ppath == 'persistent', ppath
    assert cpath.startswith('launch chain'), cpath
    assert plen == clen == steps
    (omel, opost, ogate, oalign), _, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, 2.0)
    rows = dict(shape="fp32 H=%d Ti=%d steps=%d" % (hp.attention_rnn_dim, Ti, steps))
    for i, (nm, ref) in enumerate((("mel", omel), ("mel_post", opost), ("gate", ogate), ("align", oalign))):
        d = (pout[i] - ref).abs()
        dc = (pout[i] - cout[i]).abs()
        rows[nm] = dict(vs_oracle_mean=float(d.mean()), vs_oracle_max=float(d.max()), vs_chain_max=float(dc.max()),
                        refmax=float(ref.abs().max()))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_persistent_fp32_Ti%d.json" % Ti), "w") as f:
        json.dump(rows, f, indent=1)
    for nm in ("mel", "mel_post", "gate", "align"):      # the fp32-mode tolerances of tests/test_zz5
        assert rows[nm]["vs_oracle_mean"] < 1e-4 and rows[nm]["vs_oracle_max"] < 5e-4 * max(1.0, rows[nm]["refmax"]), (nm, rows)


def test_persistent_fp32_real_gate_stop_exact_with_1e4_margin(native_lib):
    """Greedy decode to the gate stop on the ORACLE's trajectory (first crossing beyond 300 steps), once with the widest
    margin it offers and once 1e-4 below the crossing value: the f32 persistent kernel stops on the oracle's frame."""
    hp = gu.make_hparams("max_decoder_steps=520")
    sd = gu.build_state_dict(hp, 1234, perturb_bn=True)
    wg = sd['decoder.gate_layer.linear_layer.weight'].clone()
    wg[:, hp.decoder_rnn_dim:] *= -1.0
    sd['decoder.gate_layer.linear_layer.weight'] = wg
    text = gu.make_text([100], 4242)
    keep = orc.draw_masks_infer(hp, 1, 520, torch.Generator().manual_seed(9))
    (omel, _, gate_o, _), _, _ = orc.tacotron2_inference(sd, hp, text, keep, 520, 2.0)
    sig = torch.sigmoid(gate_o.reshape(-1))
    best = None
    for t in range(300, 520):
        m = float(sig[:t].max())
        if float(sig[t]) > m and (best is None or (float(sig[t]) - m) / 2 > best[2]):
            best = (m + (float(sig[t]) - m) / 2, t + 1, (float(sig[t]) - m) / 2)
    assert best is not None
    cases = [best]
    t = best[1] - 1
    if float(sig[t]) - float(sig[:t].max()) > 4e-4:
        cases.append((float(sig[t]) - 1e-4, t + 1, 1e-4))
    rows = []
    for thr, L, margin in cases:
        hp.gate_threshold = thr
        model = _model(hp, sd, 'fp32')
        pout, plen, ppath = _run(model, text, keep, True)
        rows.append(dict(threshold=thr, margin=margin, oracle_stop=L, persistent_stop=plen, path=ppath,
                         mel_mean=float((pout[0] - omel[:, :, :pout[0].shape[2]]).abs().mean()) if plen == L else None))
        with open(os.path.join(OUT, "parity_persistent_fp32_gate_stop.json"), "w") as f:
            json.dump(rows, f, indent=1)
        assert ppath == 'persistent' and plen == L, rows
        assert rows[-1]['mel_mean'] < 1e-4
    assert len(cases) == 2


@pytest.mark.parametrize("Ti", [100, 187, 5])
def test_persistent_encoder_bilstm_matches_launch_chain_and_oracle(native_lib, Ti):
    """Encoder.inference's bi-LSTM (reference model.py:192-201) for one utterance as ONE persistent launch (W_hh rows in
    registers, h exchanged as granules; csrc/decode_persist.hip) against the Ti-launch chain and the oracle, fp32 mode:
    same inputs, another summation order."""
    from tacotron2_amd import engine
    steps = 48
    hp = gu.make_hparams("max_decoder_steps=%d" % steps)
    hp.gate_threshold = 2.0
    sd = gu.build_state_dict(hp, 77, perturb_bn=True)
    text = gu.make_text([Ti], 56)
    keep = orc.draw_masks_infer(hp, 1, steps, torch.Generator().manual_seed(6))
    model = _model(hp, sd, 'fp32')
    res = {}
    old = engine.PERSISTENT_ENCODER
    try:
        for on in (True, False):
            engine.PERSISTENT_ENCODER = on
            out, _, _ = _run(model, text, keep, True)
            res[on] = out
            assert model.last_encoder_path == ('persistent' if on else 'launch chain'), model.last_encoder_path
    finally:
        engine.PERSISTENT_ENCODER = old
    (omel, opost, ogate, oalign), _, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, 2.0)
    rows = {}
    for i, (nm, ref) in enumerate((("mel", omel), ("mel_post", opost), ("gate", ogate), ("align", oalign))):
        rows[nm] = dict(vs_chain_max=float((res[True][i] - res[False][i]).abs().max()),
                        vs_oracle_mean=float((res[True][i] - ref).abs().mean()), vs_oracle_max=float((res[True][i] - ref).abs().max()),
                        refmax=float(ref.abs().max()))
    with open(os.path.join(OUT, "parity_persistent_encoder_Ti%d.json" % Ti), "w") as f:
        json.dump(rows, f, indent=1)
    for nm in rows:
        assert rows[nm]["vs_oracle_mean"] < 1e-4 and rows[nm]["vs_oracle_max"] < 5e-4 * max(1.0, rows[nm]["refmax"]), (nm, rows)
        assert rows[nm]["vs_chain_max"] < 1e-4 * max(1.0, rows[nm]["refmax"]), (nm, rows)


@pytest.mark.parametrize("precision,B", [("bf16", 2), ("fp32", 3)])
def test_two_or_three_utterances_decode_one_after_the_other_on_the_persistent_kernel(native_lib, precision, B):
    """B = 2 / 3 ragged texts: the engine decodes the utterances consecutively on the single-utterance persistent kernel, each
    against its own rows of the encoder memory and of the dropout stream (engine.SMALL_BATCH_PERSISTENT; 12 us per utterance
    and step against the launch chain's ~38 us per step).  Real gate stops: the same stop frame per utterance as the launch
    chain, outputs as close to the chain's as the single-utterance kernel's are, and against the oracle's batched inference
    the fp32 form holds the parity tolerance."""
    from tacotron2_amd import engine
    steps = 96
    hp = gu.make_hparams("max_decoder_steps=%d" % steps)
    sd = gu.build_state_dict(hp, 1234, perturb_bn=True)
    in_lens = [61, 40, 23][:B]
    text = gu.make_text(in_lens, 9)
    lens = torch.tensor(in_lens)
    keep = orc.draw_masks_infer(hp, B, steps, torch.Generator().manual_seed(12))
    wg = sd['decoder.gate_layer.linear_layer.weight'].clone()
    wg[:, hp.decoder_rnn_dim:] *= -1.0                       # gate trajectories that move (as in the single-utterance tests)
    sd['decoder.gate_layer.linear_layer.weight'] = wg
    # a threshold every utterance crosses for the first time at a frame of its own, with the widest margin the forced
    # trajectories of the oracle offer (the bf16 route's gate noise is ~1e-3)
    (_, _, ogate, _), _, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, 2.0, input_lengths=lens)
    probs = torch.sigmoid(ogate.double().reshape(B, -1))
    best = None
    for cand in probs[:, 4:80].reshape(-1).tolist():
        for thr_ in (cand - 2e-3, cand + 2e-3):
            first, margin = [], 1.0
            for b in range(B):
                over = (probs[b] > thr_).nonzero()
                if over.numel() == 0:
                    first = None
                    break
                t0 = int(over[0])
                first.append(t0)
                margin = min(margin, float((probs[b, :t0 + 1] - thr_).abs().min()))
            if first is None or min(first) < 3 or len(set(first)) < 2:
                continue
            if best is None or margin > best[0]:
                best = (margin, thr_, first)
    assert best is not None and best[0] > 5e-4, best
    thr = best[1]
    hp.gate_threshold = thr
    (omel, opost, ogate, oalign), olen, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, thr, input_lengths=lens)
    model = _model(hp, sd, precision)
    model.dropout_masks = dict(prenet_infer=keep.to(DEV))
    outs = {}
    old = engine.SMALL_BATCH_PERSISTENT
    try:
        for route, cap in (("persistent", 3), ("chain", 1)):
            engine.SMALL_BATCH_PERSISTENT = cap
            with torch.no_grad():
                o = model.inference(text.to(DEV), lens.to(DEV))
            torch.cuda.synchronize()
            outs[route] = ([t.float().cpu() for t in o], model.last_inference_lengths.cpu().tolist(), model.last_decode_path)
    finally:
        engine.SMALL_BATCH_PERSISTENT = old
    (p, plen, ppath), (c, clen, cpath) = outs["persistent"], outs["chain"]
    assert ppath.startswith('persistent (%d utterances' % B), ppath
    assert cpath.startswith('launch chain'), cpath
    assert plen == clen, (plen, clen)
    assert len(set(plen)) > 1 or B == 1                      # the utterances really stop at different frames
    if precision == "fp32":
        assert plen == [int(v) for v in olen.tolist()], (plen, olen)
    T = min(p[0].shape[2], c[0].shape[2])
    for i, nm in enumerate(("mel", "mel_post")):
        d = (p[i][:, :, :T] - c[i][:, :, :T]).abs()
        assert float(d.max()) < (2e-4 if precision == "fp32" else 5e-2), (nm, float(d.max()))
    if precision == "fp32":
        To = min(T, omel.shape[2])
        assert float((p[0][:, :, :To] - omel[:, :, :To]).abs().max()) < 1e-4
    assert all(torch.isfinite(t).all() for t in p)


def test_batched_encoder_bilstm_persistent_leaves_a_co_residency_margin(native_lib):
    """the launch takes at most 3/4 of the workgroups the runtime says can be co-resident (occupancy query, not a
    hard-coded 4 per CU): B = 256 at H = 256 would need every slot of a 256-CU device and goes to the launch chain; B = 192 fits."""
    from tacotron2_amd import native as nv
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    d = nv.LstmSeq()
    d.B, d.T, d.H = 256, 50, 256
    why = nv.lstm_seq_batch_persistent_supported(d, 2, cus)
    if cus <= 256:
        assert why is not None and "co-resident" in why
    d.B = 192 if cus >= 256 else 64
    assert nv.lstm_seq_batch_persistent_supported(d, 2, cus) is None


@pytest.mark.parametrize("B,T", [(64, 61), (7, 23), (40, 9), (192, 50)])
def test_batched_encoder_bilstm_persistent_matches_the_launch_chain(native_lib, B, T):
    """Encoder bi-LSTM of a batch (reference model.py:181-188, packed-sequence semantics) as ONE persistent launch -- W_hh
    fragments in registers, h handed on through the output slab (write-through stores + step counters) -- against the launch
    chain on the same ragged batch: h, cell states and the activated gates the backward reads agree to summation order (both
    exact f32), rows behind an utterance's length are zero in both, partial row groups (B = 7, 40) included."""
    from tacotron2_amd import native as nv
    H, E = 256, 512
    g = torch.Generator().manual_seed(B * 1000 + T)
    lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=g).sort(descending=True)[0].to(torch.int32)
    lens[0] = T
    lens = lens.to(DEV)
    Whh = [(torch.randn(4 * H, H, generator=g) * 0.06).to(DEV) for _ in range(2)]
    GX0 = [(torch.randn(B * T, 4 * H, generator=g) * 0.5).to(DEV) for _ in range(2)]

    def run(persistent):
        mem = torch.full((B, T, E), float('nan'), device=DEV)
        descs, keep = [], []
        for d in range(2):
            GX, Cst = GX0[d].clone(), torch.full((T, B, H), float('nan'), device=DEV)
            desc = nv.LstmSeq()
            desc.B, desc.T, desc.H, desc.reverse = B, T, H, d
            desc.Whh, desc.GX = nv.ptr(Whh[d]), nv.ptr(GX)
            ov = mem.view(B * T, E)[:, d * H:(d + 1) * H]
            desc.out, desc.ld_out = nv.ptr(ov), E
            desc.C, desc.lens = nv.ptr(Cst), nv.ptr(lens, torch.int32)
            descs.append(desc)
            keep.append((GX, Cst))
        if persistent:
            assert nv.lstm_seq_batch_persistent_supported(descs[0], 2, torch.cuda.get_device_properties(0).multi_processor_count) is None
            flags = torch.full((nv.lstm_seq_batch_persistent_flag_words(B, H, 2),), 77, dtype=torch.int32, device=DEV)   # poisoned: the
            status = torch.full((1,), 5, dtype=torch.int32, device=DEV)                                                   # call zeroes them
            nv.lstm_seq_fwd2_batch_persistent(descs[0], descs[1], flags, status)
            assert int(status.item()) == 0
        else:
            nv.lstm_seq_fwd2(descs[0], descs[1])
        torch.cuda.synchronize()
        return mem.cpu(), [(a.cpu(), b.cpu()) for a, b in keep]

    mp, kp = run(True)
    mc, kc = run(False)
    assert torch.isfinite(mp).all()
    assert (mp - mc).abs().max().item() < 2e-6
    for d in range(2):
        assert (kp[d][0] - kc[d][0]).abs().max().item() < 2e-6          # activated gates (GX overwritten)
        assert (kp[d][1] - kc[d][1]).abs().max().item() < 2e-6          # cell states
    valid = torch.arange(T).unsqueeze(0) < lens.cpu().unsqueeze(1)
    assert torch.all(mp[~valid] == 0)


@pytest.mark.parametrize("B,T,H", [(64, 61, 256), (7, 23, 256), (40, 9, 256), (3, 1, 256), (10, 14, 64), (33, 20, 128)])
def test_batched_encoder_bilstm_bptt_persistent_matches_the_launch_chain(native_lib, B, T, H):
    """BPTT of the encoder bi-LSTM of a batch (reference model.py:181-188 under autograd) as ONE persistent launch -- gate
    gradients written straight into the DG slab, handed on with write-through stores + step counters, the recurrent data
    gradient as a split-bf16 (hi + lo) MFMA product against W_hh^T rows in registers -- against the chain of 2 T launches
    (pointwise cell backward + exact-f32 recurrent product) on the same ragged batch.  Tolerance: the split product carries
    ~2^-17 per term over K = 4H = 1024 and the error travels T steps back: max |diff| < 2e-5 of the slab's largest gradient
    (measured ~1e-6); rows behind an utterance's length are exactly zero in both; partial row groups (B = 7, 40) included."""
    from tacotron2_amd import native as nv
    E = 2 * H
    g = torch.Generator().manual_seed(B * 977 + T)
    lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=g).sort(descending=True)[0].to(torch.int32)
    lens[0] = T
    lens = lens.to(DEV)
    Whh = [(torch.randn(4 * H, H, generator=g) * 0.06).to(DEV) for _ in range(2)]
    WhhT = [w.t().contiguous() for w in Whh]
    GX0 = [(torch.randn(B * T, 4 * H, generator=g) * 0.5).to(DEV) for _ in range(2)]
    dmem = (torch.randn(B, T, E, generator=g) * 0.3).to(DEV)
    # forward on the chain: activated gates and cell states, what the backward reads
    mem = torch.zeros(B, T, E, device=DEV)
    fwd, GX, Cst = [], [], []
    for d in range(2):
        GX.append(GX0[d].clone()); Cst.append(torch.zeros(T, B, H, device=DEV))
        desc = nv.LstmSeq()
        desc.B, desc.T, desc.H, desc.reverse = B, T, H, d
        desc.Whh, desc.GX = nv.ptr(Whh[d]), nv.ptr(GX[d])
        desc.out, desc.ld_out = nv.ptr(mem.view(B * T, E)[:, d * H:(d + 1) * H]), E
        desc.C, desc.lens = nv.ptr(Cst[d]), nv.ptr(lens, torch.int32)
        fwd.append(desc)
    nv.lstm_seq_fwd2(fwd[0], fwd[1])
    torch.cuda.synchronize()

    def run(persistent):
        descs, DGs, keep = [], [], []
        for d in range(2):
            DG = torch.full((B * T, 4 * H), float('nan'), device=DEV)
            dX, dc = torch.zeros(4, B, H, device=DEV), torch.zeros(B, H, device=DEV)
            desc = nv.LstmSeq()
            desc.B, desc.T, desc.H, desc.reverse = B, T, H, d
            desc.WhhT = nv.ptr(WhhT[d])
            desc.GX, desc.C, desc.lens = nv.ptr(GX[d]), nv.ptr(Cst[d]), nv.ptr(lens, torch.int32)
            desc.dout, desc.ld_dout = nv.ptr(dmem.view(B * T, E)[:, d * H:(d + 1) * H]), E
            desc.DG = nv.ptr(DG)
            desc.dX, desc.dc, desc.dx_splits = nv.ptr(dX), nv.ptr(dc), 4
            descs.append(desc); DGs.append(DG); keep.append((dX, dc))
        if persistent:
            assert nv.lstm_seq_bwd2_batch_persistent_supported(descs[0], 2, torch.cuda.get_device_properties(0).multi_processor_count) is None
            flags = torch.full((nv.lstm_seq_batch_persistent_flag_words(B, H, 2),), 77, dtype=torch.int32, device=DEV)
            status = torch.full((1,), 5, dtype=torch.int32, device=DEV)
            nv.lstm_seq_bwd2_batch_persistent(descs[0], descs[1], flags, status)
            assert int(status.item()) == 0
        else:
            nv.lstm_seq_bwd2(descs[0], descs[1])
        torch.cuda.synchronize()
        return [x.cpu() for x in DGs]

    chain, pers = run(False), run(True)
    lens_c = lens.cpu()
    for d in range(2):
        c, p = chain[d].view(B, T, 4 * H), pers[d].view(B, T, 4 * H)
        assert torch.isfinite(p).all()
        scale = float(c.abs().max())
        assert float((c - p).abs().max()) < 2e-5 * scale, (d, float((c - p).abs().max()), scale)
        for b in range(B):
            assert float(p[b, int(lens_c[b]):].abs().max()) == 0.0 if int(lens_c[b]) < T else True


@pytest.mark.parametrize("B", [6, 12])
def test_small_batches_on_either_side_of_the_tile_boundary(native_lib, B):
    """Free-running decode at B = 6 and B = 12, ragged, real gate stops at different frames.  The launch chain serves a batch with
    the matrix-vector kernels up to t2amd_get_small_batch_max() rows and with the 64-row MFMA tiles above; a batch that shrinks
    across the boundary is compacted onto the other kernels in mid-sequence (B = 12 -> <= 8 rows with the boundary at 8; in the
    'bf16x3' mode the rows that are left go on with the fp32 kernels, the split images belong to the tiles).  Every combination
    must stop every utterance on the oracle's frame (fp32 / bf16x3) and hold the mel bound."""
    from tacotron2_amd import native as nv
    steps = 120
    hp = gu.make_hparams("max_decoder_steps=%d" % steps)
    sd = gu.build_state_dict(hp, 1234, perturb_bn=True)
    in_lens = [97, 83, 77, 64, 58, 51, 45, 38, 33, 27, 21, 15][:B] if B == 12 else [88, 61, 47, 40, 23, 17]
    text = gu.make_text(in_lens, 9)
    lens = torch.tensor(in_lens)
    keep = orc.draw_masks_infer(hp, B, steps, torch.Generator().manual_seed(12))
    wg = sd['decoder.gate_layer.linear_layer.weight'].clone()
    wg[:, hp.decoder_rnn_dim:] *= -1.0
    sd['decoder.gate_layer.linear_layer.weight'] = wg
    (_, _, ogate, _), _, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, 2.0, input_lengths=lens)
    sig = torch.sigmoid(ogate.double().reshape(B, steps))
    best = None
    for thr_ in torch.linspace(float(sig.min()), float(sig.max()), 600)[1:-1].tolist():
        over = sig > thr_
        stop = torch.where(over.any(1), over.float().argmax(1) + 1, torch.full((B,), steps))
        # at least half of the utterances stop early and at least two keep going 16 steps past the moment B - 8 have stopped
        early = sorted(int(v) for v in stop.tolist())
        if early[B // 2] >= steps - 20 or early[0] < 4 or early[-2] < early[max(B - 8, 1)] + 16:
            continue
        marg = min(float((sig[b, :int(stop[b])] - thr_).abs().min()) for b in range(B))
        if best is None or marg > best[0]:
            best = (marg, thr_, stop.tolist())
    assert best is not None and best[0] > 2e-4, best
    marg, thr, want = best
    hp.gate_threshold = thr
    (omel, _, _, _), olen, _ = orc.tacotron2_inference(sd, hp, text, keep, steps, thr, input_lengths=lens)
    assert [int(v) for v in olen.tolist()] == [int(v) for v in want]
    old = nv.small_batch_max_setting()
    rows = {}
    try:
        for prec in ("fp32", "bf16x3", "bf16"):
            model = _model(hp, sd, prec)
            model.dropout_masks = dict(prenet_infer=keep.to(DEV))
            for small_max in (8, 3):
                nv.set_small_batch_max(small_max)
                assert nv.small_batch_max(0) == small_max == nv.small_batch_max(1)
                with torch.no_grad():
                    o = model.inference(text.to(DEV), lens.to(DEV))
                torch.cuda.synchronize()
                got = model.last_inference_lengths.cpu().tolist()
                path = model.last_decode_path
                T = min(o[0].shape[2], omel.shape[2])
                err = float((o[0].float().cpu()[:, :, :T] - omel[:, :, :T]).abs().max())
                rows["%s_%d" % (prec, small_max)] = dict(path=path, stops=got, mel_max=err)
                assert path.startswith('launch chain'), path
                if B == 12 and small_max == 8:
                    assert 'compacted' in path, path                      # crossed the boundary in mid-sequence
                if prec != "bf16":
                    assert got == [int(v) for v in want], (prec, small_max, got, want)
                    assert err < 1e-4, (prec, small_max, err)
                else:
                    # bf16 operands: the gate noise (~1e-3) is larger than this threshold's margin, so stops may move; the frames
                    # both runs produced stay within the mode's tolerance
                    om = o[0].float().cpu()
                    e16 = max(float((om[b, :, :min(got[b], int(want[b]), T)] - omel[b, :, :min(got[b], int(want[b]), T)]).abs().mean())
                              for b in range(B))
                    rows["%s_%d" % (prec, small_max)]["mel_mean_common_frames"] = e16
                    assert e16 < 2e-2 * max(float(omel.abs().mean()), 1e-3), (got, want, e16)
                assert all(torch.isfinite(t).all() for t in o)
    finally:
        nv.set_small_batch_max(old)
    with open(os.path.join(OUT, "zz7_tile_boundary_B%d.json" % B), "w") as f:
        json.dump(dict(threshold=thr, margin=marg, oracle=want, runs=rows), f, indent=1)


def test_narrow_model_in_bf16_mode_keeps_the_matrix_vector_kernels_up_to_eight_rows(native_lib):
    """The bf16 tiles need widths that are multiples of 128; a narrower model (the tiny hparams: 64) decoded in bf16 mode at
    B = 5 must not be sent to them by the round-6 boundary (3 rows): t2amd_dec_infer_uses_tiles keeps it on the matrix-vector
    kernels, and the outputs agree with the fp32 mode's to the bf16 tolerance."""
    steps, B = 24, 5
    hp = gu.make_hparams(gu.TINY_HP + ",max_decoder_steps=%d" % steps)
    hp.gate_threshold = 2.0
    sd = gu.build_state_dict(hp, 1234, perturb_bn=True)
    in_lens = [19, 16, 12, 9, 7]
    text = gu.make_text(in_lens, 9)
    lens = torch.tensor(in_lens)
    keep = orc.draw_masks_infer(hp, B, steps, torch.Generator().manual_seed(12))
    outs = {}
    for prec in ("fp32", "bf16"):
        model = _model(hp, sd, prec)
        model.dropout_masks = dict(prenet_infer=keep.to(DEV))
        with torch.no_grad():
            o = model.inference(text.to(DEV), lens.to(DEV))
        torch.cuda.synchronize()
        assert model.last_decode_path.startswith('launch chain'), model.last_decode_path
        outs[prec] = o[0].float().cpu()
        assert torch.isfinite(outs[prec]).all() and outs[prec].shape[2] == steps
    ref = outs["fp32"]
    assert float((outs["bf16"] - ref).abs().mean()) < 2e-2 * max(float(ref.abs().mean()), 1e-3)
