"""Pin the oracle against the UNMODIFIED reference at the size that is TIMED (VERDICT r03 missing #5).

    python tests/golden/make_golden_fullsize.py        # writes tests/golden/fullsize_train_B64.pt  (~2-3 min of CPU)

Every committed tensor fixture is To <= 30, B <= 3; the engine is checked against the oracle at BASELINE configs[1]
(`synth_batch(64, 1234)`: B = 64, Ti = 177, To = 870 -- tests/test_zz5_fullsize_parity_gpu.py), but the oracle itself was
never checked against the reference there.  This script runs the reference's `Tacotron2.forward` + `Tacotron2Loss` +
`backward` (reference model.py:499-515, loss_function.py:8-19, train.py:226) on exactly that batch, weights (seed 1234)
and dropout masks (`oracle.draw_masks_train(hp, 64, 177, 870, Generator(1234))`, replayed through the F.dropout tape of
make_golden.py in the reference's draw order), asserts that the oracle reproduces it, and writes a DIGEST of the
reference's results (the tensors themselves are 80 MB): per output and per gradient the float64 sum, |.|-sum, L2 norm and
64 evenly spaced samples, plus the loss.  The digest travels; `/root/reference` does not:

  * tests/test_oracle_golden.py::test_oracle_equals_reference_digest_at_the_timed_size   oracle (live, ~40 s) vs digest, anywhere;
  * tests/test_reference_dropin_cpu.py::test_fullsize_digest_is_the_reference's          reference (live) vs digest, build container;
  * tests/test_zz5_fullsize_parity_gpu.py                                                 the GPU box's oracle run vs digest.
"""
import os
import sys
import time

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import golden_util as gu  # noqa: E402
from oracle import tacotron2_oracle as orc  # noqa: E402

NAME = "fullsize_train_B64"


def fullsize_case():
    """The batch, weights and masks of the full-size parity test and of bench.py's cpu_baseline / parity_check."""
    from tacotron2_amd.synth import synth_batch
    hp = gu.make_hparams("")
    sd = gu.build_state_dict(hp, 1234)
    text, il, mel, gate, ol = synth_batch(64, 1234)
    Ti, To = int(il.max()), int(ol.max())
    batch = (text[:, :Ti].contiguous(), il, mel[:, :, :To].contiguous(), gate[:, :To].contiguous(), ol)
    masks = orc.draw_masks_train(hp, 64, Ti, To, torch.Generator().manual_seed(1234))
    return hp, sd, batch, masks, Ti, To


def replay_list(masks, To):
    """Oracle-layout masks -> the order the reference draws them in: 3 x encoder, 2 x prenet, per step the
    attention-LSTM then the decoder-LSTM mask, 5 x postnet."""
    rec = list(masks['enc']) + list(masks['prenet'])
    for t in range(To):
        rec += [masks['att'][t], masks['dec'][t]]
    return rec + list(masks['post'])


def run_reference(ref_model, ref_loss, hp, sd, batch, masks, To):
    import make_golden as mg
    torch.manual_seed(1234)
    ref = ref_model.Tacotron2(hp)
    ref.load_state_dict(sd)
    tape = mg.DropoutTape()
    tape.replay, tape.pos = replay_list(masks, To), 0
    ref_model.F.dropout = tape
    ref.train()
    x, y = ref.parse_batch(tuple(t.clone() for t in batch))
    t0 = time.perf_counter()
    out = ref(x)
    loss = ref_loss.Tacotron2Loss()(out, y)
    loss.backward()
    dt = time.perf_counter() - t0
    assert tape.pos == len(tape.replay), (tape.pos, len(tape.replay))
    grads = {k: p.grad.detach() for k, p in ref.named_parameters()}
    return [o.detach() for o in out], loss.detach(), grads, dt


def digest_of(out, loss, grads):
    return dict(outputs=[gu.grad_digest(o) for o in out], loss=float(loss),
                grads={k: gu.grad_digest(v) for k, v in grads.items()})


def compare_to_digest(dg, out, loss, grads, out_tol=2e-5, grad_tol=1e-4, loss_tol=1e-6):
    """Results (oracle or reference, any host) against the committed digest.  Sums run over up to 5.6 M elements and the
    host's thread count changes summation order inside the matmuls, so integrals are compared relative to the |.|-sum and
    samples relative to the tensor's scale.  Returns the worst relative deviations seen."""
    worst = dict(out=0.0, grad=0.0)
    names = ('mel', 'mel_post', 'gate', 'align')
    for i, d in enumerate(dg['outputs']):
        g = gu.grad_digest(out[i])
        scale = max(d['abssum'], 1e-30)
        e = max(abs(g['sum'] - d['sum']) / scale, abs(g['abssum'] - d['abssum']) / scale, abs(g['l2'] - d['l2']) / max(d['l2'], 1e-30))
        smax = float(d['sample'].abs().max())
        es = float((g['sample'].double() - d['sample'].double()).abs().max()) / max(1.0, smax)
        assert e < out_tol and es < out_tol, (names[i], e, es)
        worst['out'] = max(worst['out'], e, es)
    rel = abs(float(loss) - dg['loss']) / max(abs(dg['loss']), 1e-30)
    assert rel < loss_tol, (float(loss), dg['loss'])
    worst['loss'] = rel
    # gradients: relative to the tensor's own size, with an absolute floor tied to the WHOLE gradient's scale -- the conv
    # biases in front of a BatchNorm have an exactly-zero true gradient (only rounding noise is left: L2 ~1e-7 of the
    # rest), where a relative comparison is meaningless
    gmax = max(d['l2'] for d in dg['grads'].values())
    floor = 1e-6 * gmax
    for k, d in dg['grads'].items():
        g = gu.grad_digest(grads[k])
        if k.endswith('.0.conv.bias'):
            # a bias in front of a BatchNorm: the true gradient is exactly zero, both sides hold rounding noise only
            assert d['l2'] < 1e-4 * gmax and g['l2'] < 1e-4 * gmax, (k, d['l2'], g['l2'])
            continue
        e = abs(g['l2'] - d['l2']) / (d['l2'] + floor)
        smax = float(d['sample'].abs().max()) + floor
        es = float((g['sample'].double() - d['sample'].double()).abs().max()) / smax
        assert e < grad_tol and es < 10 * grad_tol, (k, e, es, d['l2'], floor)
        worst['grad'] = max(worst['grad'], e, es / 10)
    return worst


def main():
    import make_golden as mg
    torch.set_num_threads(8)
    hp, sd, batch, masks, Ti, To = fullsize_case()
    assert (Ti, To) == (177, 870)
    real_dropout = torch.nn.functional.dropout
    ref_model, ref_loss = mg.import_reference()
    try:
        out, loss, grads, dt_ref = run_reference(ref_model, ref_loss, hp, sd, batch, masks, To)
    finally:
        torch.nn.functional.dropout = real_dropout
    print("reference step %.1f s, loss %.9f" % (dt_ref, float(loss)))
    t0 = time.perf_counter()
    oloss, oout, ograds, _ = orc.train_step_grads(sd, hp, batch, masks)
    dt_orc = time.perf_counter() - t0
    # ---- the pin itself: oracle == reference, tensor against tensor, at the timed size ----
    errs = {}
    for i, nm in enumerate(('mel', 'mel_post', 'gate', 'align')):
        errs[nm] = mg.close(oout[i], out[i], 2e-5, NAME + ' ' + nm)
    errs['loss'] = mg.close(oloss, loss, 1e-6, NAME + ' loss')
    gmax = 0.0
    for k, gr in grads.items():
        gmax = max(gmax, mg.close(ograds[k], gr, 5e-5, NAME + ' grad ' + k))
    print("[%s] oracle == reference: out errs %s, max grad err %.2e (oracle step %.1f s)"
          % (NAME, {k: '%.1e' % v for k, v in errs.items()}, gmax, dt_orc))
    dg = digest_of(out, loss, grads)
    frames = int(batch[4].sum())
    dg['meta'] = dict(batch="synth_batch(64, 1234)", B=64, Ti=Ti, To=To, valid_frames=frames, threads=torch.get_num_threads(),
                      # cpu_baseline calibration: the port and the real reference timed on the SAME
                      # batch, same host, same thread count, in the build container
                      reference_step_s=dt_ref, oracle_step_s=dt_orc,
                      reference_frames_per_s=frames / dt_ref, oracle_frames_per_s=frames / dt_orc,
                      oracle_vs_reference_max_abs=dict(outputs={k: float(v) for k, v in errs.items()}, grads=gmax),
                      torch=torch.__version__)
    print("gradient L2 norms: " + ", ".join("%s %.2e" % (k.split('.')[-3] + '.' + k.split('.')[-1], d['l2']) for k, d in
                                             sorted(dg['grads'].items(), key=lambda kv: kv[1]['l2'])[:12]))
    torch.save(dg, os.path.join(gu.GOLDEN_DIR, NAME + ".pt"))
    print("wrote", os.path.join(gu.GOLDEN_DIR, NAME + ".pt"))
    worst = compare_to_digest(dg, oout, oloss, ograds)
    print("oracle vs digest (the check that travels): %s" % worst)


if __name__ == "__main__":
    main()
