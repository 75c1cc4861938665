"""Generate the golden fixtures by running the UNMODIFIED reference (NVIDIA/tacotron2 at
/root/reference) on CPU, and pin the oracle against it.

    python tests/golden/make_golden.py            # writes tests/golden/*.pt

Only runs in the build container (the reference tree does not travel to the GPU box).  The
reference is imported through the shim documented in SURVEY.md Appendix C: stub modules for the
third-party imports the container lacks (librosa, unidecode, inflect), a CPU-safe
``get_mask_from_lengths`` (the reference hard-codes torch.cuda.LongTensor, utils.py:8) and a plain
attribute bag for hparams (tensorflow is absent).  ``torch.nn.functional.dropout`` is replaced by
a recorder/replayer with identical arithmetic (``x * (keep / (1 - p))``) so that the masks the
reference draws can be handed to the oracle and to the HIP engine.

For every case the script asserts, before writing anything:
  * tacotron2_amd.model.Tacotron2 reproduces the reference's state_dict bit-for-bit under the
    same seed (drop-in boundary: key set, shapes, init order);
  * oracle/tacotron2_oracle.py reproduces the reference's outputs, loss, every gradient and the
    BatchNorm buffer updates to tight tolerance on the same weights, inputs and masks.
"""
import os
import sys
import types

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_util as gu  # noqa: E402
from oracle import tacotron2_oracle as orc  # noqa: E402

REF = "/root/reference"


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub('librosa')
    stub('librosa.filters', mel=lambda *a, **k: None)
    stub('librosa.util', pad_center=None, tiny=None, normalize=None)
    sys.modules['librosa'].filters = sys.modules['librosa.filters']
    sys.modules['librosa'].util = sys.modules['librosa.util']
    stub('unidecode', unidecode=lambda s: s)
    stub('inflect', engine=lambda: types.SimpleNamespace(number_to_words=lambda *a, **k: 'num'))
    sys.path.insert(0, REF)
    import utils as ref_utils

    def get_mask_from_lengths(lengths):
        ids = torch.arange(0, torch.max(lengths).item(), dtype=torch.long, device=lengths.device)
        return ids < lengths.unsqueeze(1)
    ref_utils.get_mask_from_lengths = get_mask_from_lengths
    import model as ref_model
    ref_model.get_mask_from_lengths = get_mask_from_lengths
    import loss_function as ref_loss
    sys.path.remove(REF)
    return ref_model, ref_loss


class DropoutTape(object):
    """Stand-in for F.dropout: records the keep-masks it draws, or replays given ones."""

    def __init__(self):
        self.record = []
        self.replay = None
        self.pos = 0

    def __call__(self, x, p=0.5, training=True, inplace=False):
        if not training:
            return x
        if self.replay is not None:
            keep = self.replay[self.pos]
            self.pos += 1
        else:
            keep = (torch.rand(x.shape) >= p).to(torch.uint8)
            self.record.append(keep)
        scale = torch.ones((), dtype=x.dtype) / torch.tensor(1.0 - p, dtype=x.dtype)
        return x * (keep.to(x.dtype) * scale)


def close(a, b, tol, what):
    err = (a.double() - b.double()).abs().max().item()
    ref = b.double().abs().max().item()
    if not err <= tol * max(1.0, ref):
        raise AssertionError("%s: max abs err %.3e (ref max %.3e)" % (what, err, ref))
    return err


def run_train_case(name, case, ref_model, ref_loss):
    hp = gu.make_hparams(case['hp'])
    sd0 = gu.build_state_dict(hp, case['seed'])
    torch.manual_seed(case['seed'])
    ref = ref_model.Tacotron2(hp)
    # boundary pin: identical key set and bit-identical init
    rsd = ref.state_dict()
    assert list(rsd.keys()) == list(sd0.keys()), "state_dict key order differs"
    for k in rsd:
        assert torch.equal(rsd[k], sd0[k]), "init differs for %s" % k
    batch = gu.make_train_batch(case['in_lens'], case['out_lens'], hp.n_mel_channels, case['seed'])
    tape = DropoutTape()
    ref_model.F.dropout = tape
    torch.manual_seed(case['seed'] + 1)
    ref.train()
    x, y = ref.parse_batch(batch)
    out = ref(x)
    loss = ref_loss.Tacotron2Loss()(out, y)
    loss.backward()
    B, To = len(case['in_lens']), max(case['out_lens'])
    rec = tape.record
    ne = hp.encoder_n_convolutions
    masks = dict(enc=rec[:ne], prenet=rec[ne:ne + 2])
    steps = rec[ne + 2:ne + 2 + 2 * To]
    masks['att'] = torch.stack(steps[0::2])
    masks['dec'] = torch.stack(steps[1::2])
    masks['post'] = rec[ne + 2 + 2 * To:]
    assert len(masks['post']) == hp.postnet_n_convolutions, len(rec)
    ref_grads = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    ref_bufs = {k: v.detach().clone() for k, v in ref.state_dict().items()
                if 'running' in k or 'num_batches' in k}

    # ---- pin the oracle ----
    oloss, oout, ograds, obufs = orc.train_step_grads(sd0, hp, batch, masks)
    errs = {}
    for i, nm in enumerate(('mel', 'mel_post', 'gate', 'align')):
        errs[nm] = close(oout[i], out[i].detach(), 2e-5, name + ' ' + nm)
    errs['loss'] = close(oloss, loss.detach(), 1e-6, name + ' loss')
    gmax = 0.0
    for k, gr in ref_grads.items():
        gmax = max(gmax, close(ograds[k], gr, 5e-5, name + ' grad ' + k))
    for k, v in ref_bufs.items():
        close(obufs[k].float(), v.float(), 1e-6, name + ' buffer ' + k)
    print("[%s] oracle == reference: out errs %s, max grad err %.2e" % (name, {k: '%.1e' % v for k, v in errs.items()}, gmax))

    fx = dict(kind='train', hp=case['hp'], seed=case['seed'], in_lens=case['in_lens'], out_lens=case['out_lens'],
              sd_digest={k: gu.grad_digest(v.float()) for k, v in sd0.items() if v.dtype.is_floating_point},
              masks=gu.pack_masks(masks),
              outputs=[o.detach().clone() for o in out], loss=loss.detach().clone(),
              grad_digest={k: gu.grad_digest(v) for k, v in ref_grads.items()},
              grad_small={k: v for k, v in ref_grads.items() if v.numel() <= 4096},
              buffers={k: v for k, v in ref_bufs.items()})
    torch.save(fx, os.path.join(gu.GOLDEN_DIR, name + ".pt"))


def pick_threshold(sig, lo_frac=0.25, hi_frac=0.9):
    """Choose (threshold, stop_index) so that the first strict crossing happens mid-sequence."""
    T = len(sig)
    best = None
    for k in range(int(T * lo_frac), int(T * hi_frac)):
        prev = max(sig[:k])
        if sig[k] > prev:
            margin = sig[k] - prev
            if best is None or margin > best[0]:
                best = (margin, k, 0.5 * (sig[k] + prev))
    return best


def ref_infer(ref, tape, text, steps_masks=None):
    if steps_masks is not None:
        tape.replay, tape.pos = steps_masks, 0
    else:
        tape.replay, tape.record = None, []
    with torch.no_grad():
        out = ref.inference(text)
    return out


def run_infer_case(name, case, ref_model):
    hp = gu.make_hparams(case['hp'])
    sd0 = gu.build_state_dict(hp, case['seed'], perturb_bn=True)
    torch.manual_seed(case['seed'])
    ref = ref_model.Tacotron2(hp)
    ref.load_state_dict(sd0)
    ref.eval()
    text_all = gu.make_text(case['in_lens'], case['seed'])
    B = len(case['in_lens'])
    steps = case['steps']
    tape = DropoutTape()
    ref_model.F.dropout = tape
    per_utt = []
    masks_all = torch.zeros(steps, 2, B, hp.prenet_dim, dtype=torch.uint8)
    thr_all = []
    for b in range(B):
        text = text_all[b:b + 1, :case['in_lens'][b]]
        # pass 1: never stop (threshold 2.0), record the prenet masks and the gate trajectory
        ref.decoder.gate_threshold = 2.0
        torch.manual_seed(case['seed'] + 100 + b)
        out = ref_infer(ref, tape, text)
        rec = tape.record
        assert len(rec) == 2 * steps, len(rec)
        masks_all[:, 0, b] = torch.stack(rec[0::2]).squeeze(1)
        masks_all[:, 1, b] = torch.stack(rec[1::2]).squeeze(1)
        sig = torch.sigmoid(out[2].reshape(-1)).tolist()
        thr_all.append((sig, out))
    # one threshold for the whole batch: choose it on utterance 0, then read off every stop index
    best = pick_threshold(thr_all[0][0])
    assert best is not None, "no usable gate crossing; change the seed"
    thr = best[2]
    lengths = []
    for b in range(B):
        sig = thr_all[b][0]
        stop = next((i + 1 for i, s in enumerate(sig) if s > thr), steps)
        lengths.append(stop)
    # pass 2: replay with the chosen threshold; the reference must stop exactly there
    for b in range(B):
        text = text_all[b:b + 1, :case['in_lens'][b]]
        ref.decoder.gate_threshold = thr
        rec = []
        for t in range(steps):
            rec += [masks_all[t, 0, b:b + 1], masks_all[t, 1, b:b + 1]]
        out = ref_infer(ref, tape, text, rec)
        assert out[0].shape[2] == lengths[b], (out[0].shape, lengths[b])
        per_utt.append([o.detach().clone() for o in out])
    # ---- pin the oracle (B == 1 path per utterance, and the batched path) ----
    hp_o = gu.make_hparams(case['hp'])
    for b in range(B):
        text = text_all[b:b + 1, :case['in_lens'][b]]
        o, ln, hit = orc.tacotron2_inference(sd0, hp_o, text, masks_all[:, :, b:b + 1], steps, thr)
        assert int(ln[0]) == lengths[b], (ln, lengths[b])
        for i, nm in enumerate(('mel', 'mel_post', 'gate', 'align')):
            close(o[i], per_utt[b][i], 2e-5, '%s utt %d %s' % (name, b, nm))
    if B > 1:
        o, ln, hit = orc.tacotron2_inference(sd0, hp_o, text_all, masks_all, steps, thr,
                                             input_lengths=torch.tensor(case['in_lens']))
        assert ln.tolist() == lengths, (ln, lengths)
        for b in range(B):
            L, Tb = lengths[b], case['in_lens'][b]
            close(o[0][b, :, :L], per_utt[b][0][0], 5e-5, '%s batched mel utt %d' % (name, b))
            close(o[1][b, :, :L], per_utt[b][1][0], 5e-5, '%s batched mel_post utt %d' % (name, b))
            close(o[3][b, :L, :Tb], per_utt[b][3][0], 5e-5, '%s batched align utt %d' % (name, b))
    print("[%s] oracle == reference; threshold %.6f, stop lengths %s" % (name, thr, lengths))
    fx = dict(kind=case['kind'], hp=case['hp'], seed=case['seed'], in_lens=case['in_lens'], steps=steps,
              threshold=thr, lengths=lengths, text=text_all, masks=gu.pack_mask(masks_all),
              sd_digest={k: gu.grad_digest(v.float()) for k, v in sd0.items() if v.dtype.is_floating_point},
              outputs=per_utt)
    torch.save(fx, os.path.join(gu.GOLDEN_DIR, name + ".pt"))


def main():
    torch.set_num_threads(8)
    ref_model, ref_loss = import_reference()
    real_dropout = torch.nn.functional.dropout
    try:
        only = set(sys.argv[1:])                      # optional: regenerate just the named cases
        for name, case in gu.CASES.items():
            if only and name not in only:
                continue
            if case['kind'] == 'train':
                run_train_case(name, case, ref_model, ref_loss)
            else:
                run_infer_case(name, case, ref_model)
    finally:
        torch.nn.functional.dropout = real_dropout


if __name__ == "__main__":
    main()
